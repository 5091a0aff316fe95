"""Diagnostic: per-phase cycle shares of the step kernel (s_memtime-instrumented build; never timed)."""
import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'brax-rodent-run_amd'))
import numpy as np, torch
from rodent_amd import assets, hip, envs
from tests import util
dev=torch.device('cuda:0'); N=int(sys.argv[1]) if len(sys.argv)>1 else 2048
env=envs.get_environment('rodent', track_pos=util.synthetic_track(), num_envs=N, xml_path='rodent_optimized.xml', iterations=8, ls_iterations=8, device=dev)
state=env.reset(0)
for _ in range(30): state=env.step(state, torch.rand(N,30,device=dev)*2-1)
ps=state.pipeline_state
st=dict(qpos=ps.qpos.clone(), qvel=ps.qvel.clone(), act=ps.act.clone(), qacc_warmstart=ps.qacc_warmstart.clone())
buf=torch.zeros(N,24,dtype=torch.int64,device=dev)
b=env._batch; b.set_profile(buf)
b.pipeline_step(st, torch.rand(N,30,device=dev)*2-1, 10)
torch.cuda.synchronize()
c=buf.cpu().numpy().astype(np.float64)
names=['kinematics','com_pos','velocity_sweep','backward_sweep','mass_matrix','factor','smooth+solve','constraints','solver_init(3 ctx)','linesearch','update_constraint','update_gradient','cg_misc','euler(factor+solve)','epilogue','frame_head','ls: put_vec+jac_mul','ls: staging+sums','ls: 2 first points','ls: iterations','-','-','-','-']
tot=c.sum(1)
print(f'N={N} envs, 10 substeps; cycles per env (median over envs): total {np.median(tot):.0f}')
for i,n in enumerate(names):
    print(f'  {n:26s} {np.median(c[:,i]):10.0f} cyc  {100*np.median(c[:,i]/tot):5.1f} %')
print('per-env total cycles: min %.0f  p10 %.0f  median %.0f  p90 %.0f  max %.0f  (launch time = slowest SIMD pair)' % (tot.min(), np.percentile(tot,10), np.median(tot), np.percentile(tot,90), tot.max()))
order = np.argsort(tot)
slow, mid = order[-N // 20:], order[N // 2 - N // 40:N // 2 + N // 40]
print('slowest 5 %% of the envs vs the median 5 %%, cycles per phase (the launch lasts as long as the slowest env):')
for i, n in enumerate(names):
    if c[:, i].max() > 0:
        print(f'  {n:26s} slow {c[slow, i].mean():10.0f}   median {c[mid, i].mean():10.0f}   diff {c[slow, i].mean() - c[mid, i].mean():+10.0f}')
