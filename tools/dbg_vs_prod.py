import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/brax-rodent-run_amd')
import torch
from tests.test_gpu_env import _mk_env
N=64
env=_mk_env(N)
state=env.reset(5)
g=torch.Generator(device="cuda:0"); g.manual_seed(3)
for _ in range(2): state=env.step(state, torch.rand(N, env.action_size, device="cuda:0", generator=g)*2-1)
ps=state.pipeline_state
ctrl=torch.rand(N, env.action_size, device="cuda:0", generator=g)*2-1
for nf in (1,2,10):
    a=dict(qpos=ps.qpos.clone(), qvel=ps.qvel.clone(), act=ps.act.clone(), qacc_warmstart=ps.qacc_warmstart.clone())
    b={k:v.clone() for k,v in a.items()}
    env._batch.pipeline_step(a, ctrl, nf)
    dbg=torch.zeros(N, env._batch.dims.dbg_floats, device="cuda:0")
    env._batch.pipeline_step(b, ctrl, nf, out=dict(debug=dbg))
    torch.cuda.synchronize()
    print(nf, {k: float((a[k]-b[k]).abs().max()) for k in a}, 'envs differing', int(((a['qvel']-b['qvel']).abs().amax(1)>0).sum()))
