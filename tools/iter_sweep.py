import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'brax-rodent-run_amd'))
import numpy as np, torch
from rodent_amd import assets, hip, envs
from tests import util
dev=torch.device('cuda:0'); N=2048
env=envs.get_environment('rodent', track_pos=util.synthetic_track(), num_envs=N, xml_path='rodent_optimized.xml', iterations=8, ls_iterations=8, device=dev)
st0=env.reset(0)
# advance 30 steps so the population is in contact
state=st0
for _ in range(30): state=env.step(state, torch.rand(N,30,device=dev)*2-1)
ps=state.pipeline_state
base=dict(qpos=ps.qpos, qvel=ps.qvel, act=ps.act, qacc_warmstart=ps.qacc_warmstart)
for it,ls in ((0,0),(1,8),(2,8),(4,8),(8,8),(8,1),(8,0)):
    b=hip.Batch(hip.Model(assets.asset_path('rodent_optimized'),it,ls), N, dev)
    b.set_timing(True)
    for rep in range(4):
        st={k:v.clone() for k,v in base.items()}
        b.pipeline_step(st, torch.rand(N,30,device=dev)*2-1, 10)
    torch.cuda.synchronize(); ms,n=b.kernel_time()
    print(f'iterations={it} ls={ls}: {ms/n:.3f} ms per launch (10 substeps, {N} envs)')
