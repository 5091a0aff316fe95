import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'brax-rodent-run_amd'))
import numpy as np, torch, time
from oracle import ref
from tests import util
from rodent_amd import assets, hip
N=8
st, M, m = util.settled_states(ref, 'rodent_optimized', N, seed=1, iterations=(8,8))
dev=torch.device('cuda:0')
batch=hip.Batch(hip.Model(assets.asset_path('rodent_optimized'),8,8), N, dev)
ctrl=np.random.default_rng(5).uniform(-1,1,(N,M.nu))
ds={k: torch.tensor(v,dtype=torch.float32,device=dev) for k,v in st.items()}
dbg=torch.zeros(N,batch.dims.dbg_floats,device=dev)
batch.pipeline_step(ds, torch.tensor(ctrl,dtype=torch.float32,device=dev), 1, out=dict(debug=dbg))
torch.cuda.synchronize()
lay=batch.debug_layout(); dbg=dbg.cpu().numpy().astype(np.float64)
d=util.oracle_forward(ref,M,st,0,ctrl[0])
o,n=lay['xpos']; got=dbg[0,o:o+n].reshape(-1,3); want=d.get('xpos').reshape(-1,3)
err=np.abs(got-want).max(axis=1)
print('xpos err per body', np.round(err,6))
b=int(err.argmax()); print('worst body',b,'got',got[b],'want',want[b],'parent',m['body_parentid'][b],'jntnum',m['body_jntnum'][b])
# quick timing
for Nb in (2048, 4096):
    batch=hip.Batch(hip.Model(assets.asset_path('rodent_optimized'),8,8), Nb, dev)
    idx=np.random.default_rng(0).integers(0,N,Nb)
    ds={k: torch.tensor(v[idx],dtype=torch.float32,device=dev).contiguous() for k,v in st.items()}
    c=torch.rand(Nb,M.nu,device=dev)*2-1
    for _ in range(3): batch.pipeline_step(ds,c,10)
    torch.cuda.synchronize(); t=time.time()
    K=20
    for _ in range(K):
        c=torch.rand(Nb,M.nu,device=dev)*2-1
        batch.pipeline_step(ds,c,10)
    torch.cuda.synchronize(); dt=(time.time()-t)/K
    print(f'N={Nb}: {dt*1e3:.3f} ms/env-step-batch -> {Nb/dt:.0f} env-steps/s ; z mean {ds["qpos"][:,2].mean().item():.4f} finite {torch.isfinite(ds["qpos"]).all().item()}')
