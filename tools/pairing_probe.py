"""Does pairing a heavy with a light environment on every SIMD shorten the launch now that waves carry a priority?
Blocks b and b + N/2 share a SIMD (tools/ubench/wg_placement.hip).  Per-env cost = cycles of the instrumented build for the
same state; orders tried: as is, heavy|light pairs, heavy|heavy pairs (worst case), random.  Prints kernel ms per order."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "brax-rodent-run_amd"))
import numpy as np, torch
from rodent_amd import envs
from tests import util
dev = torch.device("cuda:0"); N = 2048
env = envs.get_environment("rodent", track_pos=util.synthetic_track(), num_envs=N, xml_path="rodent_optimized.xml", iterations=8, ls_iterations=8, device=dev)
s = env.reset(0)
g = torch.Generator(device=dev); g.manual_seed(3)
for _ in range(30):
    s = env.step(s, torch.rand(N, 30, device=dev, generator=g) * 2 - 1)
ps = s.pipeline_state
st0 = dict(qpos=ps.qpos.clone(), qvel=ps.qvel.clone(), act=ps.act.clone(), qacc_warmstart=ps.qacc_warmstart.clone())
ctrl = torch.rand(N, 30, device=dev, generator=g) * 2 - 1
b = env._batch
buf = torch.zeros(N, 24, dtype=torch.int64, device=dev)
b.set_profile(buf)
st = {k: v.clone() for k, v in st0.items()}
b.pipeline_step(st, ctrl, 10)
torch.cuda.synchronize()
cost = buf.sum(1).cpu().numpy().astype(np.float64)
b.set_profile(None)
order = np.argsort(-cost)                       # heaviest first
half = N // 2
perms = {"as is": np.arange(N)}
p = np.empty(N, int); p[:half] = order[:half]; p[half:] = order[::-1][:half]; perms["heavy with light"] = p
p = np.empty(N, int); p[:half] = order[0::2]; p[half:] = order[1::2]; perms["heavy with heavy"] = p
perms["random"] = np.random.default_rng(0).permutation(N)
print("per-env cycles: median %.0f max %.0f" % (np.median(cost), cost.max()))
for name, p in perms.items():
    pt = torch.tensor(p, device=dev)
    times = []
    for rep in range(5):
        st = {k: v[pt].clone() for k, v in st0.items()}
        c = ctrl[pt].contiguous()
        b.set_timing(True)
        b.pipeline_step(st, c, 10)
        torch.cuda.synchronize()
        ms, n = b.kernel_time()
        times.append(ms / n)
        b.set_timing(False)
    print(f"{name:18s} {min(times):.4f} ms  (runs {['%.4f' % t for t in times]})", flush=True)
