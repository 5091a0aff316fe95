"""Shared child-process snippets of ab_bench.py / multi_bench.py."""
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os, time
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "brax-rodent-run_amd"))
import torch, numpy as np
from rodent_amd import assets, hip
adir = os.environ["RR_ASSETS"]
N = 2048
b = hip.Batch(hip.Model(os.path.join(adir, "rodent_optimized.rrm"), 8, 8), N, torch.device("cuda:0"))
st0 = torch.load(os.path.join(%r, "gpurun_out", "ab_state.pt"))
b.set_timing(True)
for rep in range(6):
    st = {k: v.clone().cuda() for k, v in st0.items()}
    b.pipeline_step(st, torch.rand(N, 30, device="cuda:0") * 2 - 1, 10)
torch.cuda.synchronize(); ms, n = b.kernel_time()
print(ms / n)
''' % (ROOT, ROOT, ROOT)
# a common in-contact start state, produced with build A
gen = r'''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "brax-rodent-run_amd"))
import torch
from rodent_amd import envs
from tests import util
env = envs.get_environment("rodent", track_pos=util.synthetic_track(), num_envs=2048, xml_path="rodent_optimized.xml", iterations=8, ls_iterations=8, device="cuda:0")
s = env.reset(0)
for _ in range(30): s = env.step(s, torch.rand(2048, 30, device="cuda:0") * 2 - 1)
ps = s.pipeline_state
os.makedirs(os.path.join(%r, "gpurun_out"), exist_ok=True)
torch.save({k: getattr(ps, k).cpu() for k in ("qpos", "qvel", "act", "qacc_warmstart")}, os.path.join(%r, "gpurun_out", "ab_state.pt"))
''' % (ROOT, ROOT, ROOT, ROOT)
