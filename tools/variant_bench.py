"""Time several builds of librodent_hip.so on ONE in-contact start state with ONE action sequence (seeded), kernel hipEvent
time, interleaved repeats; prints mean ms per 2048-env step and a checksum of the final state (pure scheduling / memory-
instruction experiments must leave it bit-identical to the baseline).
usage: variant_bench.py lib0.so lib1.so[:assets_dir] ...      (lib0 = baseline; the start state is generated with the in-tree
build; an optional assets directory after ':' gives that build its own model blobs, e.g. other schedule-table parameters)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ab_bench_code as C
code = r'''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "brax-rodent-run_amd"))
import torch, hashlib
from rodent_amd import assets, hip
N = 2048
b = hip.Batch(hip.Model(assets.asset_path("rodent_optimized"), 8, 8), N, torch.device("cuda:0"))
st0 = torch.load(os.path.join(%r, "gpurun_out", "ab_state.pt"))
g = torch.Generator(device="cuda:0"); g.manual_seed(7)
st = {k: v.clone().cuda() for k, v in st0.items()}
for rep in range(24):
    if rep == 4: b.set_timing(True)
    b.pipeline_step(st, torch.rand(N, 30, device="cuda:0", generator=g) * 2 - 1, 10)
torch.cuda.synchronize(); ms, n = b.kernel_time()
h = hashlib.sha1(b"".join(st[k].cpu().numpy().tobytes() for k in sorted(st))).hexdigest()[:12]
print(ms / n, h)
''' % (ROOT, ROOT, ROOT)
libs = sys.argv[1:]
subprocess.check_call([sys.executable, "-c", C.gen])
res = {l: [] for l in libs}
sums = {}
for rep in range(3):
    for l in libs:
        lib, _, adir = l.partition(":")
        env = dict(os.environ, RR_LIB=os.path.abspath(lib), **({"RR_ASSETS": os.path.abspath(adir)} if adir else {}))
        out = subprocess.check_output([sys.executable, "-c", code], env=env).decode().strip().split("\n")[-1].split()
        res[l].append(float(out[0])); sums[l] = out[1]
base = min(res[libs[0]])
for l in libs:
    print(f"{os.path.basename(l):34s} {min(res[l]):8.4f} ms  ({100 * (min(res[l]) / base - 1):+6.2f} %)  runs {['%.4f' % x for x in res[l]]}  state {sums[l]} {'== baseline' if sums[l] == sums[libs[0]] else 'DIFFERS'}", flush=True)
