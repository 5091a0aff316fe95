"""Time several builds of librodent_hip.so on the same in-contact start state (interleaved repeats, kernel hipEvent time).
usage: multi_bench.py assets_dir lib1.so lib2.so ...   (the state is generated with the in-tree build)"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ab_bench_code as C
adir = os.path.abspath(sys.argv[1]); libs = sys.argv[2:]
subprocess.check_call([sys.executable, "-c", C.gen])
res = {l: [] for l in libs}
for rep in range(2):
    for l in libs:
        env = dict(os.environ, RR_LIB=os.path.abspath(l), RR_ASSETS=adir)
        res[l].append(float(subprocess.check_output([sys.executable, "-c", C.code], env=env).decode().strip().split("\n")[-1]))
base = min(res[libs[0]])
for l in libs: print(f"{os.path.basename(l):28s} {min(res[l]):8.3f} ms   delta {min(res[l]) - base:+7.3f}", flush=True)
