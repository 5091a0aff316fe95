"""A/B timing of two builds of librodent_hip.so in ONE process sequence on the same GPU (interleaved repeats).
usage: ab_bench.py libA.so assetsA_dir libB.so assetsB_dir"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from ab_bench_code import code, gen
subprocess.check_call([sys.executable, "-c", gen])
libs = [(sys.argv[1], sys.argv[2]), (sys.argv[3], sys.argv[4])]
res = {0: [], 1: []}
for rep in range(3):
    for i, (lib, adir) in enumerate(libs):
        env = dict(os.environ, RR_LIB=os.path.abspath(lib), RR_ASSETS=os.path.abspath(adir))
        out = subprocess.check_output([sys.executable, "-c", code], env=env).decode().strip().split("\n")[-1]
        res[i].append(float(out))
print(json.dumps({"A": sys.argv[1], "A_ms": res[0], "B": sys.argv[3], "B_ms": res[1]}))
