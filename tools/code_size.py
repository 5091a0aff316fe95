"""Attribute the instructions of one rr_step_kernel instance to the rr_kernel.h functions they were inlined from.
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -gline-tables-only -o k.s rr_api.hip
       python tools/code_size.py k.s [kernel-symbol-substring]"""
import re, sys, collections
asm = open(sys.argv[1]).read().split("\n")
sym = sys.argv[2] if len(sys.argv) > 2 else "ILi2ELi2ELi1ELb0EE"
src = open(__file__.replace("tools/code_size.py", "brax-rodent-run_amd/csrc/rr_kernel.h")).read().split("\n")
funcs = []   # (line, name)
for i, l in enumerate(src, 1):
    m = re.match(r"\s*(?:template <[^>]*>\s*)?(?:static )?__device__ (?:__forceinline__ )?[\w:<>\*& ]+?\b(\w+)\(", l)
    if m: funcs.append((i, m.group(1)))
    m = re.match(r"__global__", l)
    if m: funcs.append((i, "kernel_body"))
def owner(line):
    o = "?"
    for s, n in funcs:
        if s <= line: o = n
    return o
fid = None
for l in asm:
    m = re.match(r'\s*\.file\s+(\d+) "[^"]*" "rr_kernel.h"', l)
    if m: fid = m.group(1)
start = next(i for i, l in enumerate(asm) if l.startswith("_Z14rr_step_kernel" + sym))
cnt, lines = collections.Counter(), collections.Counter()
cur = "?"; curline = 0; total = 0
for l in asm[start:]:
    if l.startswith(".Lfunc_end"): break
    m = re.match(r"\s*\.loc\s+(\d+) (\d+)", l)
    if m:
        if m.group(1) == fid: curline = int(m.group(2)); cur = owner(curline)
        else: cur = "<hip headers>"
        continue
    if re.match(r"\t[a-z]\w+", l) and not l.startswith("\t."):
        cnt[cur] += 1; total += 1; lines[(cur, curline)] += 1
print("total instructions", total)
for k, v in cnt.most_common(40): print(f"{v:7d} {k}")
if len(sys.argv) > 3:
    for (k, ln), v in lines.most_common(40): print(f"{v:6d} {k}:{ln}  {src[ln-1].strip()[:90] if ln else ''}")
