"""List the backward branches (loops) of one kernel in an llvm-objdump -d --symbolize-operands listing by byte span.
usage: loop_spans.py k.dis <kernel-symbol-substring>"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
sym = sys.argv[2]
start = next(i for i, l in enumerate(lines) if sym in l and l.endswith(">:"))
labels, br, base, end = {}, [], None, None
ins = []
for l in lines[start + 1:]:
    m = re.match(r"^([0-9a-f]+) <(\w+)>:", l)
    if m:
        if not m.group(2).startswith("L"): break
        labels[m.group(2)] = int(m.group(1), 16); continue
    m = re.search(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]{12}):", l)
    if m:
        a = int(m.group(3), 16)
        if base is None: base = a
        end = a
        ins.append((a, m.group(1), m.group(2)))
for k, (a, op, args) in enumerate(ins):
    if op.startswith("s_cbranch") or op == "s_branch":
        t = labels.get(args.split()[-1])
        if t is not None and t <= a: br.append((a - t, t - base, a - base, "short"))
    if op == "s_getpc_b64" and k + 2 < len(ins):
        o1, o2 = ins[k + 1], ins[k + 2]
        if o1[1] in ("s_add_u32", "s_sub_u32"):
            lo = int(o1[2].split(",")[-1].strip(), 0)
            hi = int(o2[2].split(",")[-1].strip(), 0)
            off = lo + (hi << 32)
            if off >= 1 << 63: off -= 1 << 64
            if o1[1] == "s_sub_u32": off = -off
            t = o1[0] + off
            if t <= a: br.append((a - t, t - base, a - base, "long"))
print("kernel bytes", end - base)
for s, t, a, k in sorted(br, reverse=True)[:int(sys.argv[3]) if len(sys.argv) > 3 else 20]:
    print(f"span {s:7d} B   +{t:7d} .. +{a:7d}  {k}")
