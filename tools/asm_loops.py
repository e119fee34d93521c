#!/usr/bin/env python3
"""List the loops of one kernel in build/asm/*.s with instruction-class counts; print the innermost one with -p.
usage: tools/asm_loops.py build/asm/tomo_project.hip.s <substring of mangled name> [-p]"""
import re, sys
from collections import Counter
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w*:', l) and key in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
loops = []
for i, l in enumerate(body):
    m = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)', l) or re.search(r's_branch (\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
def cls(op):
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_'): return 'salu'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
    if op.startswith('ds_'): return 'lds'
    return 'other'
for a, b in loops:
    seg = [l.strip() for l in body[a:b + 1] if l.strip() and not l.strip().startswith((';', '.'))]
    print(a, b, len(seg), dict(Counter(cls(l.split()[0]) for l in seg)))
if '-p' in sys.argv and loops:
    a, b = min(loops, key=lambda t: t[1] - t[0])
    print("\n".join(body[a:b + 1]))
