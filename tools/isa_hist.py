#!/usr/bin/env python
"""Opcode histogram of the largest loop of one kernel in a hipcc -S listing: tools/isa_hist.py file.s <mangled-prefix>"""
import collections, re, sys
lines = open(sys.argv[1]).read().split('\n')
pref = sys.argv[2]
s = next(i for i, l in enumerate(lines) if l.startswith(pref) and ':' in l and not l.startswith('\t'))
e = next(i for i in range(s, len(lines)) if 's_endpgm' in lines[i])
body = lines[s:e]
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
best = None
for i, l in enumerate(body):
    m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        span = (labels[m.group(1)], i)
        if best is None or span[1] - span[0] > best[1] - best[0]: best = span
ops = collections.Counter()
for l in body[best[0]:best[1]]:
    t = l.strip().split()
    if t and re.match(r'^(v_|s_|ds_|global_|buffer_)', t[0]): ops[t[0]] += 1
grp = lambda p: sum(c for o, c in ops.items() if o.startswith(p))
print("loop lines", best, "VALU", grp('v_'), "SALU", grp('s_'), "DS", grp('ds_'), "GLOBAL", grp('global'))
for o, c in ops.most_common(45): print(f"{o:28s}{c}")
