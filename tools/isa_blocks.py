#!/usr/bin/env python
"""Per-basic-block instruction counts of one kernel in a hipcc -S listing: tools/isa_blocks.py file.s <mangled-prefix>"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
pref = sys.argv[2]
s = next(i for i, l in enumerate(lines) if l.startswith(pref) and ':' in l and not l.startswith('\t'))
e = next(i for i in range(s, len(lines)) if 's_endpgm' in lines[i])
name, cnt, out = 'entry', {}, []
def flush():
    if cnt: out.append((name, dict(cnt)))
for l in lines[s + 1:e + 1]:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m or l.startswith('; %bb.'):
        flush(); cnt = {}
        name = m.group(1) if m else l.split()[1]
        continue
    t = l.strip().split()
    if not t or not re.match(r'^(v_|s_|ds_|global_|buffer_)', t[0]): continue
    k = 'VALU' if t[0].startswith('v_') else 'SALU' if t[0].startswith('s_') else 'DS' if t[0].startswith('ds_') else 'VMEM'
    cnt[k] = cnt.get(k, 0) + 1
    if t[0].startswith('s_cbranch') or t[0] == 's_branch': cnt['br'] = cnt.get('br', '') + ' ' + t[0][2:] + '->' + t[-1]
flush()
for n, c in out: print(f"{n:14s} VALU {c.get('VALU',0):4d} SALU {c.get('SALU',0):4d} DS {c.get('DS',0):3d} VMEM {c.get('VMEM',0):3d} {c.get('br','')}")
