#!/usr/bin/env python3
"""Instruction mix of a line range of an assembly listing (as cut by tools/isa_fn.py):  python tools/isa_range.py file.s first last"""
import collections
import sys
body = open(sys.argv[1]).read().split('\n')
a, b = int(sys.argv[2]) - 1, int(sys.argv[3])
c, ops = collections.Counter(), collections.Counter()
for l in body[a:b]:
    t = l.strip().split()
    if not t or t[0].startswith((';', '.')) or t[0].endswith(':'):
        continue
    op = t[0]
    if op.startswith('v_mfma'): c['mfma'] += 1
    elif op.startswith('v_'): c['valu'] += 1
    elif op.startswith('s_waitcnt'): c['waitcnt'] += 1
    elif op.startswith('s_'): c['salu'] += 1
    elif op.startswith('ds_'): c['lds'] += 1
    else: c['vmem'] += 1
    ops[op + (' dpp' if ('quad_perm' in l or 'row_sh' in l) else '')] += 1
print(dict(c))
print(ops.most_common(60))
