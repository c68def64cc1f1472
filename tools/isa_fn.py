#!/usr/bin/env python3
"""Cut one kernel's body out of a device assembly file and count its instructions by class.
   python tools/isa_fn.py file.s <mangled-name substring> [out.s]"""
import collections
import re
import sys
s = open(sys.argv[1]).read()
m = re.search(r'\n(_Z\w*' + re.escape(sys.argv[2]) + r'\w*):', s)
if not m:
    sys.exit('no such function')
i = m.start()
j = s.find('.Lfunc_end', i)
body = s[i:j].split('\n')
c = collections.Counter()
for l in body:
    t = l.strip().split()
    if not t or t[0].startswith((';', '.')) or t[0].endswith(':'):
        continue
    op = t[0]
    if op.startswith('v_mfma'): c['mfma'] += 1
    elif op.startswith('v_'): c['valu'] += 1
    elif op.startswith('s_waitcnt'): c['waitcnt'] += 1
    elif op.startswith('s_'): c['salu'] += 1
    elif op.startswith('ds_'): c['lds'] += 1
    elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): c['vmem:' + op] += 1
    else: c[op] += 1
print(m.group(1), len(body), 'lines')
print(dict(c))
if len(sys.argv) > 3:
    open(sys.argv[3], 'w').write('\n'.join(body))
