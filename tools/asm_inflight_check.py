#!/usr/bin/env python3
"""Static check of a hand-waited loop in a hipcc assembly listing: no instruction may touch a VGPR that is the destination of a
global_load still in flight, given that loads return in issue order and `s_waitcnt vmcnt(N)` leaves the N youngest in flight.
    python tools/asm_inflight_check.py kernel.s [first_line last_line]   (lines of ONE loop body; it is walked twice)"""
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1, len(lines))
body = lines[lo - 1:hi]


def regs(tok):
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r'v(\d+)', tok)
    return {int(m.group(1))} if m else set()


inflight = []   # list of (dest regs, line)
bad = 0
for rep in range(2):
    for n, ln in enumerate(body):
        t = ln.strip()
        if not t or t.startswith((';', '.', '//')) or t.endswith(':'):
            continue
        t = t.split(';')[0].strip()
        op, _, rest = t.partition(' ')
        toks = [x.strip() for x in re.split(r'[,\s]+', rest) if x.strip()]
        if op == 's_waitcnt':
            m = re.search(r'vmcnt\((\d+)\)', t)
            if m:
                k = int(m.group(1))
                inflight = inflight[len(inflight) - k:] if k < len(inflight) else inflight
                if k == 0:
                    inflight = []
            continue
        used = set()
        for x in toks:
            used |= regs(x)
        for dest, where in inflight:
            if used & dest:
                bad += 1
                if bad <= 20:
                    print(f'pass {rep}: line {lo + n}: `{t}` touches v{sorted(used & dest)} of the load issued at line {where}')
        if op.startswith('global_load') or op.startswith('scratch_load') or op.startswith('buffer_load'):
            inflight.append((regs(toks[0]), lo + n))
print('in-flight register hazards:', bad)
