#!/usr/bin/env python3
"""Where the compiler's scratch (spill) traffic of conv_wino4_kernel sits: scratch loads / stores per region between the `; W4MARK`
comments the kernel source emits (a spill reload inside the chunk step or the fast epilogue is a `vmcnt(0)` in front of its use: it drains
the hand-counted loads and waits for the previous pass's stores).   python tools/asm_scratch_map.py kernel.s"""
import re
import sys
lines = open(sys.argv[1]).read().split('\n')
starts = [i for i, l in enumerate(lines) if re.match(r'^_Z\w*conv_wino4_kernel\w*:', l)]
bad = 0
for st in starts:
    en = next(i for i in range(st, len(lines)) if lines[i].strip() == 's_endpgm')
    region, counts, order = 'prologue', {}, []
    for i in range(st, en):
        t = lines[i].strip()
        m = re.search(r'W4MARK (\w+)', t)
        if m:
            region = m.group(1)
            continue
        if region not in counts:
            counts[region] = [0, 0, 0]
            order.append(region)
        if t.startswith('scratch_load'):
            counts[region][0] += 1
        elif t.startswith('scratch_store'):
            counts[region][1] += 1
        elif t and not t.startswith((';', '.')) and not t.endswith(':'):
            counts[region][2] += 1
    name = lines[st].split(':')[0]
    print(name)
    for r in order:
        ld, stn, n = counts[r]
        print(f'   after mark {r:16s}: {n:6d} instructions, {ld:3d} scratch loads, {stn:3d} scratch stores')
    hot = sum(counts[r][0] + counts[r][1] for r in counts if r in ('step_begin', 'fast_begin'))
    bad += hot
print('scratch operations inside the chunk step or the fast epilogue:', bad)
sys.exit(1 if bad else 0)
