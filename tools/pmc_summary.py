#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (rocpd sqlite databases) per kernel:
    python tools/pmc_summary.py pass1.db pass2.db ... [--per-step MARKER_KERNEL] > profiles/x_pmc.json
Counters of every launch of a kernel are summed over the whole run; with --per-step they are divided by the
number of launches of MARKER_KERNEL (one per benchmark step), otherwise by the kernel's own launch count.
Derived: hbm_bytes = 2*FETCH_SIZE*1024... (see below), mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * GRBM_GUI_ACTIVE / 8)."""
import collections
import json
import re
import sqlite3
import sys


def short(name):
    m = re.search(r'(\w+)(<[^>]*>)?\(', name)
    return (m.group(1) + (m.group(2) or '')) if m else name[:60]


def main():
    args = sys.argv[1:]
    marker = None
    if '--per-step' in args:
        i = args.index('--per-step')
        marker = args[i + 1]
        del args[i:i + 2]
    out = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(dict)
    for db in args:
        c = sqlite3.connect(db)
        for k, n, v, cnt in c.execute('select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection '
                                      'group by kernel_name, counter_name'):
            out[short(k)][n] += v
            launches[short(k)][n] = cnt
    steps = None
    if marker:
        steps = max(v for k, d in launches.items() if marker in k for v in d.values())
    res = {}
    for k, d in sorted(out.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', 0)):
        n = {c: (steps or launches[k][c]) for c in d}
        e = {c: d[c] / n[c] for c in d}
        e['launches_per_unit'] = {c: launches[k][c] / n[c] for c in d}.get('GRBM_GUI_ACTIVE', None)
        # gfx950 note of the guide: FETCH_SIZE / WRITE_SIZE count 64-byte... units of 1 KiB?  keep raw + the guide's x2 on FETCH
        if 'FETCH_SIZE' in d:
            e['hbm_read_bytes(FETCH_SIZE*1024*2)'] = e['FETCH_SIZE'] * 1024 * 2
        if 'WRITE_SIZE' in d:
            e['hbm_write_bytes(WRITE_SIZE*1024)'] = e['WRITE_SIZE'] * 1024
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in d and 'GRBM_GUI_ACTIVE' in d and d['GRBM_GUI_ACTIVE'] > 0:
            e['mfma_busy'] = d['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * d['GRBM_GUI_ACTIVE'] / 8.0)
        # The same against the shader engines' own busy time (SQ_BUSY_CYCLES is summed over the 32 shader engines).  In single-kernel
        # profiling runs GRBM_GUI_ACTIVE spans the whole serialized dispatch window -- about 3x the kernel's duration for the 2-ms
        # convolution launches of tools/conv_wino_one.py (GRBM / 8 = 11-12 M cycles against SQ_BUSY / 32 = 3.5-4 M = the launch's
        # measured duration x clock) -- so `mfma_busy` under-reports there; `mfma_busy_sq` is the figure that matches MFMA cycles /
        # (1024 SIMDs x duration x clock): 0.73 for conv_nhwc8_kernel where the in-step PMC of round 3 said 0.66-0.67.
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in d and d.get('SQ_BUSY_CYCLES', 0) > 0:
            e['mfma_busy_sq'] = d['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * d['SQ_BUSY_CYCLES'] / 32.0)
        res[k] = e
    print(json.dumps({'unit': f'per launch of {marker}' if marker else 'per launch of each kernel', 'kernels': res}, indent=1))


if __name__ == '__main__':
    main()
