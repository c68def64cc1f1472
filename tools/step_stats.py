#!/usr/bin/env python3
"""Per-kernel table of ONE steady-state step from a rocprofv3 rocpd database (kernel-trace): the dispatches between the last
two launches of a marker kernel (default: the correlation pre-filter, once per step).
    python tools/step_stats.py gpurun_out/ptrain/t_results.db [marker] [--top 40]"""
import re
import sqlite3
import sys
from collections import defaultdict

db = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith('--') else 'corr_prefilter_rs16'
top = int(sys.argv[sys.argv.index('--top') + 1]) if '--top' in sys.argv else 40
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
idx = [i for i, r in enumerate(rows) if marker in r[0]]
seg = rows[idx[-2]:idx[-1]]
t = defaultdict(lambda: [0, 0])
for n, s, e in seg:
    n = re.sub(r'\(.*', '', n)[:120]
    t[n][0] += e - s
    t[n][1] += 1
busy = sum(v[0] for v in t.values())
print(f'# {db}: one step = {len(seg)} launches, {(seg[-1][2] - seg[0][1]) / 1e6:.2f} ms wall, {busy / 1e6:.2f} ms of kernel time')
print(f'{"ms":>9} {"calls":>6}  name')
for n, v in sorted(t.items(), key=lambda x: -x[1][0])[:top]:
    print(f'{v[0] / 1e6:9.3f} {v[1]:6d}  {n}')
