#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into the per-kernel table that
`--stats` prints: name, calls, total / average / min / max duration, % of GPU kernel time.
    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--top 40] > profiles/x_kernel_stats.txt"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    top = int(sys.argv[sys.argv.index('--top') + 1]) if '--top' in sys.argv else 45
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = 'name' if 'name' in cols else 'kernel_name'
    dur = 'duration' if 'duration' in cols else '(end - start)'
    where, note = '', ''
    if '--window' in sys.argv:
        # steady-state window: from the END of the i-th launch of a marker kernel to the END of the
        # j-th (whole steps, phase-shifted), e.g. --window corr_top1_kernel 3 5
        k = sys.argv.index('--window')
        marker, i, j = sys.argv[k + 1], int(sys.argv[k + 2]), int(sys.argv[k + 3])
        ends = [r[0] for r in c.execute(f"select end from kernels where {name_col} like ? order by start", (f'%{marker}%',))]
        t0, t1 = ends[i - 1], ends[j - 1]
        where = f'where start >= {t0} and end <= {t1}'
        note = f' [window: end of {marker} #{i} .. end of #{j} = {j-i} steps, {(t1-t0)/1e6:.2f} ms wall]'
    rows = c.execute(f"select {name_col}, count(*), sum({dur}), avg({dur}), min({dur}), max({dur}) from kernels {where} "
                     f"group by {name_col} order by sum({dur}) desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f'# {db}: {sum(r[1] for r in rows)} kernel dispatches, {total/1e6:.2f} ms total GPU kernel time{note}')
    print(f'{"calls":>7} {"total_ms":>10} {"avg_us":>11} {"min_us":>10} {"max_us":>11} {"pct":>6}  name')
    for n, cnt, tot, avg, mn, mx in rows[:top]:
        print(f'{cnt:7d} {tot/1e6:10.3f} {avg/1e3:11.2f} {mn/1e3:10.2f} {mx/1e3:11.2f} {100*tot/total:6.2f}  {n[:150]}')
    rest = rows[top:]
    if rest:
        print(f'{sum(r[1] for r in rest):7d} {sum(r[2] for r in rest)/1e6:10.3f} {"":>11} {"":>10} {"":>11} '
              f'{100*sum(r[2] for r in rest)/total:6.2f}  ({len(rest)} more kernels)')


if __name__ == '__main__':
    main()
