#!/bin/bash
# kernel-trace stats of a short benchmark run:  bash tools/bench_kstats.sh TAG [bench args] -> gpurun_out/kstats_TAG.txt (top kernels)
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; shift
O=$R/gpurun_out/kst_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o k -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train-step "$@" > $O/log.txt 2>&1
cd $R
python3 - <<PY > gpurun_out/kstats_$TAG.txt
import csv, glob
f = glob.glob('$O/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'# {len(rows)} kernels, {tot/1e6:.1f} ms of kernel time (4 steps: 1 warm-up + 3)')
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:60]:
    print(f"{float(r['TotalDurationNs'])/1e6:10.2f} ms {r['Calls']:>6s} calls  avg {float(r['AverageNs'])/1e3:10.1f} us  {r['Name'][:110]}")
PY
rm -rf $O/*/  2>/dev/null; ls $O | head -3
