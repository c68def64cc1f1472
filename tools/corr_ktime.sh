#!/bin/bash
# kernel-trace time of the correlation call's kernels for one library build:  bash tools/corr_ktime.sh [lib.so]  (prints per-kernel averages)
R=${GRAFT_REPO_ROOT:-$PWD}
[ -n "${1:-}" ] && export MREFSR_HIP_LIB=$1
O=$R/gpurun_out/corr_kt
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o k -- python3 $R/tools/corr_time.py 2 > $O/log.txt 2>&1
cd $R
python3 - <<PY
import csv, glob
f = glob.glob('$O/**/*kernel_stats.csv', recursive=True)
for row in csv.DictReader(open(f[0])):
    if 'corr' in row['Name'] or 'pixnorm' in row['Name']:
        print(f"{row['Name'][:70]:70s} calls {row['Calls']:>4s} avg_us {float(row['AverageNs'])/1e3:10.1f}")
PY
