#!/bin/bash
# PMC pass (SQ / GRBM counters, kernel trace only) around training steps at the configs[2] shape -> gpurun_out/pmc_train.json
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/pmc_train
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format rocpd -d $O/p1 -o b -- python3 $R/bench.py --mode train --batch 4 --lr 40 --steps 2 --warmup 2 --no-cpu-baseline > $O/p1.log 2>&1
cd $R
python3 tools/pmc_summary.py $(ls $O/p1/*.db $O/p1/*/*.db 2>/dev/null) --per-step corr_prefilter_r > gpurun_out/pmc_train.json
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/pmc_train.json'))['kernels']
for k, v in list(d.items())[:8]:
    print(k[:60], {c: (round(x, 3) if isinstance(x, float) else x) for c, x in v.items() if c in ('mfma_busy', 'GRBM_GUI_ACTIVE', 'launches_per_unit')})
PY
rm -rf $O/p1
