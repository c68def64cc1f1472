#!/bin/bash
# L1-side counters of the three DCN launches of one benchmark step:  bash tools/dcn_pmc.sh -> gpurun_out/dcn_l1_pmc.json
# (separate rocprofv3 --pmc passes, kernel trace only; TCP/TA sums are over all CUs)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/dcn_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for c in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum" "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "GRBM_GUI_ACTIVE TD_TD_BUSY_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES"; do
    i=$((i + 1))
    rm -rf $O/p$i
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format rocpd -d $O/p$i -o b -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-train-step > $O/p$i.log 2>&1
done
cd $R
python3 tools/pmc_summary.py $(ls $O/p*/*.db $O/p*/*/*.db 2>/dev/null) > $O/all.json
python3 - <<PY
import json
d = json.load(open('$O/all.json'))['kernels']
out = {k: {c: v for c, v in d[k].items() if c.split('_')[0] in ('TCP', 'TA', 'TD', 'SQ', 'GRBM')} for k in d if k.startswith('dcn_fwd')}
json.dump(dict(unit='per launch (one benchmark step = one launch of each kernel: 40 images at 160^2 x 256 / 320^2 x 128 / 640^2 x 64); '
               'GRBM_GUI_ACTIVE summed over the 8 XCDs (one pass)', kernels=out,
               command='bash tools/dcn_pmc.sh (five rocprofv3 --pmc passes around bench.py --steps 1 --warmup 1)'), open('gpurun_out/dcn_l1_pmc.json', 'w'), indent=1)
PY
rm -rf $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 $O/p6
