#!/bin/bash
# PMC passes around the stand-alone correlation call (tools/corr_time.py): SQ occupancy / stall / MFMA / LDS counters
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/pmc_corr
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" \
         "SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format rocpd -d $O/p$i -o b -- python3 $R/tools/corr_time.py 2 > $O/p$i.log 2>&1
    tail -1 $O/p$i.log
done
cd $R
python3 tools/pmc_summary.py $(ls $O/p*/*.db $O/p*/*/*.db 2>/dev/null) > gpurun_out/pmc_corr_rs.json
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/pmc_corr_rs.json'))['kernels']
for k, v in d.items():
    if k.startswith('corr_'):
        print(k, {c: round(x, 1) if isinstance(x, float) else x for c, x in v.items() if c.isupper() or c == 'mfma_busy'})
PY
rm -rf $O/p1 $O/p2 $O/p3
