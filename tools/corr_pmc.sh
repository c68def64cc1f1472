#!/bin/bash
# PMC passes over tools/corr_time.py (the correlation call at the benchmark size): bash tools/corr_pmc.sh [tag] -> gpurun_out/corr_pmc_<tag>.json
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-x}
O=$R/gpurun_out/corr_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC"; do
    i=$((i + 1))
    rm -rf $O/p$i
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format rocpd -d $O/p$i -o b -- python3 $R/tools/corr_time.py 2 > $O/p$i.log 2>&1
done
cd $R
python3 tools/pmc_summary.py $(ls $O/p*/*.db $O/p*/*/*.db 2>/dev/null) > gpurun_out/corr_pmc_$TAG.json
rm -rf $O/p1 $O/p2 $O/p3
