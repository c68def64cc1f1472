#!/bin/bash
# PMC passes over tools/conv_wino_one.py (one shape, Winograd and direct kernels):
#   bash tools/conv_wino_pmc.sh N H W CIN COUT [iters] [res] -> gpurun_out/conv_wino_pmc_<shape>.json
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/conv_wino_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "FETCH_SIZE WRITE_SIZE TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    rm -rf $O/p$i
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format rocpd -d $O/p$i -o b -- python3 $R/tools/conv_wino_one.py "$@" > $O/p$i.log 2>&1
done
cd $R
python3 tools/pmc_summary.py $(ls $O/p*/*.db $O/p*/*/*.db 2>/dev/null) > gpurun_out/conv_wino_pmc_$(echo "$@" | tr ' ' '_').json
rm -rf $O/p1 $O/p2 $O/p3 $O/p4
