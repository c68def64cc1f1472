#!/bin/bash
# PMC passes over tools/conv_one.py (one shape, both convolution kernels):  bash tools/conv_pmc.sh N H W CIN COUT -> gpurun_out/conv_pmc_<shape>.json
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/conv_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE"; do
    i=$((i + 1))
    rm -rf $O/p$i
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format rocpd -d $O/p$i -o b -- python3 $R/tools/conv_one.py "$@" > $O/p$i.log 2>&1
done
cd $R
python3 tools/pmc_summary.py $(ls $O/p*/*.db $O/p*/*/*.db 2>/dev/null) > gpurun_out/conv_pmc_$(echo "$@" | tr ' ' '_').json
rm -rf $O/p1 $O/p2 $O/p3 $O/p4
