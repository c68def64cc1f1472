#!/bin/bash
# kernel trace of the training step at BASELINE configs[2]'s per-GPU shape (B=4, K=5, LR 40x40) -> gpurun_out/ptrain/
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/ptrain
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format rocpd -d $O -o t -- python3 $R/bench.py --mode train --batch 4 --lr 40 --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/ptrain.log 2>&1
grep '"metric"' $R/gpurun_out/ptrain.log | cut -c1-260
