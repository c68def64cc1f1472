#!/bin/bash
# the whole GPU suite, then the round's evidence
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/full_gpu_tests.txt 2>&1; tail -5 gpurun_out/full_gpu_tests.txt
bash tools/pmc_refresh.sh > gpurun_out/pmc_refresh.log 2>&1
bash tools/gpu_r6.sh
