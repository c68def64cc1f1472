#!/bin/bash
# round-5 evidence in one GPU call: test suite, bench line, kernel statistics, convolution layers, four-wave Winograd kernel (check, stamps,
# counters), per-step PMC passes.  Everything lands under gpurun_out/r5/ (copied to profiles/r5_* by hand).
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/r5
cd $R
python -m pytest tests -x -q -m gpu > gpurun_out/r5/t_gpu.txt 2>&1; grep -E "passed|failed" gpurun_out/r5/t_gpu.txt | tail -2
python bench.py > gpurun_out/r5/bench_n1.json 2> gpurun_out/r5/bench_n1.err; tail -c 300 gpurun_out/r5/bench_n1.json; echo
bash tools/bench_kstats.sh r5 > /dev/null 2>&1; cp gpurun_out/kstats_r5.txt gpurun_out/r5/bench_kernel_stats.txt; head -8 gpurun_out/r5/bench_kernel_stats.txt | cut -c1-150
python3 tools/conv_layers.py > gpurun_out/r5/conv_layers_default.txt 2>/dev/null
MREFSR_WINO_WAVES=8 python3 tools/conv_layers.py > gpurun_out/r5/conv_layers_eight.txt 2>/dev/null
head -6 gpurun_out/r5/conv_layers_default.txt | cut -c1-160
python3 tools/conv_wino4_check.py > gpurun_out/r5/conv_wino4_check.txt 2>&1; tail -2 gpurun_out/r5/conv_wino4_check.txt
MREFSR_HIP_LIB=mrefsr_amd/lib_wstamp/libmrefsr_hip.so python3 tools/conv_wino4_stamp.py > gpurun_out/r5/conv_wino4_stamps.txt 2>&1; tail -4 gpurun_out/r5/conv_wino4_stamps.txt | cut -c1-200
bash tools/conv_wino_pmc.sh 8 320 320 256 256 > /dev/null 2>&1
bash tools/conv_wino_pmc.sh 8 640 640 64 64 5 res > /dev/null 2>&1
bash tools/conv_wino_pmc.sh 8 160 160 512 512 > /dev/null 2>&1
cp gpurun_out/conv_wino_pmc_*.json gpurun_out/r5/ 2>/dev/null
bash tools/pmc_refresh.sh 2>&1 | tail -2
cp gpurun_out/pmc_per_step.json gpurun_out/pmc_corr.json gpurun_out/pmc_dcn.json gpurun_out/r5/ 2>/dev/null
ls gpurun_out/r5
