#!/bin/bash
# round-2 evidence: bench line, kernel statistics of steady-state steps (configs[1], configs[4], configs[2] shapes), PMC passes
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
cd $R
python bench.py > gpurun_out/prof/bench_n1.json 2> gpurun_out/prof/bench_n1.err; tail -c 600 gpurun_out/prof/bench_n1.json; echo
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/k1 -o b -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof/k1.log 2>&1
python3 tools/rocpd_stats.py gpurun_out/prof/k1/b_results.db --window corr_prefilter_rs16 3 6 --top 40 > gpurun_out/prof/bench_kernel_stats_steady.txt; head -12 gpurun_out/prof/bench_kernel_stats_steady.txt | cut -c1-150
python bench.py --dtype bf16 --batch 1 --refs 10 --lr 320 --no-cpu-baseline > gpurun_out/prof/bench_config4.json 2>/dev/null
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/k4 -o b -- python3 bench.py --dtype bf16 --batch 1 --refs 10 --lr 320 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/prof/k4.log 2>&1
python3 tools/rocpd_stats.py gpurun_out/prof/k4/b_results.db --window corr_prefilter_rs16 3 5 --top 25 > gpurun_out/prof/config4_kernel_stats_steady.txt; head -10 gpurun_out/prof/config4_kernel_stats_steady.txt | cut -c1-150
python bench.py --mode train --batch 4 --lr 40 --steps 10 --warmup 6 --no-cpu-baseline > gpurun_out/prof/bench_train_b4_lr40.json 2>/dev/null; cut -c1-200 gpurun_out/prof/bench_train_b4_lr40.json
python bench.py --mode train --batch 4 --lr 40 --steps 10 --warmup 6 --no-cpu-baseline > gpurun_out/prof/bench_train_b4_lr40_miopen_nchw.json 2>/dev/null; cut -c100-200 gpurun_out/prof/bench_train_b4_lr40_miopen_nchw.json
bash tools/train_profile.sh
python3 tools/step_stats.py gpurun_out/ptrain/t_results.db --top 40 > gpurun_out/prof/train_step_kernel_stats.txt; head -8 gpurun_out/prof/train_step_kernel_stats.txt | cut -c1-150
python3 tools/conv_layers.py > gpurun_out/prof/conv_layers.txt 2>/dev/null; head -4 gpurun_out/prof/conv_layers.txt
bash tools/pmc_refresh.sh
cp gpurun_out/pmc_per_step.json gpurun_out/pmc_corr.json gpurun_out/pmc_dcn.json gpurun_out/prof/ 2>/dev/null
rm -rf gpurun_out/prof/k1 gpurun_out/prof/k4
