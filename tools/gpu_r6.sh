#!/bin/bash
# round-6 evidence in one GPU call (tools/gpu_full6.sh runs the whole GPU suite and the counter passes in front of it): bench line (with the CPU baseline and the training-step figure), kernel statistics, convolution layers,
# the other BASELINE configs' bench lines, DCN and correlation A/B.  Everything lands under gpurun_out/r6/ (copied to profiles/r6_* by hand).
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/r6
cd $R
python bench.py > gpurun_out/r6/bench_n1.json 2> gpurun_out/r6/bench_n1.err; tail -c 200 gpurun_out/r6/bench_n1.json; echo
bash tools/bench_kstats.sh r6 > /dev/null 2>&1; cp gpurun_out/kstats_r6.txt gpurun_out/r6/bench_kernel_stats.txt; head -6 gpurun_out/r6/bench_kernel_stats.txt | cut -c1-150
python3 tools/conv_layers.py > gpurun_out/r6/conv_layers.txt 2>/dev/null; head -4 gpurun_out/r6/conv_layers.txt | cut -c1-120
python bench.py --batch 1 --refs 1 --lr 40 --steps 50 --warmup 10 --no-train-step > gpurun_out/r6/bench_n1_config0.json 2>/dev/null; cut -c1-200 gpurun_out/r6/bench_n1_config0.json
python bench.py --dtype bf16 --batch 1 --refs 10 --lr 320 --steps 5 --warmup 2 --no-cpu-baseline --no-train-step > gpurun_out/r6/bench_n1_config4_bf16_storage.json 2>/dev/null; cut -c1-200 gpurun_out/r6/bench_n1_config4_bf16_storage.json
python bench.py --mode train --batch 4 --lr 40 --steps 20 --warmup 6 --no-cpu-baseline > gpurun_out/r6/bench_train_b4_lr40.json 2>/dev/null; cut -c1-200 gpurun_out/r6/bench_train_b4_lr40.json
python tools/dcn_ab.py 40 > gpurun_out/r6/dcn_ab.txt 2>&1; tail -8 gpurun_out/r6/dcn_ab.txt
python tools/corr_time.py 4 2>/dev/null | tail -1 > gpurun_out/r6/corr_time.txt; MREFSR_CORR_W=4 python tools/corr_time.py 4 2>/dev/null | tail -1 >> gpurun_out/r6/corr_time.txt; cat gpurun_out/r6/corr_time.txt
ls gpurun_out/r6
