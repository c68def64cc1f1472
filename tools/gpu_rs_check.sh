#!/bin/bash
# one gpurun call: debug dump of the row-stationary pre-filter, the correlation tests, and the correlation call timed
# inside the benchmark step for the new and the previous default kernel
mkdir -p gpurun_out
(timeout 300 env MREFSR_HIP_LIB=mrefsr_amd/lib_dbg/libmrefsr_hip.so python tools/corr_rs_debug.py) > gpurun_out/rs_debug.log 2>&1; echo "debug rc=$?"; tail -30 gpurun_out/rs_debug.log
(timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "corr or prefilter") > gpurun_out/rs_tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/rs_tests.log
(timeout 300 python tools/corr_bench_step.py) > gpurun_out/rs_bench.log 2>&1; tail -2 gpurun_out/rs_bench.log
(timeout 300 env MREFSR_CORR_PREFILTER_WS16=1 python tools/corr_bench_step.py) > gpurun_out/ws16_bench.log 2>&1; tail -2 gpurun_out/ws16_bench.log
