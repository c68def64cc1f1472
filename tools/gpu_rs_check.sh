#!/bin/bash
# one gpurun call: debug dump of the row-stationary pre-filter, the correlation tests, timing of the correlation call alone
# (synthetic maps) and inside the benchmark step
mkdir -p gpurun_out
(timeout 300 env MREFSR_HIP_LIB=mrefsr_amd/lib_dbg/libmrefsr_hip.so python tools/corr_rs_debug.py) > gpurun_out/rs_debug.log 2>&1; echo "debug rc=$?"; tail -7 gpurun_out/rs_debug.log
(timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "corr or prefilter") > gpurun_out/rs_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/rs_tests.log
timeout 200 python tools/corr_time.py 3 2>&1 | grep -v amdgpu.ids
(timeout 300 python tools/corr_bench_step.py) 2>&1 | grep -E "corr ms|flagged"
