python tools/conv_wino4_check.py > gpurun_out/w4_check_sched.txt 2>&1
MREFSR_HIP_LIB=mrefsr_amd/lib_w4u/libmrefsr_hip.so python tools/conv_wino4_check.py > gpurun_out/w4_check_unsched.txt 2>&1
tail -32 gpurun_out/w4_check_sched.txt; tail -16 gpurun_out/w4_check_unsched.txt
