python tools/conv_wino4_check.py > gpurun_out/w4_check2.txt 2>&1
MREFSR_HIP_LIB=mrefsr_amd/lib_wstamp/libmrefsr_hip.so python tools/conv_wino4_stamp.py > gpurun_out/w4_stamps2.txt 2>&1
grep -v amdgpu.ids gpurun_out/w4_check2.txt | tail -30; grep -v amdgpu.ids gpurun_out/w4_stamps2.txt
