MREFSR_HIP_LIB=mrefsr_amd/lib_h/libmrefsr_hip.so python tools/conv_wino4_abl.py base > gpurun_out/w4_abl.txt 2>&1
for v in 1 2 3 4 5 7 8; do MREFSR_HIP_LIB=mrefsr_amd/lib_abl$v/libmrefsr_hip.so python tools/conv_wino4_abl.py abl$v >> gpurun_out/w4_abl.txt 2>&1; done
MREFSR_HIP_LIB=mrefsr_amd/lib_h/libmrefsr_hip.so python tools/conv_wino4_abl.py base >> gpurun_out/w4_abl.txt 2>&1
grep -v amdgpu.ids gpurun_out/w4_abl.txt
