MREFSR_HIP_LIB=mrefsr_amd/lib_wstamp/libmrefsr_hip.so python tools/conv_wino4_stamp.py > gpurun_out/w4_stamps.txt 2>&1
python tools/conv_wino4_abl.py base > gpurun_out/w4_abl.txt 2>&1
for v in 1 2 3 4 5 6; do MREFSR_HIP_LIB=mrefsr_amd/lib_abl$v/libmrefsr_hip.so python tools/conv_wino4_abl.py abl$v >> gpurun_out/w4_abl.txt 2>&1; done
cat gpurun_out/w4_stamps.txt gpurun_out/w4_abl.txt | grep -v amdgpu.ids
