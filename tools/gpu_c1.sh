#!/bin/bash
# conv1x1 kernel: bit test + A/B timing
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "conv1x1" 2>&1 | tail -15 > gpurun_out/c1_test.txt
python tools/conv1x1_ab.py > gpurun_out/c1_ab.txt 2>&1
cat gpurun_out/c1_test.txt gpurun_out/c1_ab.txt
