#!/bin/bash
# the round-6 kernels' bit tests repeated (hand-counted waits and wave-private LDS protocols: a race would show as an intermittent failure)
n=${1:-15}
fail=0
for i in $(seq 1 $n); do
  python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "conv1x1 or rescore or image_to_nhwc4 or channels_last_variants or offsets or dcn_chunk_outer or block_geometries" 2>&1 | tail -1 | grep -q " passed" || { fail=$((fail + 1)); echo "run $i FAILED"; }
done
echo "stress: $n runs, $fail failed"
