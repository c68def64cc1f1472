#!/bin/bash
# re-scoring kernel A/B: bit test, then the correlation call inside the benchmark step under both kernels
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "rescore or corr_prefilter_block" 2>&1 | tail -12 > gpurun_out/rs_test.txt
cat gpurun_out/rs_test.txt
for f in quad lds; do
  MREFSR_CORR_RESCORE=$f python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train-step 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$f', d['value'], d['ms_per_step'], 'corr call', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'], 'clock', d['clock_mhz']['median'])"
done
