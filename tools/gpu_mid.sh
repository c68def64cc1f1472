#!/bin/bash
# quick check of the latest changes: their tests, then a short benchmark line
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "channels_last_variants or mrattn" 2>&1 | tail -4
python -m pytest tests/test_archs_gpu.py tests/test_configs_gpu.py -m gpu -x -q 2>&1 | tail -4
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train-step 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], 'corr call', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'], 'conv', d['roofline_conv']['ms_per_step'], 'attn', d['roofline_attn']['ms_per_step'], 'clock', d['clock_mhz']['median'])"
