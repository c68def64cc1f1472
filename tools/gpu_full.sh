#!/bin/bash
# full GPU validation: pytest -m gpu, smoke, bench (N=1).  usage: bash tools/gpu_full.sh [pytest -k expression]
mkdir -p gpurun_out
if [ -n "$1" ]; then K=(-k "$1"); else K=(); fi
(timeout 3000 python -m pytest tests/ -q -m gpu --durations=12 "${K[@]}") > gpurun_out/gputests.log 2>&1; echo "gpu tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/gputests.log | head -40; grep -E "^[0-9.]+s (call|setup)" gpurun_out/gputests.log | head -12
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()") 2>&1 | tail -1
