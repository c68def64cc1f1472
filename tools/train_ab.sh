#!/bin/bash
# training step (B=4, K=5, LR 40) A/B on ONE box: default vs the round-2 launch structure (per-tensor packs, two-node residual
# blocks), eager and replayed as hipGraphs (the graph figure is the GPU side alone)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for rep in 1 2; do
  for g in "" "--graph"; do
    for cfg in "A=1" "MREFSR_TRAIN_PACK_MULTI=0 MREFSR_TRAIN_RESBLOCK=0" $EXTRA_CFG; do
      echo -n "[$cfg] $g: "
      env $cfg python bench.py --mode train --batch 4 --lr 40 --steps 20 --warmup 6 --no-cpu-baseline $g 2>&1 | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
    done
  done
done
