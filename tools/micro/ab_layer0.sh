#!/bin/bash
# NEEDS tools/micro/patches/r06_rejected_forms.patch applied to the tree (the form measured here was taken out of csrc/ in round 6).
# A/B of "layer 0 through the gather" (round 5; fused.layer0_pre / ELO_LAYER0_PRE): batch 8 through 8 lanes, fp16 and fp32 feature
# storage, with the first layer of (a) no operator, (b) the set-upconvs, (c) + cost-volume stage 2, (d) + stage 1 from 32 channels,
# (e) + stage 1 at every level commuted with the gather.  Three alternating rounds (the pool's boxes drift by ~1 %).
#   usage (GPU box, repo root): bash tools/ab_layer0.sh [rounds] > gpurun_out/r05/ab_layer0.txt
ROUNDS=${1:-3}
for r in $(seq 1 $ROUNDS); do
  for spec in 0 setconv setconv,cv2 setconv,cv2,cv1:32 setconv,cv2,cv1:0; do
    for feat in f16 f32; do
      v=$(ELO_LAYER0_PRE=$spec python bench.py --batch 8 --features $feat --steps 240 --warmup 16 --no-legs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])")
      echo "round $r  ELO_LAYER0_PRE=$spec  features=$feat  pairs/s, ms/step: $v"
    done
  done
done
