#!/bin/bash
# A/B of where cv1_tile reads its six layer descriptors: at the layer through kernarg_dense() (the product) or from the
# by-value argument block (-DELO_CV1_EARLY_DESCRIPTORS, built as tools/micro/build/libelo_early.so by the recipe of
# tools/rr_bisect.sh).  profiles/r04_ab_descriptors.txt holds a run.
mkdir -p gpurun_out/ab_desc
for r in 1 2 3; do for v in late early; do
  if [ $v = early ]; then export ELO_LIB_PATH=$PWD/tools/micro/build/libelo_early.so; else unset ELO_LIB_PATH; fi
  python bench.py --no-legs --steps 200 --warmup 16 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v b1', d['value'], d['ms_per_step'])"
  python bench.py --no-legs --steps 100 --warmup 16 --batch 8 --features f16 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v b8f16', d['value'], d['ms_per_step'])"
done; done | tee gpurun_out/ab_desc/ab.txt
