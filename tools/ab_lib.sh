#!/bin/bash
# A/B of the shipped library against another build of it (ELO_LIB_PATH), alternating runs: batch 1 and batch 8 fp16.
#   bash tools/ab_lib.sh tools/micro/build/libelo_<variant>.so [rounds]
ALT=$(realpath "$1"); R=${2:-3}
val() { python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])"; }
for r in $(seq $R); do for v in shipped alt; do
  if [ $v = alt ]; then export ELO_LIB_PATH=$ALT; else unset ELO_LIB_PATH; fi
  echo "$v b1     $(python bench.py --no-legs --steps 200 --warmup 16 2>/dev/null | val)"
  echo "$v b8f16  $(python bench.py --no-legs --steps 100 --warmup 16 --batch 8 --features f16 2>/dev/null | val)"
done; done
