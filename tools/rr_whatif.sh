#!/bin/bash
# Timing-only what-if builds of the chain kernels (WRONG results; built by hand into tools/micro/build/ with the recipe of
# tools/rr_bisect.sh from the PATCHED source -- tools/micro/experiment_source.sh: -DELO_RR_WHATIF_HALF_READS, -DELO_RR_BARRIER=3 --): HALF_READS = every step reads two of its four W fragments from the ring, NO_BARRIER = the superstep
# barrier is only its s_waitcnt (the eight waves run free), BOTH.  Batch 8, fp16 storage: throughput at 8 lanes and the
# chain kernels' durations at one lane.
OUT=$(realpath -m "${1:-gpurun_out/rr_whatif}"); REPO=$(pwd); mkdir -p $OUT
for v in shipped HALF_READS NO_BARRIER BOTH; do
  if [ $v = shipped ]; then unset ELO_LIB_PATH; else export ELO_LIB_PATH=$REPO/tools/micro/build/libelo_whatif_$v.so; fi
  echo "== $v: $(python bench.py --no-legs --steps 100 --warmup 16 --batch 8 --features f16 --check-every 0 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], 'pairs/s')")"
  ( cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_$v -o t -- python $REPO/bench.py --batch 8 --features f16 --no-legs --steps 40 --warmup 16 --lanes 1 --check-every 0 > /dev/null 2>&1 )
  python - "$(find $OUT/t_$v -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if '_rr_kernel' in r['Name']]
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs'])): print("   %6.1f us avg x%-4s %s" % (float(r['AverageNs'])/1e3, r['Calls'], r['Name'][:70]))
PY
done
find $OUT -name "*.db" -delete; find $OUT -size +2M -delete
