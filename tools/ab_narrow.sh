#!/bin/bash
# A/B of the narrow set-conv kernel forms (ELO_SETCONV_NARROW_MFMA=1 matrix cores / 0 VALU) on the two timed configurations,
# and the per-kernel times of a batch-8 fp16 forward at one lane:   bash tools/ab_narrow.sh gpurun_out/ab_narrow
OUT=$(realpath -m "${1:-gpurun_out/ab_narrow}"); REPO=$(pwd); mkdir -p $OUT
val() { python -c "import json,sys; d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])" $1; }
for v in 1 0; do
  ELO_SETCONV_NARROW_MFMA=$v python bench.py --no-legs --steps 200 --warmup 16 > $OUT/b1_$v.json 2>/dev/null; echo "narrow_mfma=$v batch 1:        $(val $OUT/b1_$v.json)"
  ELO_SETCONV_NARROW_MFMA=$v python bench.py --no-legs --steps 100 --warmup 16 --batch 8 --features f16 > $OUT/b8_$v.json 2>/dev/null; echo "narrow_mfma=$v batch 8 fp16:   $(val $OUT/b8_$v.json)"
done
cd /tmp; export TMPDIR=/tmp
for v in 1 0; do
  ELO_SETCONV_NARROW_MFMA=$v rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$v -o t -- python $REPO/bench.py --batch 8 --features f16 --no-legs --steps 40 --warmup 16 --lanes 1 > $OUT/trace_$v.log 2>&1
  f=$(find $OUT/trace_$v -name "*kernel_stats.csv" | head -1)
  echo "--- narrow_mfma=$v, batch 8 fp16, one lane: kernels by total time"; python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:14]:
    print("%6.1f us avg  x%-5s %5.1f%%  %s" % (float(r['AverageNs'])/1e3, r['Calls'], 100*float(r['TotalDurationNs'])/tot, r['Name'][:110]))
PY
done
find $OUT -name "*.db" -delete; find $OUT -size +3M -delete
