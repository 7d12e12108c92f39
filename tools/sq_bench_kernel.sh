#!/bin/bash
# SQ / memory-pipe counter passes over the kernels of a bench.py run (counters only, one group per run), summarised for the
# kernels whose name contains <pattern>:
#   bash tools/sq_bench_kernel.sh <pattern> <outdir> [bench.py arguments; default: batch 8, fp16 features, one lane]
set -u
PAT=$1; OUT=$(realpath -m "$2"); REPO=$(pwd); shift 2
ARGS="${*:---batch 8 --features f16 --no-legs --steps 6 --warmup 16 --lanes 1}"
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_TA_BUSY_sum TCP_TA_TCP_STATE_READ_sum" \
           "MeanOccupancyPerActiveCU GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" ; do
    i=$((i+1))
    rocprofv3 --pmc $grp --output-format csv -d "$OUT/g$i" -o c -- python $REPO/bench.py $ARGS > "$OUT/g$i.log" 2>&1
done
cd "$REPO"
python - "$OUT" "$PAT" <<'PY'
import csv, glob, sys, collections
out, pat = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:90], r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(agg.items()):
    print(k)
    for c, v in sorted(cs.items()):
        v = v[2:] or v
        print("   %-32s %16.1f" % (c, sum(v) / len(v)))
PY
find "$OUT" -name "*.db" -delete
