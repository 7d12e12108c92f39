#!/bin/bash
# SQ counter passes over one kernel of tools/roofline_micro.py (counters only, one group per run).
#   bash tools/sq_counters.sh <kernel> <batch> <outdir> [extra roofline_micro.py arguments, e.g. --pregrouped]
set -u
K=$1; B=$2; OUT=$(realpath -m "$3"); REPO=$(pwd); shift 3; EXTRA="$*"
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
cmd="python $REPO/tools/roofline_micro.py --kernel $K --batch $B --reps 12 $EXTRA"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_CYCLES" \
           "MfmaUtil LdsUtil MeanOccupancyPerActiveCU" ; do
    i=$((i+1))
    rocprofv3 --pmc $grp --output-format csv -d "$OUT/g$i" -o c -- $cmd > "$OUT/g$i.log" 2>&1
done
cd "$REPO"
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    if "elo" not in k: continue
    print(k)
    for c, v in sorted(cs.items()):
        v = v[3:] or v
        print("   %-28s %14.1f" % (c, sum(v) / len(v)))
PY
