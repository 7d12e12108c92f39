#!/bin/bash
# Where the SIMD time of a batch-8 fp16 forward goes, kernel by kernel, in INSTRUCTIONS (one rocprofv3 --pmc pass over a
# one-lane bench.py run): per forward, waves, wave-cycles, VALU / MFMA / SALU / LDS instructions, and an issue-cycle estimate
# (4 cycles per VALU, 16 per MFMA, 1 per SALU: per-SIMD pipes) -- with eight forwards in flight throughput follows this
# column, not the one-lane durations (profiles/r04_rr_whatif.txt).
#   bash tools/sq_forward_budget.sh <outdir> [bench.py arguments]
set -u
OUT=$(realpath -m "$1"); REPO=$(pwd); shift 1
STEPS=6
ARGS="${*:---batch 8 --features f16 --no-legs --steps $STEPS --warmup 16 --lanes 1 --check-every 0}"
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d "$OUT/g" -o c -- python $REPO/bench.py $ARGS > "$OUT/g.log" 2>&1
cd "$REPO"
python - "$OUT" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(out + "/g/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"elo::\(anonymous namespace\)::|void ", "", r["Kernel_Name"])[:60] + " g" + r.get("Grid_Size", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": calls[k] += 1
rows = []
for k, c in agg.items():
    valu_only = c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]
    issue = 4 * valu_only + 16 * c["SQ_INSTS_MFMA"] + c["SQ_INSTS_SALU"]
    rows.append((issue, k, c, calls[k], valu_only))
tot = sum(r[0] for r in rows)
print("%-75s %6s %9s %9s %9s %9s %8s %7s" % ("kernel (grid)", "calls", "waves", "VALU", "MFMA", "SALU", "issueMcy", "share"))
for issue, k, c, n, vo in sorted(rows, reverse=True)[:40]:
    print("%-75s %6d %9.0f %9.0f %9.0f %9.0f %8.2f %6.1f%%" % (k, n, c["SQ_WAVES"] / n, vo / n, c["SQ_INSTS_MFMA"] / n, c["SQ_INSTS_SALU"] / n, issue / n / 1e6, 100 * issue / tot))
PY
find "$OUT" -name "*.db" -delete; find "$OUT" -size +3M -delete
