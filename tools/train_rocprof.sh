#!/bin/bash
# rocprofv3 evidence for the training kernels (GPU box, repo root): bash tools/train_rocprof.sh -> gpurun_out/r06/train_prof/
#   kernel trace + stats of captured training steps (tools/train_step_time.py 8), and FETCH_SIZE / WRITE_SIZE of elo_dense_rows /
#   weight_grad on the 172 800 x 128 -> 128 layer, each counter in its own pass (never --pmc with a trace domain).
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r06/train_prof; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
(cd $REPO && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/step -o t -- python tools/train_step_time.py 8 > $OUT/step.log 2>&1)
for mode in fwd dx wgrad; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/${mode}_$c -o p -- python $REPO/tools/dense_rows_pmc.py 172800 128 128 $mode > /dev/null 2>&1
  done
done
cd $REPO
python - <<'PY'
import csv, glob, collections
out = "gpurun_out/r06/train_prof"
st = glob.glob(out + "/step/**/*kernel_stats.csv", recursive=True)
if st:
    rows = list(csv.DictReader(open(st[0])))
    print("# rocprofv3 --kernel-trace --stats -- python tools/train_step_time.py 8   (eager + capture warm-up + 40 replayed steps; top 25 by total time)")
    for r in rows[:25]:
        print("%-90s calls %6s  total %10.1f us  avg %8.2f us  %5s %%" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, r["Percentage"]))
print("# HBM traffic per launch of the 172800 x 128 -> 128 layer (KB; FETCH_SIZE / WRITE_SIZE, separate passes; algorithmic: 88.5 MB per (rows x 128) tensor)")
for mode, kern in (("fwd", "dense_rows"), ("dx", "dense_rows"), ("wgrad", "weight_grad")):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = []
        for f in glob.glob(out + "/%s_%s/**/*counter_collection.csv" % (mode, c), recursive=True):
            for r in csv.DictReader(open(f)):
                if kern in r["Kernel_Name"] and r["Counter_Name"] == c: vals.append(float(r["Counter_Value"]))
        if vals: print("%-6s %-11s %s launches, mean %.0f KB" % (mode, c, len(vals), sum(vals) / len(vals)))
PY
