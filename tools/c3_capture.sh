#!/bin/bash
# BASELINE configs[2] under rocprofv3: the batch-8, fp16-feature-storage forward that bench.py's `batch8.f16_features`
# leg times (`bench.py --batch 8 --features f16 --no-legs`), in four separate passes -- kernel trace at 8 lanes (the
# leg's own setting), kernel trace at 1 lane (clean per-kernel durations: under the profiler the lanes serialise),
# FETCH_SIZE and WRITE_SIZE at 1 lane (counter passes never share a run with a trace domain).
#   usage (on the GPU box, from the repo root):  bash tools/c3_capture.sh gpurun_out/r02_c3
set -u
OUT=$(realpath -m "${1:-gpurun_out/r02_c3}"); REPO=$(pwd)
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
base="python $REPO/bench.py --batch 8 --features f16 --no-legs --steps 40 --warmup 16"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_8lanes" -o t -- $base --lanes 8 > "$OUT/trace_8lanes.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_1lane" -o t -- $base --lanes 1 > "$OUT/trace_1lane.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch_1lane" -o f -- $base --lanes 1 > "$OUT/fetch_1lane.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write_1lane" -o w -- $base --lanes 1 > "$OUT/write_1lane.log" 2>&1
cd "$REPO"; python tools/c3_summary.py "$OUT" "$OUT/summary.json"
