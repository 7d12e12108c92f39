#!/bin/bash
# NEEDS tools/micro/patches/r06_rejected_forms.patch applied to the tree (the form measured here was taken out of csrc/ in round 6).
# Per-kernel evidence for "layer 0 through the gather" (round 5): the batch-8, fp16-storage forward (bench.py --batch 8 --features
# f16 --no-legs --lanes 1) under rocprofv3 --kernel-trace, and FETCH_SIZE / WRITE_SIZE in passes of their own, for three settings
# of ELO_LAYER0_PRE: none, the set-upconvs, every grouped operator.   usage (GPU box): bash tools/layer0_capture.sh gpurun_out/r05/layer0
set -u
OUT=$(realpath -m "${1:-gpurun_out/r05/layer0}"); REPO=$(pwd)
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
base="python $REPO/bench.py --batch 8 --features f16 --no-legs --steps 40 --warmup 16 --lanes 1"
for tag in none:0 setconv:setconv all:setconv,cv2,cv1:0; do
    name=${tag%%:*}; spec=${tag#*:}
    ELO_LAYER0_PRE=$spec rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name/trace_1lane" -o t -- $base > "$OUT/$name.trace.log" 2>&1
    ELO_LAYER0_PRE=$spec rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/$name/fetch_1lane" -o f -- $base > "$OUT/$name.fetch.log" 2>&1
    ELO_LAYER0_PRE=$spec rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/$name/write_1lane" -o w -- $base > "$OUT/$name.write.log" 2>&1
    mkdir -p "$OUT/$name/trace_8lanes"
    (cd "$REPO"; python tools/c3_summary.py "$OUT/$name" "$OUT/$name.summary.json" > "$OUT/$name.summary.txt" 2>&1)
done
find "$OUT" -name "*.db" -delete; find "$OUT" -size +3M -delete
