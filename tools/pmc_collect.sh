#!/bin/bash
# Three rocprofv3 passes per (kernel, batch) of tools/roofline_micro.py -- kernel trace, FETCH_SIZE, WRITE_SIZE, each
# in its own run (never --pmc together with a trace domain) -- into $1 (default gpurun_out/pmc3), then
# tools/pmc_summary.py folds them.   usage (on the GPU box):  bash tools/pmc_collect.sh gpurun_out/pmc3
# ELO_PMC_GRID=32x256 (read by roofline_micro.py): the l0 grid of a 128x2048 scan instead of 16x225; ELO_PMC_SPECS: "kernel[_f16]:batch ..."
set -u
OUT=$(realpath -m "${1:-gpurun_out/pmc3}"); REPO=$(pwd)
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
SPECS=${ELO_PMC_SPECS:-"cv1:1 cv1:8 encode1:8 pool:8 encode2:8 pool2:8 encode1:64 pool:64 encode2:64 pool2:64 select32:1 select32:8 select32_l2:8 random16:1 random16:8"}
for spec in $SPECS; do
    k=${spec%%:*}; b=${spec##*:}; tag=${k}_b${b}; half=""
    case "$k" in *_f16) k=${k%_f16}; half="--half";; esac      # e.g. encode1_f16:64
    cmd="python $REPO/tools/roofline_micro.py --kernel $k --batch $b --reps 25 $half ${ELO_PMC_COLD:+--cold}"      # ELO_PMC_COLD=1: ring of tensor sets (HBM-cold launches)
    $cmd > "$OUT/$tag.info" 2>/dev/null
    rocprofv3 --kernel-trace --output-format csv -d "$OUT/$tag.trace" -o t -- $cmd > /dev/null 2>&1
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/$tag.fetch" -o f -- $cmd > /dev/null 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/$tag.write" -o w -- $cmd > /dev/null 2>&1
done
cd "$REPO"; python tools/pmc_summary.py "$OUT" "$OUT/summary.json"
