#!/bin/bash
# Round-5 evidence in one go (GPU box, repo root): bash tools/capture_round5.sh -> gpurun_out/r05/final/
#   the driver's bench line, kernel stats at 8 lanes / 1 lane, the launch sequence of one forward, configs[2] (batch 8, fp16 features)
#   trace + counter passes, the SQ instruction budget of a batch-8 forward, the parity statistic with its ATTRIBUTED flips, training.
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r05/final; mkdir -p $OUT
python bench.py --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench_stderr.log
python tools/parity_flips.py > $OUT/parity_flips.txt 2>/dev/null
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_8lanes -o t -- python $REPO/bench.py --steps 200 --warmup 16 --no-legs > $OUT/trace_8lanes.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_1lane -o t -- python $REPO/bench.py --steps 200 --warmup 16 --no-legs --lanes 1 > $OUT/trace_1lane.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/seq -o t -- python $REPO/bench.py --lanes 1 --steps 3 --warmup 1 --no-legs --pool 1 > $OUT/seq.log 2>&1
cd $REPO
python tools/forward_sequence.py $(find $OUT/seq -name "*kernel_trace.csv" | head -1) > $OUT/forward_sequence_1lane.txt 2>&1
bash tools/c3_capture.sh $OUT/c3 > $OUT/c3.log 2>&1
bash tools/sq_forward_budget.sh $OUT/sq_budget > $OUT/sq_forward_budget.txt 2>&1
python tools/train_step_time.py > $OUT/training.txt 2>&1
python tools/train_kernel_stats.py 8 >> $OUT/training.txt 2>&1
find $OUT -name "*.db" -delete; find $OUT -size +3M -delete
ls $OUT
