set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_final; mkdir -p $OUT
python bench.py > $OUT/bench_line.json 2> $OUT/bench_stderr.log
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_8lanes -o t -- python $REPO/bench.py --steps 200 --warmup 16 --no-legs > $OUT/trace_8lanes.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_1lane -o t -- python $REPO/bench.py --steps 200 --warmup 16 --no-legs --lanes 1 > $OUT/trace_1lane.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/seq -o t -- python $REPO/bench.py --lanes 1 --steps 3 --warmup 1 --no-legs --pool 1 > $OUT/seq.log 2>&1
cd $REPO
python tools/forward_sequence.py $(find $OUT/seq -name "*kernel_trace.csv" | head -1) > $OUT/forward_sequence_1lane.txt 2>&1
ELO_PMC_SPECS="cv1:1 cv1:8 cv1_f16:8" bash tools/pmc_collect.sh $OUT/pmc > $OUT/pmc.log 2>&1
bash tools/sq_counters.sh cv1 8 $OUT/sq > $OUT/sq_cv1_b8.txt 2>&1
bash tools/c3_capture.sh $OUT/c3 > $OUT/c3.log 2>&1
bash tools/cv1_phases.sh 8 > $OUT/cv1_phases_b8.txt 2>&1
bash tools/cv1_phases.sh 1 > $OUT/cv1_phases_b1.txt 2>&1
bash tools/cv1_clock.sh 1 > $OUT/cv1_clock_b1.txt 2>&1
find $OUT -name "*.db" -delete; find $OUT -size +3M -delete
ls $OUT
