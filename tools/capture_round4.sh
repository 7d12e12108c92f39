#!/bin/bash
# Round-4 evidence in one go (on the GPU box, from the repo root): bash tools/capture_round4.sh
# -> gpurun_out/r04/: the driver's bench line, kernel stats at 8 lanes / 1 lane, the launch sequence of one forward,
#    configs[2] (batch 8, fp16 features) trace + FETCH_SIZE / WRITE_SIZE passes, SQ counters of cost-volume stage 1 at batch 8
#    (register-resident kernel from idx / mask, and the tile kernel with in-kernel grouping), PMC traffic of the timed
#    kernel, the cost-volume variants table, the free-running parity statistic, the phase clock of the register-resident
#    kernel (tools/rr_clock.sh), the MFMA SrcC hazard table (tools/micro/mfma_srcc_hazard.hip), the A/B of the narrow set-conv
#    kernel forms and the training step times / per-kernel GPU time.
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r04; mkdir -p $OUT
python bench.py > $OUT/bench_line.json 2> $OUT/bench_stderr.log
python tools/cv1_variants.py --levels l0,l1,l2,full725 --batches 1,8 2>/dev/null | grep "^{" > $OUT/cv1_variants.txt
python tools/parity_flips.py > $OUT/parity_flips.txt 2>/dev/null
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_8lanes -o t -- python $REPO/bench.py --steps 200 --warmup 16 --no-legs > $OUT/trace_8lanes.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_1lane -o t -- python $REPO/bench.py --steps 200 --warmup 16 --no-legs --lanes 1 > $OUT/trace_1lane.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/seq -o t -- python $REPO/bench.py --lanes 1 --steps 3 --warmup 1 --no-legs --pool 1 > $OUT/seq.log 2>&1
cd $REPO
python tools/forward_sequence.py $(find $OUT/seq -name "*kernel_trace.csv" | head -1) > $OUT/forward_sequence_1lane.txt 2>&1
ELO_PMC_SPECS="cv1:1 cv1:8 cv1_f16:8" bash tools/pmc_collect.sh $OUT/pmc > $OUT/pmc.log 2>&1
bash tools/sq_counters.sh cv1 8 $OUT/sq_rr --pregrouped > $OUT/sq_cv1_rr_b8.txt 2>&1
ELO_CV_PREPASS=0 bash tools/sq_counters.sh cv1 8 $OUT/sq_tile > $OUT/sq_cv1_tile_b8.txt 2>&1
bash tools/c3_capture.sh $OUT/c3 > $OUT/c3.log 2>&1
bash tools/rr_clock.sh 8 0 700 2>&1 | grep workgroup > $OUT/rr_clock_b8.txt
ELO_RR_FLAGS=-DELO_RR_CLOCK_POOL bash tools/rr_clock.sh 8 700 2>&1 | grep workgroup >> $OUT/rr_clock_b8.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_srcc_hazard.hip -o /tmp/mfma_srcc_hazard > /dev/null 2>&1 && /tmp/mfma_srcc_hazard > $OUT/mfma_srcc_hazard.txt 2>&1
bash tools/ab_narrow.sh $OUT/ab_narrow > $OUT/ab_narrow.txt 2>&1
python tools/train_step_time.py > $OUT/training.txt 2>&1
python tools/train_kernel_stats.py 8 >> $OUT/training.txt 2>&1
find $OUT -name "*.db" -delete; find $OUT -size +3M -delete
ls $OUT
