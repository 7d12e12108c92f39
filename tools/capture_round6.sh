#!/bin/bash
# Round-6 evidence in one go (GPU box, repo root): bash tools/capture_round6.sh -> gpurun_out/r06/final/
#   the driver's bench line; kernel stats at 8 lanes / 1 lane and the launch sequence of one forward (dense scene AND the KITTI-density
#   scene is in the bench line's `sparse` leg); configs[2] trace + counter passes; the HBM-COLD counter passes of the per-operator
#   cost-volume kernels at the 128 x 2048 l0 shape (ring of tensor sets: ELO_PMC_COLD=1) beside the warm ones; the cold sweep;
#   the streaming ceiling of this box (tools/micro/hbm_probe); training step time + kernel histogram; the parity statistic.
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r06/final; mkdir -p $OUT
python bench.py --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench_stderr.log
python tools/parity_flips.py > $OUT/parity_flips.txt 2>/dev/null
for dt in "" "--half"; do python tools/cold_sweep.py $dt > $OUT/cold_sweep${dt:+_f16}.txt 2>&1; done
python tools/cold_sweep.py --half --grid 16x225 > $OUT/cold_sweep_f16_16x225.txt 2>&1
tools/micro/build/hbm_probe 200 > $OUT/hbm_probe_200MB.txt 2>&1
tools/micro/build/hbm_probe 64 > $OUT/hbm_probe_64MB.txt 2>&1
ELO_PMC_GRID=32x256 ELO_PMC_COLD=1 ELO_PMC_SPECS="encode1:8 pool:8 encode2:8 pool2:8 encode1_f16:8 pool_f16:8 encode2_f16:8 pool2_f16:8" bash tools/pmc_collect.sh $OUT/pmc_hires_cold > $OUT/pmc_hires_cold.log 2>&1
ELO_PMC_GRID=32x256 ELO_PMC_SPECS="encode1:8 pool:8 encode2:8 pool2:8 encode1_f16:8 pool_f16:8 encode2_f16:8 pool2_f16:8" bash tools/pmc_collect.sh $OUT/pmc_hires_warm > $OUT/pmc_hires_warm.log 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_8lanes -o t -- python $REPO/bench.py --steps 200 --warmup 16 --no-legs > $OUT/trace_8lanes.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_1lane -o t -- python $REPO/bench.py --steps 200 --warmup 16 --no-legs --lanes 1 > $OUT/trace_1lane.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/seq -o t -- python $REPO/bench.py --lanes 1 --steps 3 --warmup 1 --no-legs --pool 1 > $OUT/seq.log 2>&1
cd $REPO
python tools/forward_sequence.py $(find $OUT/seq -name "*kernel_trace.csv" | head -1) > $OUT/forward_sequence_1lane.txt 2>&1
bash tools/c3_capture.sh $OUT/c3 > $OUT/c3.log 2>&1
python tools/train_step_time.py 8 > $OUT/training.txt 2>&1
python tools/train_kernel_stats.py 8 >> $OUT/training.txt 2>&1
find $OUT -name "*.db" -delete; find $OUT -size +3M -delete
ls $OUT
