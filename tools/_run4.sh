mkdir -p gpurun_out/r05
python -m pytest tests/test_grouping_gpu.py -q -k "lds_tiled_setconv" > gpurun_out/r05/gputests_tiled.log 2>&1; tail -15 gpurun_out/r05/gputests_tiled.log
python tools/ab_tiled.py > gpurun_out/r05/ab_tiled.txt 2>&1; cat gpurun_out/r05/ab_tiled.txt
for t in 0 1 2; do for f in f16 f32; do ELO_TILED_SETCONV=$t python bench.py --batch 8 --features $f --steps 240 --warmup 16 --no-legs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ELO_TILED_SETCONV=$t $f', d['value'])"; done; done | tee gpurun_out/r05/ab_tiled_b8.txt
for t in 0 1; do ELO_TILED_SETCONV=$t python bench.py --steps 200 --warmup 16 --no-legs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ELO_TILED_SETCONV=$t batch1', d['value'])"; done | tee -a gpurun_out/r05/ab_tiled_b8.txt
python -m pytest tests -m gpu -q -x > gpurun_out/r05/gputests_3.log 2>&1; tail -8 gpurun_out/r05/gputests_3.log
