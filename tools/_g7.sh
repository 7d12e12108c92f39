set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python tools/cold_sweep.py --half > gpurun_out/r06/cold_sweep_f16_pw1d.txt 2>&1
python tools/cold_sweep.py --half --grid 16x225 > gpurun_out/r06/cold_sweep_f16_16x225_pw1d.txt 2>&1
python tools/cold_sweep.py > gpurun_out/r06/cold_sweep_f32_pw1d.txt 2>&1
tail -n 5 gpurun_out/r06/cold_sweep_*d.txt
timeout 3000 python -m pytest tests -m gpu -q -x --deselect tests/test_bench_gpu.py 2>&1 | tail -40 > gpurun_out/r06/tests_g7.txt
tail -15 gpurun_out/r06/tests_g7.txt
timeout 1500 python -m pytest tests/test_bench_gpu.py -q 2>&1 | tail -60 > gpurun_out/r06/tests_g7_bench.txt
tail -30 gpurun_out/r06/tests_g7_bench.txt | cut -c1-300
python bench.py --steps 20 --warmup 5 --no-legs --submit-order stream > gpurun_out/r06/bench_stream_order.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-legs > gpurun_out/r06/bench_caller_order.json 2>/dev/null
cut -c1-200 gpurun_out/r06/bench_stream_order.json gpurun_out/r06/bench_caller_order.json
