set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python tools/cold_sweep.py --half > gpurun_out/r06/cold_sweep_f16_pw1c.txt 2>&1
python tools/cold_sweep.py --half --grid 16x225 > gpurun_out/r06/cold_sweep_f16_16x225_pw1c.txt 2>&1
python tools/cold_sweep.py > gpurun_out/r06/cold_sweep_f32_pw1c.txt 2>&1
tail -n 5 gpurun_out/r06/cold_sweep_*c.txt
timeout 2400 python -m pytest tests/test_ops_gpu.py tests/test_sv_ride_gpu.py tests/test_chain_ops_gpu.py tests/test_train_kernels_gpu.py tests/test_training_gpu.py "tests/test_model_gpu.py::test_kitti_density_scene_matches_oracle" "tests/test_model_gpu.py::test_batch8_kitti_density_matches_oracle_level_by_level" tests/test_bench_gpu.py -q 2>&1 | tail -40 > gpurun_out/r06/tests_g6.txt
cat gpurun_out/r06/tests_g6.txt
