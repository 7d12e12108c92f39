set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
for pw in 1 0; do
  ELO_POOL_WAVE=$pw python tools/cold_sweep.py > gpurun_out/r06/cold_sweep_f32_pw$pw.txt 2>&1
  ELO_POOL_WAVE=$pw python tools/cold_sweep.py --half > gpurun_out/r06/cold_sweep_f16_pw$pw.txt 2>&1
  ELO_POOL_WAVE=$pw python tools/cold_sweep.py --half --grid 16x225 > gpurun_out/r06/cold_sweep_f16_16x225_pw$pw.txt 2>&1
done
tail -n 5 gpurun_out/r06/cold_sweep_*pw*.txt
for at in 1 0; do
  ELO_TRAIN_ATOMICS=$at python tools/train_step_time.py 8 > gpurun_out/r06/train_time_atomics$at.txt 2>&1
  ELO_TRAIN_ATOMICS=$at python tools/train_kernel_stats.py 8 > gpurun_out/r06/train_stats_atomics$at.txt 2>&1
done
cat gpurun_out/r06/train_time_atomics*.txt; head -30 gpurun_out/r06/train_stats_atomics1.txt
timeout 2400 python -m pytest tests/test_sv_ride_gpu.py tests/test_chain_ops_gpu.py tests/test_host_logic.py tests/test_train_kernels_gpu.py tests/test_training_gpu.py "tests/test_model_gpu.py::test_kitti_density_scene_matches_oracle" "tests/test_model_gpu.py::test_batch8_kitti_density_matches_oracle_level_by_level" tests/test_ops_gpu.py -x -q 2>&1 | tail -30 > gpurun_out/r06/tests_g3.txt
cat gpurun_out/r06/tests_g3.txt
