mkdir -p gpurun_out/r03
python -m pytest tests/test_gpu_mesh_adversarial.py tests/test_gpu_production.py tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -5 > gpurun_out/r03/tests_a2.txt
bash tools/gpu_ab.sh a2 c5b 10000000 "libdsim_hip_r02.so libdsim_hip.so" > gpurun_out/r03/ab_a2.txt 2>&1
bash tools/gpu_ab.sh a2c4 c4 1000000 "libdsim_hip_r02.so libdsim_hip.so" >> gpurun_out/r03/ab_a2.txt 2>&1
DSIM_LIB=$PWD/disimpy_amd/csrc/libdsim_hip_timers.so python tools/mesh_timers.py c5b 1000000 > gpurun_out/r03/timers_a2.txt 2>&1
DSIM_LIB=$PWD/disimpy_amd/csrc/libdsim_hip_timers.so python tools/mesh_timers.py c4 1000000 > gpurun_out/r03/timers_a2_c4.txt 2>&1
DSIM_LIB=$PWD/disimpy_amd/csrc/libdsim_hip_stats.so python tools/mesh_stats.py c5b > gpurun_out/r03/stats_a2.txt 2>&1
DSIM_LIB=$PWD/disimpy_amd/csrc/libdsim_hip_stats.so python tools/mesh_stats.py c4 > gpurun_out/r03/stats_a2_c4.txt 2>&1
cat gpurun_out/r03/tests_a2.txt gpurun_out/r03/ab_a2.txt gpurun_out/r03/timers_a2.txt gpurun_out/r03/timers_a2_c4.txt gpurun_out/r03/stats_a2.txt gpurun_out/r03/stats_a2_c4.txt
