mkdir -p gpurun_out/r03
timeout -k 5 300 python -m pytest tests/test_gpu_mesh_adversarial.py tests/test_gpu_production.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r03/tests_a5.txt
bash tools/gpu_ab.sh a5 c5b 10000000 "libdsim_hip_r02.so libdsim_hip.so" > gpurun_out/r03/ab_a5.txt 2>&1
bash tools/gpu_ab.sh a5c4 c4 1000000 "libdsim_hip_r02.so libdsim_hip.so" >> gpurun_out/r03/ab_a5.txt 2>&1
DSIM_LIB=$PWD/disimpy_amd/csrc/libdsim_hip_timers.so timeout -k 5 300 python tools/mesh_timers.py c5b 1000000 > gpurun_out/r03/timers_a5.txt 2>&1
cat gpurun_out/r03/tests_a5.txt gpurun_out/r03/ab_a5.txt gpurun_out/r03/timers_a5.txt
