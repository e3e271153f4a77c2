mkdir -p gpurun_out/r03
bash tools/gpu_ab.sh a1 c5b 10000000 "libdsim_hip_r02.so libdsim_hip.so" > gpurun_out/r03/ab_a1.txt 2>&1
bash tools/gpu_ab.sh a1s c5b 1250000 "libdsim_hip_r02.so libdsim_hip.so" >> gpurun_out/r03/ab_a1.txt 2>&1
bash tools/gpu_ab.sh a1c4 c4 1000000 "libdsim_hip_r02.so libdsim_hip.so" >> gpurun_out/r03/ab_a1.txt 2>&1
DSIM_LIB=$PWD/disimpy_amd/csrc/libdsim_hip_timers.so python tools/mesh_timers.py c5b 1000000 > gpurun_out/r03/timers_a1.txt 2>&1
DSIM_LIB=$PWD/disimpy_amd/csrc/libdsim_hip_stats.so python tools/mesh_stats.py c5b > gpurun_out/r03/stats_a1.txt 2>&1
cat gpurun_out/r03/ab_a1.txt gpurun_out/r03/timers_a1.txt gpurun_out/r03/stats_a1.txt
