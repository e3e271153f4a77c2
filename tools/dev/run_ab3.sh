mkdir -p gpurun_out/r03
bash tools/gpu_ab.sh a3 c5b 10000000 "libdsim_hip_r02.so libdsim_hip.so" > gpurun_out/r03/ab_a3.txt 2>&1
bash tools/gpu_ab.sh a3c4 c4 1000000 "libdsim_hip_r02.so libdsim_hip.so" >> gpurun_out/r03/ab_a3.txt 2>&1
DSIM_LIB=$PWD/disimpy_amd/csrc/libdsim_hip_timers.so timeout -k 5 300 python tools/mesh_timers.py c5b 1000000 > gpurun_out/r03/timers_a3.txt 2>&1
cat gpurun_out/r03/ab_a3.txt gpurun_out/r03/timers_a3.txt
