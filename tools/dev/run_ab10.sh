mkdir -p gpurun_out/r03
bash tools/gpu_ab.sh a10 c5b 10000000 "libdsim_hip.so libdsim_hip_of.so libdsim_hip_nd.so libdsim_hip_ofnd.so" > gpurun_out/r03/ab_a10.txt 2>&1
cat gpurun_out/r03/ab_a10.txt
