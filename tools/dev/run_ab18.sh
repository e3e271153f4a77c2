mkdir -p gpurun_out/r03
bash tools/gpu_ab.sh a26 c5b 10000000 "libdsim_hip.so libdsim_hip_p13.so libdsim_hip_p14.so" > gpurun_out/r03/ab_a26.txt 2>&1
cat gpurun_out/r03/ab_a26.txt
