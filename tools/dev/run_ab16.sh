mkdir -p gpurun_out/r03
L="libdsim_hip.so libdsim_hip_p7.so libdsim_hip_p10.so libdsim_hip_p11.so libdsim_hip_p12.so"
bash tools/gpu_ab.sh a21 c5b 10000000 "$L" > gpurun_out/r03/ab_a21.txt 2>&1
bash tools/gpu_ab.sh a21c4 c4 1000000 "$L" >> gpurun_out/r03/ab_a21.txt 2>&1
cat gpurun_out/r03/ab_a21.txt
