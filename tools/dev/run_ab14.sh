mkdir -p gpurun_out/r03
L="libdsim_hip.so libdsim_hip_max-ilp.so libdsim_hip_max-memory-clause.so libdsim_hip_iterative-maxocc.so"
bash tools/gpu_ab.sh a15 c5b 10000000 "$L" > gpurun_out/r03/ab_a15.txt 2>&1
bash tools/gpu_ab.sh a15c2 c2 0 "$L" >> gpurun_out/r03/ab_a15.txt 2>&1
bash tools/gpu_ab.sh a15c3 c3 0 "$L" >> gpurun_out/r03/ab_a15.txt 2>&1
bash tools/gpu_ab.sh a15c4 c4 1000000 "$L" >> gpurun_out/r03/ab_a15.txt 2>&1
cat gpurun_out/r03/ab_a15.txt
