mkdir -p gpurun_out/r03
timeout -k 5 400 python -m pytest tests/test_gpu_production.py tests/test_gpu_parity.py tests/test_gpu_api.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r03/tests_a6.txt
cat gpurun_out/r03/tests_a6.txt
bash tools/gpu_ab.sh a6s c5b 1250000 "libdsim_hip_r02.so libdsim_hip.so" > gpurun_out/r03/ab_a6.txt 2>&1
bash tools/gpu_ab.sh a6 c5b 10000000 "libdsim_hip_r02.so libdsim_hip.so" >> gpurun_out/r03/ab_a6.txt 2>&1
bash tools/gpu_ab.sh a6c4 c4 1000000 "libdsim_hip_r02.so libdsim_hip.so" >> gpurun_out/r03/ab_a6.txt 2>&1
for b in 15 18 24; do DSIM_SORT_BITS=$b bash tools/gpu_ab.sh a6bits$b c5b 1250000 "libdsim_hip.so" | head -1 >> gpurun_out/r03/ab_a6.txt 2>&1; done
cat gpurun_out/r03/ab_a6.txt
