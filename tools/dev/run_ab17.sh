mkdir -p gpurun_out/r03
timeout -k 5 600 python -m pytest tests/test_gpu_production.py -x -q -m gpu -k "many_measurements or recording or not_the_references" 2>&1 | grep -E "passed|failed|rror|assert" | tail -5
bash tools/gpu_ab.sh a25m32 c5b 10000000 "libdsim_hip_cur.so libdsim_hip.so" "--measurements 32" > gpurun_out/r03/ab_a25.txt 2>&1
bash tools/gpu_ab.sh a25m8 c5b 10000000 "libdsim_hip_cur.so libdsim_hip.so" "--measurements 8" >> gpurun_out/r03/ab_a25.txt 2>&1
cat gpurun_out/r03/ab_a25.txt
