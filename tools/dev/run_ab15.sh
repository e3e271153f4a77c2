mkdir -p gpurun_out/r03
timeout -k 5 600 python -m pytest tests/test_gpu_production.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror|assert" | tail -5
bash tools/gpu_ab.sh a17s c5b 1250000 "libdsim_hip_cur.so libdsim_hip.so" > gpurun_out/r03/ab_a17.txt 2>&1
bash tools/gpu_ab.sh a17 c5b 10000000 "libdsim_hip_cur.so libdsim_hip.so" >> gpurun_out/r03/ab_a17.txt 2>&1
bash tools/gpu_ab.sh a17c4 c4 2500000 "libdsim_hip_cur.so libdsim_hip.so" >> gpurun_out/r03/ab_a17.txt 2>&1
cat gpurun_out/r03/ab_a17.txt
