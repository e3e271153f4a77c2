mkdir -p gpurun_out/r03
timeout -k 5 400 python -m pytest tests/test_gpu_mesh_adversarial.py tests/test_gpu_production.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -3
bash tools/gpu_ab.sh a12 c5b 10000000 "libdsim_hip_cur.so libdsim_hip.so" > gpurun_out/r03/ab_a12.txt 2>&1
bash tools/gpu_ab.sh a12c4 c4 1000000 "libdsim_hip_cur.so libdsim_hip.so" >> gpurun_out/r03/ab_a12.txt 2>&1
cat gpurun_out/r03/ab_a12.txt
