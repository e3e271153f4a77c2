mkdir -p gpurun_out/r03
timeout -k 5 500 python -m pytest tests/test_gpu_device_units.py tests/test_gpu_parity.py tests/test_gpu_mesh_adversarial.py tests/test_gpu_production.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -3
bash tools/gpu_ab.sh a13 c5b 10000000 "libdsim_hip_cur.so libdsim_hip.so" > gpurun_out/r03/ab_a13.txt 2>&1
bash tools/gpu_ab.sh a13c2 c2 0 "libdsim_hip_cur.so libdsim_hip.so" >> gpurun_out/r03/ab_a13.txt 2>&1
bash tools/gpu_ab.sh a13c3 c3 0 "libdsim_hip_cur.so libdsim_hip.so" >> gpurun_out/r03/ab_a13.txt 2>&1
bash tools/gpu_ab.sh a13c4 c4 1000000 "libdsim_hip_cur.so libdsim_hip.so" >> gpurun_out/r03/ab_a13.txt 2>&1
cat gpurun_out/r03/ab_a13.txt
