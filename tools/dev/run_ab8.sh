mkdir -p gpurun_out/r03
timeout -k 5 300 python -m pytest tests/test_gpu_mesh_adversarial.py tests/test_gpu_production.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2
bash tools/gpu_ab.sh a8 c5b 10000000 "libdsim_hip_cur.so libdsim_hip.so libdsim_hip_pf.so" > gpurun_out/r03/ab_a8.txt 2>&1
cat gpurun_out/r03/ab_a8.txt
