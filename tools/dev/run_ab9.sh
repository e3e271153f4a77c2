mkdir -p gpurun_out/r03
DSIM_LIB=$PWD/disimpy_amd/csrc/libdsim_hip_v2.so timeout -k 5 400 python -m pytest tests/test_gpu_production.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -3
bash tools/gpu_ab.sh a9 c5b 10000000 "libdsim_hip.so libdsim_hip_v2.so" > gpurun_out/r03/ab_a9.txt 2>&1
bash tools/gpu_ab.sh a9s c5b 1250000 "libdsim_hip.so libdsim_hip_v2.so" >> gpurun_out/r03/ab_a9.txt 2>&1
bash tools/gpu_ab.sh a9c4 c4 1000000 "libdsim_hip.so libdsim_hip_v2.so" >> gpurun_out/r03/ab_a9.txt 2>&1
cat gpurun_out/r03/ab_a9.txt
