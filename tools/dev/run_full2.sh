mkdir -p gpurun_out/r03
timeout -k 10 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -30 > gpurun_out/r03/pytest_gpu_b2.txt
cat gpurun_out/r03/pytest_gpu_b2.txt
