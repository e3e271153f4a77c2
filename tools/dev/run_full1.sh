mkdir -p gpurun_out/r03
timeout -k 10 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -25 > gpurun_out/r03/pytest_gpu_b1.txt
cat gpurun_out/r03/pytest_gpu_b1.txt
timeout -k 10 600 python bench.py > gpurun_out/r03/bench_default_b1.json 2> gpurun_out/r03/bench_default_b1.err
tail -c 3000 gpurun_out/r03/bench_default_b1.json
