set -x
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_gpu_1.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r02/pytest_gpu_1.txt
tail -5 gpurun_out/r02/pytest_gpu_1.txt
python bench.py > gpurun_out/r02/bench_base.json 2> gpurun_out/r02/bench_base.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/r02/bench_base.json
DSIM_LIB=disimpy_amd/csrc/libdsim_hip_stats.so python tools/mesh_stats.py c5b > gpurun_out/r02/stats_c5b_base.txt 2>&1
DSIM_LIB=disimpy_amd/csrc/libdsim_hip_timers.so python tools/mesh_timers.py c5b > gpurun_out/r02/timers_c5b_base.txt 2>&1
cat gpurun_out/r02/stats_c5b_base.txt gpurun_out/r02/timers_c5b_base.txt
python3 tools/pmc_profile.py c5b --tag r02_base > gpurun_out/r02/pmc_c5b_base.txt 2>&1
cat gpurun_out/r02/pmc_c5b_base.txt
cp profiles/r02_base_c5b_pmc.json profiles/kernel_counters.json gpurun_out/r02/
