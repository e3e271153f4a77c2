mkdir -p gpurun_out/r03/last
timeout -k 10 900 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -4 > gpurun_out/r03/last/pytest_gpu.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03/last/smoke.txt 2>&1
timeout -k 10 600 python bench.py > gpurun_out/r03/last/bench_default_line.json 2> gpurun_out/r03/last/bench_default.err
cat gpurun_out/r03/last/pytest_gpu.txt; tail -2 gpurun_out/r03/last/smoke.txt
python - <<PY
import json
d=json.loads(open("gpurun_out/r03/last/bench_default_line.json").read().strip().splitlines()[-1]); r=d["roofline"]
print(d["value"], r["frac"], r["counters_match_lib"], r["counters_match_config"], r["valu"]["valu_busy_measured"], r["traffic"], r["avg_launch_ms"])
print({k:v["value"] for k,v in d["secondary"].items()}); print(d["cpu_baseline"])
PY
