mkdir -p gpurun_out/r03
timeout -k 5 400 python -m pytest tests/test_gpu_production.py tests/test_gpu_parity.py tests/test_gpu_api.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -3
run() { # tag wl N env extra
  DSIM_BENCH_POSITIONS_CACHE=/tmp/pos_$2_$3.npy env $4 timeout -k 5 400 python bench.py --workload $2 --walkers $3 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check $5 > gpurun_out/r03/sw_$1.json 2> gpurun_out/r03/sw_$1.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03/sw_$1.json").read().strip().splitlines()[-1]); print("$1: %.4g  walk %.4g  launches %s kernel %s" % (d["value"], d["walk_kernel_spin_steps_per_s"], d["roofline"]["launches_per_step"], d["roofline"]["kernel"]))
except Exception as e: print("$1 FAILED", e); print(open("gpurun_out/r03/sw_$1.err").read()[-800:])
PY
}
run r_m3 c5 1000000 A=1 "--measurements 3"
run r_m4 c5 1000000 A=1 "--measurements 4"
run r_m5 c5 1000000 A=1 "--measurements 5"
run r_m1 c5 1000000 A=1 ""
