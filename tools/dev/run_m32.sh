mkdir -p gpurun_out/r03
run() { # tag wl N env extra
  DSIM_BENCH_POSITIONS_CACHE=/tmp/pos_$2_$3.npy env $4 timeout -k 5 400 python bench.py --workload $2 --walkers $3 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check $5 > gpurun_out/r03/sw_$1.json 2> gpurun_out/r03/sw_$1.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03/sw_$1.json").read().strip().splitlines()[-1]); print("$1: %.4g  walk %.4g  launches %s avg_launch_ms %.1f" % (d["value"], d["walk_kernel_spin_steps_per_s"], d["roofline"]["launches_per_step"], d["roofline"]["avg_launch_ms"]))
except Exception as e: print("$1 FAILED", e); print(open("gpurun_out/r03/sw_$1.err").read()[-800:])
PY
}
run m1 c5 1000000 A=1 ""
run m2 c5 1000000 A=1 "--measurements 2"
run m32 c5 1000000 A=1 "--measurements 32"
run m4 c5 1000000 A=1 "--measurements 4"
