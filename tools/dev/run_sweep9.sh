mkdir -p gpurun_out/r03
run() { # tag wl N env extra
  DSIM_BENCH_POSITIONS_CACHE=/tmp/pos_$2_$3.npy env $4 timeout -k 5 400 python bench.py --workload $2 --walkers $3 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check $5 > gpurun_out/r03/sw_$1.json 2> gpurun_out/r03/sw_$1.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03/sw_$1.json").read().strip().splitlines()[-1]); print("$1: %.4g  walk %.4g  launches %s" % (d["value"], d["walk_kernel_spin_steps_per_s"], d["roofline"]["launches_per_step"]))
except Exception as e: print("$1 FAILED", e); print(open("gpurun_out/r03/sw_$1.err").read()[-800:])
PY
}
run p_auto c5b 10000000 A=1 ""
run p_r256 c5b 10000000 A=1 "--resort-steps 256"
run p_r512 c5b 10000000 A=1 "--resort-steps 512"
run p_r2048 c5b 10000000 A=1 "--resort-steps 2048"
run p_auto2 c5b 10000000 A=1 ""
