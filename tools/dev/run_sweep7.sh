mkdir -p gpurun_out/r03
run() { # tag N env extra
  DSIM_BENCH_POSITIONS_CACHE=/tmp/pos_c5b_$2.npy env $3 timeout -k 5 300 python bench.py --workload c5b --walkers $2 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check $4 > gpurun_out/r03/sw_$1.json 2> gpurun_out/r03/sw_$1.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03/sw_$1.json").read().strip().splitlines()[-1]); print("$1: %.4g  walk %.4g" % (d["value"], d["walk_kernel_spin_steps_per_s"]))
except Exception as e: print("$1 FAILED", e)
PY
}
for x in 3 2 1 0; do run g_x$x 1250000 DSIM_XCD_RUN=$x ""; done
for x in 3 2 1 0; do run g10_x$x 10000000 DSIM_XCD_RUN=$x ""; done
run g_x3_r512 1250000 DSIM_XCD_RUN=3 "--resort-steps 512"
run g_x2_r512 1250000 DSIM_XCD_RUN=2 "--resort-steps 512"
