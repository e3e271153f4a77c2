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
run e_b15 10000000 DSIM_SORT_BITS=15 ""
run e_b18 10000000 DSIM_SORT_BITS=18 ""
run e_b21 10000000 DSIM_SORT_BITS=21 ""
run e_b24 10000000 DSIM_SORT_BITS=24 ""
