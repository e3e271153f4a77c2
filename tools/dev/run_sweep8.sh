mkdir -p gpurun_out/r03
run() { # tag wl N env extra
  DSIM_BENCH_POSITIONS_CACHE=/tmp/pos_$2_$3.npy env $4 timeout -k 5 300 python bench.py --workload $2 --walkers $3 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check $5 > gpurun_out/r03/sw_$1.json 2> gpurun_out/r03/sw_$1.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03/sw_$1.json").read().strip().splitlines()[-1]); print("$1: %.4g  walk %.4g" % (d["value"], d["walk_kernel_spin_steps_per_s"]))
except Exception as e: print("$1 FAILED", e)
PY
}
run h_b15 c5b 1250000 "DSIM_XCD_RUN=3 DSIM_SORT_BITS=15" ""
run h_b18 c5b 1250000 "DSIM_XCD_RUN=3 DSIM_SORT_BITS=18" ""
run h_b21 c5b 1250000 "DSIM_XCD_RUN=3 DSIM_SORT_BITS=21" ""
run h_g3 c5b 1250000 "DSIM_XCD_RUN=3 DSIM_GROUPS=3" ""
run h_g1 c5b 1250000 "DSIM_XCD_RUN=3 DSIM_GROUPS=1" ""
run h_c4_x5 c4 1000000 "DSIM_XCD_RUN=5" ""
run h_c4_x3 c4 1000000 "DSIM_XCD_RUN=3" ""
run h_c4_x2 c4 1000000 "DSIM_XCD_RUN=2" ""
run h_25_x3 c5b 2500000 "DSIM_XCD_RUN=3" ""
run h_25_x5 c5b 2500000 "DSIM_XCD_RUN=5" ""
