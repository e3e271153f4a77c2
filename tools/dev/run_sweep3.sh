mkdir -p gpurun_out/r03
run() { # tag N env extra
  DSIM_BENCH_POSITIONS_CACHE=/tmp/pos_c5b_$2.npy env $3 timeout -k 5 300 python bench.py --workload c5b --walkers $2 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check $4 > gpurun_out/r03/sw_$1.json 2> gpurun_out/r03/sw_$1.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03/sw_$1.json").read().strip().splitlines()[-1]); print("$1: %.4g  S/N %.9f" % (d["value"], d["signal_check"]["last_pass_mean_S_over_N"]))
except Exception as e: print("$1 FAILED", e)
PY
}
run c_g2 1250000 DSIM_GROUPS=2 ""
run c_g3 1250000 DSIM_GROUPS=3 ""
run c_g4 1250000 DSIM_GROUPS=4 ""
run c_g3_512 1250000 DSIM_GROUPS=3 "--resort-steps 512"
run c_g4_512 1250000 DSIM_GROUPS=4 "--resort-steps 512"
run c_g2b 1250000 DSIM_GROUPS=2 ""
run c_g2_10m 10000000 DSIM_GROUPS=2 ""
run c_g4_10m 10000000 DSIM_GROUPS=4 ""
