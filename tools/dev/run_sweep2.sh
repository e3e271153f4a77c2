mkdir -p gpurun_out/r03
N=1250000
run() { # tag lib env extra
  DSIM_BENCH_POSITIONS_CACHE=/tmp/pos_c5b_$N.npy DSIM_LIB=$PWD/disimpy_amd/csrc/$2 env $3 timeout -k 5 300 python bench.py --workload c5b --walkers $N --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check $4 > gpurun_out/r03/sw_$1.json 2> gpurun_out/r03/sw_$1.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03/sw_$1.json").read().strip().splitlines()[-1]); print("$1: %.4g" % d["value"])
except Exception as e: print("$1 FAILED", e)
PY
}
run b_auto libdsim_hip.so A=1 ""
run b_r128 libdsim_hip.so A=1 "--resort-steps 128"
run b_r512 libdsim_hip.so A=1 "--resort-steps 512"
run b_r1024 libdsim_hip.so A=1 "--resort-steps 1024"
run b_g1_256 libdsim_hip.so DSIM_GROUPS=1 ""
run b_g1_1024 libdsim_hip.so DSIM_GROUPS=1 "--resort-steps 1024"
run b_bits18_512 libdsim_hip.so DSIM_SORT_BITS=18 "--resort-steps 512"
run b_auto2 libdsim_hip.so A=1 ""
