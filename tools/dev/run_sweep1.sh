mkdir -p gpurun_out/r03
N=2500000
run() { # tag lib env
  DSIM_BENCH_POSITIONS_CACHE=/tmp/pos_c5b_$N.npy DSIM_LIB=$PWD/disimpy_amd/csrc/$2 env $3 timeout -k 5 300 python bench.py --workload c5b --walkers $N --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check > gpurun_out/r03/sw_$1.json 2> gpurun_out/r03/sw_$1.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03/sw_$1.json").read().strip().splitlines()[-1]); print("$1: %.4g" % d["value"])
except Exception as e: print("$1 FAILED", e)
PY
}
run r02 libdsim_hip_r02.so A=1
run a2 libdsim_hip.so A=1
run a2nu libdsim_hip_nu.so A=1
run a2_lk2 libdsim_hip.so DSIM_MICRO=2,4,1000,1000,1.5
run a2_lk4 libdsim_hip.so DSIM_MICRO=2,4,16,32,0.8
run a2_lk3_min1 libdsim_hip.so DSIM_MICRO=2,4,16,1000,1.0
run r02b libdsim_hip_r02.so A=1
run a2b libdsim_hip.so A=1
