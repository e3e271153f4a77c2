mkdir -p gpurun_out/r03
run() { # tag lib N extra
  DSIM_BENCH_POSITIONS_CACHE=/tmp/pos_c5b_$3.npy DSIM_LIB=$PWD/disimpy_amd/csrc/$2 timeout -k 5 300 python bench.py --workload c5b --walkers $3 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check $4 > gpurun_out/r03/sw_$1.json 2> gpurun_out/r03/sw_$1.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r03/sw_$1.json").read().strip().splitlines()[-1]); print("$1: %.4g  walk %.4g S/N %.9f" % (d["value"], d["walk_kernel_spin_steps_per_s"], d["signal_check"]["last_pass_mean_S_over_N"]))
except Exception as e: print("$1 FAILED", e); print(open("gpurun_out/r03/sw_$1.err").read()[-600:])
PY
}
run q_cur libdsim_hip.so 10000000 ""
run q_w5 libdsim_hip_w5.so 10000000 "--block-size 320"
run q_w5h libdsim_hip_w5h.so 10000000 "--block-size 320"
run q_w5_b256 libdsim_hip_w5.so 10000000 ""
run q_cur2 libdsim_hip.so 10000000 ""
run q_w5_s libdsim_hip_w5.so 1250000 "--block-size 320"
run q_cur_s libdsim_hip.so 1250000 ""
