mkdir -p gpurun_out/r03
O=gpurun_out/r03/mtable.txt; : > $O
run() { # tag workload walkers M
  DSIM_BENCH_POSITIONS_CACHE=/tmp/pos_$2_$3.npy timeout -k 5 400 python bench.py --workload $2 --walkers $3 --measurements $4 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check > gpurun_out/r03/mt_$1.json 2> gpurun_out/r03/mt_$1.err
  python - <<PY >> $O
import json
try:
    d=json.loads(open("gpurun_out/r03/mt_$1.json").read().strip().splitlines()[-1])
    print("$1: $2 n=$3 M=$4 value %.4g  walk-kernel %.4g" % (d["value"], d["walk_kernel_spin_steps_per_s"]))
except Exception as e:
    print("$1 FAILED", e); print(open("gpurun_out/r03/mt_$1.err").read()[-800:])
PY
}
run c5b_m32 c5b 10000000 32
run c5b_m8 c5b 10000000 8
run c5b_m4 c5b 10000000 4
run c5_m1 c5 1000000 1
run c5_m2 c5 1000000 2
run c5_m4 c5 1000000 4
run c5_m32 c5 1000000 32
run c4_10m c4 10000000 1
cat $O
