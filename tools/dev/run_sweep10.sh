mkdir -p gpurun_out/r03
O=gpurun_out/r03/sweep10.txt; : > $O
run() { # tag env... -- extra bench args
  tag=$1; shift
  envs=""; while [ "$1" != "--" ]; do envs="$envs $1"; shift; done; shift
  env $envs DSIM_BENCH_POSITIONS_CACHE=/tmp/pos_c5b_10000000.npy timeout -k 5 300 python bench.py --workload c5b --walkers 10000000 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check "$@" > gpurun_out/r03/sw_$tag.json 2> gpurun_out/r03/sw_$tag.err
  python - <<PY >> $O
import json
try:
    d=json.loads(open("gpurun_out/r03/sw_$tag.json").read().strip().splitlines()[-1])
    print("$tag: value %.4g  walk-kernel %.4g  launches/step %s" % (d["value"], d["walk_kernel_spin_steps_per_s"], d["roofline"]["launches_per_step"]))
except Exception as e:
    print("$tag FAILED", e)
PY
}
run base --
run r768 -- --resort-steps 768
run r1536 -- --resort-steps 1536
run r2048 -- --resort-steps 2048
run xcd2 DSIM_XCD_RUN=2 --
run xcd4 DSIM_XCD_RUN=4 --
run g3 DSIM_GROUPS=3 --
run base2 --
cat $O
