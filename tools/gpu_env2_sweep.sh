# same-box sweep over (environment variable value, re-sort interval) pairs (development aid)
# usage: bash tools/gpu_env2_sweep.sh <workload> <walkers> <ENVNAME> "<val>:<resort>" ...
WL=$1; N=$2; VAR=$3; shift 3
mkdir -p gpurun_out/r02
for pair in "$@"; do
  val=${pair%%:*}; rs=${pair##*:}
  out=gpurun_out/r02/e2_${WL}_${N}_${val}_${rs}.json
  env $VAR="$val" DSIM_BENCH_POSITIONS_CACHE=/tmp/pos_${WL}_${N}.npy python bench.py --workload $WL --walkers $N --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check --resort-steps $rs > $out 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$out").read().strip().splitlines()[-1]); r=d["roofline"]
print("${WL} n=${N} ${VAR}=${val} resort ${rs}: value %.4g busy %.1f sum %.1f launches %d" % (d["value"], r["walk_busy_ms_per_step"], r["walk_sum_ms_per_step"], r["launches_per_step"]))
PY
done
