# same-box sweep over the re-sort interval (development aid)
# usage: bash tools/gpu_resort_sweep.sh <workload> <walkers> <interval> ...
WL=$1; N=$2; shift 2
mkdir -p gpurun_out/r02
for rs in "$@"; do
  out=gpurun_out/r02/rs_${WL}_${N}_$rs.json
  DSIM_BENCH_POSITIONS_CACHE=/tmp/pos_${WL}_${N}.npy python bench.py --workload $WL --walkers $N --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check --resort-steps $rs > $out 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$out").read().strip().splitlines()[-1]); r=d["roofline"]
print("${WL} n=${N} resort $rs: value %.4g busy %.1f sum %.1f launches %d" % (d["value"], r["walk_busy_ms_per_step"], r["walk_sum_ms_per_step"], r["launches_per_step"]))
PY
done
