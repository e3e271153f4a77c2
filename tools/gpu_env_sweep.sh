# same-box sweep over values of one environment variable (development aid)
# usage: bash tools/gpu_env_sweep.sh <tag> <workload> <walkers> <ENVNAME> "<val1>" "<val2>" ...
TAG=$1; WL=$2; N=$3; VAR=$4; shift 4
mkdir -p gpurun_out/r02
i=0
for val in "$@"; do
  i=$((i+1)); out=gpurun_out/r02/env_${TAG}_$i
  env $VAR="$val" DSIM_BENCH_POSITIONS_CACHE=/tmp/pos_${WL}_${N}.npy python bench.py --workload $WL --walkers $N --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check > $out.json 2> $out.err
  python - <<PY
import json
try:
    d=json.loads(open("$out.json").read().strip().splitlines()[-1])
    print("${TAG} ${WL} n=${N} ${VAR}=${val}: value %.4g  walk-kernel %.4g  S/N %.9f" % (d["value"], d["walk_kernel_spin_steps_per_s"], d["signal_check"]["last_pass_mean_S_over_N"]))
except Exception as e:
    print("${TAG} ${val} FAILED", e); print(open("$out.err").read()[-1500:])
PY
done
