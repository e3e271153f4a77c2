# bench lines for a list of extra-argument variants: bash tools/gpu_sweep.sh <tag> <workload> <walkers> "<args1>" "<args2>" ...
TAG=$1; WL=$2; N=$3; shift 3
mkdir -p gpurun_out/r02
i=0
for extra in "$@"; do
  i=$((i+1))
  python bench.py --workload $WL --walkers $N --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check $extra > gpurun_out/r02/sw_${TAG}_$i.json 2> gpurun_out/r02/sw_${TAG}_$i.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r02/sw_${TAG}_$i.json").read().strip().splitlines()[-1])
    print("${TAG} ${WL} n=${N} [${extra}]: value %.4g  walk-kernel %.4g  launches %d  S/N %.6f" % (d["value"], d["walk_kernel_spin_steps_per_s"], d["roofline"]["launches_per_step"], d["signal_check"]["last_pass_mean_S_over_N"]))
except Exception as e:
    print("${TAG} [${extra}] FAILED", e); print(open("gpurun_out/r02/sw_${TAG}_$i.err").read()[-1500:])
PY
done
