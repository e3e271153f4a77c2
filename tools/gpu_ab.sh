# A/B of the mesh kernel variants on the GPU box: bench lines for the given --mesh-traversal values
# usage: bash tools/gpu_ab.sh <tag> "<walkers list>" "<workloads>" "<traversals>"
TAG=$1; WALKERS=${2:-"1000000 10000000"}; WLS=${3:-"c5b"}; TRS=${4:-"3 2"}
mkdir -p gpurun_out/r02
for wl in $WLS; do for n in $WALKERS; do for tr in $TRS; do
  python bench.py --workload $wl --walkers $n --steps 2 --warmup 1 --mesh-traversal $tr --no-cpu-baseline --no-secondary --no-signal-check > gpurun_out/r02/ab_${TAG}_${wl}_${n}_tr${tr}.json 2> gpurun_out/r02/ab_${TAG}_${wl}_${n}_tr${tr}.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r02/ab_${TAG}_${wl}_${n}_tr${tr}.json").read().strip().splitlines()[-1])
    print("${TAG} ${wl} n=${n} traversal=${tr}: value %.4g  walk-kernel %.4g  S/N %.6f" % (d["value"], d["walk_kernel_spin_steps_per_s"], d["signal_check"]["last_pass_mean_S_over_N"]))
except Exception as e:
    print("${TAG} ${wl} ${n} ${tr} FAILED", e); print(open("gpurun_out/r02/ab_${TAG}_${wl}_${n}_tr${tr}.err").read()[-1500:])
PY
done; done; done
