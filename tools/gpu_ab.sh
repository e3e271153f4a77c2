# Same-box A/B of kernel variants: every library in LIBS runs the same bench line (MI355X boxes differ by
# several per cent between calls, so only numbers from ONE call compare).  Initial positions are computed
# once per (workload, walkers) and cached in /tmp.
# usage: bash tools/gpu_ab.sh <tag> <workload> <walkers> "<lib1> <lib2> ..." ["extra bench args"]
TAG=$1; WL=$2; N=$3; LIBS=$4; EXTRA=$5
mkdir -p gpurun_out/${ROUND:-r04}
for rep in $(seq 1 ${REPS:-2}); do for lib in $LIBS; do
  name=$(basename $lib .so)
  out=gpurun_out/${ROUND:-r04}/ab_${TAG}_${WL}_${N}_${name}_${rep}
  DSIM_BENCH_POSITIONS_CACHE=/tmp/pos_${WL}_${N}.npy DSIM_LIB_OLD_ABI=1 DSIM_LIB=$PWD/disimpy_amd/csrc/$lib timeout -k 5 400 python bench.py --workload $WL --walkers $N --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check $EXTRA > $out.json 2> $out.err
  python - <<PY
import json
try:
    d=json.loads(open("$out.json").read().strip().splitlines()[-1])
    print("${TAG} ${WL} n=${N} ${name} rep${rep}: value %.4g  walk-kernel %.4g  S/N %.9f" % (d["value"], d["walk_kernel_spin_steps_per_s"], d["signal_check"]["last_pass_mean_S_over_N"]))
except Exception as e:
    print("${TAG} ${name} FAILED", e); print(open("$out.err").read()[-1500:])
PY
done; done
