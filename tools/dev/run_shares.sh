mkdir -p gpurun_out/r03
O=gpurun_out/r03/shares.txt; : > $O
for n in 5000000 2500000 1250000; do
  timeout -k 5 300 python bench.py --workload c5b --walkers $n --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check > gpurun_out/r03/sh_$n.json 2> gpurun_out/r03/sh_$n.err
  python - <<PY >> $O
import json
try:
    d=json.loads(open("gpurun_out/r03/sh_$n.json").read().strip().splitlines()[-1])
    print("c5b n=$n M=1: value %.4g  walk-kernel %.4g  launches/step %s" % (d["value"], d["walk_kernel_spin_steps_per_s"], d["roofline"]["launches_per_step"]))
except Exception as e:
    print("$n FAILED", e)
PY
done
cat $O
