mkdir -p gpurun_out/r03
timeout -k 10 600 python bench.py --gpus 2 --backend gloo --walkers 2000000 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check > gpurun_out/r03/bench_gloo_2ranks_1gpu.json 2> gpurun_out/r03/bench_gloo_2ranks_1gpu.err
tail -c 600 gpurun_out/r03/bench_gloo_2ranks_1gpu.json; tail -5 gpurun_out/r03/bench_gloo_2ranks_1gpu.err
timeout -k 10 600 python bench.py --walkers 2000000 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check > gpurun_out/r03/bench_1rank_2m.json 2> gpurun_out/r03/bench_1rank_2m.err
python - <<PY
import json
a=json.loads(open("gpurun_out/r03/bench_gloo_2ranks_1gpu.json").read().strip().splitlines()[-1])
b=json.loads(open("gpurun_out/r03/bench_1rank_2m.json").read().strip().splitlines()[-1])
print("2 ranks: value %.4g S/N %.12f setup %s" % (a["value"], a["signal_check"]["last_pass_mean_S_over_N"], a["setup_s"]["initial_positions"]))
print("1 rank : value %.4g S/N %.12f setup %s" % (b["value"], b["signal_check"]["last_pass_mean_S_over_N"], b["setup_s"]["initial_positions"]))
PY
