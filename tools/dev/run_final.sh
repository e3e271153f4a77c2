# final evidence of a round: everything under gpurun_out/r03/final, copied into profiles/ afterwards
O=gpurun_out/r03/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
timeout -k 10 900 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -15 > $O/pytest_gpu.txt
timeout -k 10 600 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/c5b_bench_kernel_stats.csv \;
rm -rf $O/stats
timeout -k 10 1000 python3 tools/pmc_profile.py c5b --walkers 10000000 --tag r03 --passes sq,mem,fetch,write > $O/pmc_c5b.txt 2>&1
for wl in c2 c3 c4 c5; do timeout -k 10 500 python3 tools/pmc_profile.py $wl --walkers 1000000 --tag r03 --passes sq,mem > $O/pmc_$wl.txt 2>&1; done
cp profiles/kernel_counters.json profiles/r03_*_pmc.json $O/ 2>/dev/null
DSIM_LIB=$PWD/disimpy_amd/csrc/libdsim_hip_timers.so timeout -k 5 300 python tools/mesh_timers.py c5b 1000000 > $O/c5b_phase_timers.txt 2>&1
DSIM_LIB=$PWD/disimpy_amd/csrc/libdsim_hip_stats.so timeout -k 5 300 python tools/mesh_stats.py c5b > $O/c5b_traversal_stats.txt 2>&1
timeout -k 5 500 rocprofv3 --kernel-trace --output-format csv -d $O/trace1250k -- python3 bench.py --workload c5b --walkers 1250000 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check > $O/trace1250k.json 2> $O/trace1250k.err
python tools/timeline.py $O/trace1250k > $O/two_stream_timeline_1250k.txt 2>&1
rm -rf $O/trace1250k
timeout -k 10 400 python bench.py --no-cpu-baseline --no-secondary --no-signal-check > $O/bench_after_counters.json 2> $O/bench_after_counters.err
timeout -k 10 600 python bench.py --gpus 2 --backend gloo --walkers 2000000 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $O/bench_gloo_2ranks_1gpu_line.json 2> $O/bench_gloo_2ranks.err
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command_line.json 2> $O/bench_driver_command.err
ls -la $O; tail -3 $O/pytest_gpu.txt; cat $O/smoke.txt | tail -2; python - <<PY
import json
for f in ("bench_default_line","bench_after_counters","bench_gloo_2ranks_1gpu_line","bench_driver_command_line"):
    try:
        d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1]); r=d["roofline"]
        print(f, d["value"], r["frac"], r["counters_match_lib"], r["valu"]["valu_busy_measured"], r["traffic"])
    except Exception as e: print(f, "FAILED", e)
PY
