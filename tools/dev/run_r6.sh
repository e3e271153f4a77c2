# Round-6 GPU call driver: bash tools/dev/run_r6.sh <what> ...  (several commands chained with "--")
#   tests [pytest args]   the GPU suite (or a selection)
#   ab <tag> <workload> <walkers> "<libs>" ["bench args"]   same-box A/B (tools/gpu_ab.sh)
#   pcs <tag> <lib> <workload> <walkers> [method]   PC-sampling histogram of the walk kernel (tools/pc_hist.py)
#   sh <command ...>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export ROUND=r06
mkdir -p gpurun_out/r06
FILTER="^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|socket.cpp\|Gloo\|amdgpu.ids"
run_one() {
  what=$1; shift
  case $what in
    tests) timeout -k 10 1800 python -m pytest tests -m gpu -q -x "$@" 2>&1 | grep -v "$FILTER" | tail -25 | tee gpurun_out/r06/pytest_last.txt ;;
    ab) bash tools/gpu_ab.sh "$@" | tee -a gpurun_out/r06/ab_log.txt ;;
    sh) "$@" ;;
    # final evidence of the round, part 1: correctness + the headline lines + kernel trace (-> gpurun_out/r06/final)
    final1)
      O=gpurun_out/r06/final; mkdir -p $O
      timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt
      timeout -k 10 1800 python -m pytest tests -m gpu -q 2>&1 | grep -v "$FILTER" | tail -15 > $O/pytest_gpu.txt
      timeout -k 10 900 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err
      timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check --no-end-to-end > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
      find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/c5b_bench_kernel_stats.csv \;
      rm -rf $O/stats
      timeout -k 10 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command_line.json 2> $O/bench_driver_command.err
      timeout -k 10 600 python bench.py --gpus 2 --backend gloo --walkers 2000000 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $O/bench_gloo_2ranks_1gpu_line.json 2> $O/bench_gloo_2ranks.err
      DSIM_DEVICES=0,0 timeout -k 10 600 python bench.py --gpus 2 --mode inprocess --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $O/bench_inprocess_2shards_1gpu_line.json 2> $O/bench_inprocess.err
      tail -3 $O/pytest_gpu.txt; tail -2 $O/smoke.txt ;;
    # part 2: counters (separate --pmc passes, --kernel-trace only), stage timers, traversal statistics, rates by size and M
    final2)
      O=gpurun_out/r06/final; mkdir -p $O
      timeout -k 10 1500 python3 tools/pmc_profile.py c5b --walkers 10000000 --tag r06 --passes sq,mem,lds,mix,mix2,fetch,write > $O/pmc_c5b.txt 2>&1
      for wl in c2 c3 c4; do timeout -k 10 500 python3 tools/pmc_profile.py $wl --walkers 1000000 --tag r06 --passes sq,mem > $O/pmc_$wl.txt 2>&1; done
      cp profiles/kernel_counters.json profiles/r06_*_pmc.json $O/ 2>/dev/null
      DSIM_LIB=$PWD/disimpy_amd/csrc/libdsim_hip_timers.so timeout -k 5 300 python tools/mesh_timers.py c5b 1000000 > $O/c5b_phase_timers.txt 2>&1
      DSIM_LIB=$PWD/disimpy_amd/csrc/libdsim_hip_stats.so timeout -k 5 300 python tools/mesh_stats.py c5b > $O/c5b_traversal_stats.txt 2>&1
      timeout -k 5 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace1250k -- python3 bench.py --workload c5b --walkers 1250000 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check --no-end-to-end > $O/trace1250k.json 2> $O/trace1250k.err
      python tools/timeline.py /tmp/trace1250k > $O/two_stream_timeline_1250k.txt 2>&1
      bash tools/dev/rates.sh "10000000 4" "10000000 8" "10000000 32" "5000000 1" "2500000 1" "1250000 1" "1250000 32" > $O/rates_by_size_and_measurements.txt 2>&1
      timeout -k 5 900 python tools/dev/parts_rate.py 8 c5b 2>&1 | grep "^part" > $O/spatial_parts_of_8.txt
      # the two headline lines once more, now that profiles/kernel_counters.json belongs to this build
      timeout -k 10 900 python bench.py > $O/bench_line_with_counters.json 2> $O/bench_line_with_counters.err
      python tools/codeobj_meta.py > $O/codeobj_meta.txt 2>&1
      cat $O/rates_by_size_and_measurements.txt $O/spatial_parts_of_8.txt; tail -30 $O/pmc_c5b.txt ;;
    # last call of the round: the suite and the two headline lines on the committed tree
    final4)
      O=gpurun_out/r06/final; mkdir -p $O
      timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt
      timeout -k 10 1800 python -m pytest tests -m gpu -q 2>&1 | grep -v "$FILTER" | tail -15 > $O/pytest_gpu.txt
      timeout -k 10 900 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err
      timeout -k 10 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command_line.json 2> $O/bench_driver_command.err
      tail -3 $O/pytest_gpu.txt; tail -2 $O/smoke.txt ;;
    # the counter passes alone + the headline lines on the build they belong to
    final3)
      O=gpurun_out/r06/final; mkdir -p $O
      timeout -k 10 1500 python3 tools/pmc_profile.py c5b --walkers 10000000 --tag r06 --passes sq,mem,lds,mix,mix2,fetch,write > $O/pmc_c5b.txt 2>&1
      for wl in c2 c3 c4; do timeout -k 10 500 python3 tools/pmc_profile.py $wl --walkers 1000000 --tag r06 --passes sq,mem > $O/pmc_$wl.txt 2>&1; done
      cp profiles/kernel_counters.json profiles/r06_*_pmc.json $O/ 2>/dev/null
      timeout -k 10 900 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err
      timeout -k 10 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command_line.json 2> $O/bench_driver_command.err
      timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check --no-end-to-end > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
      find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/c5b_bench_kernel_stats.csv \;
      rm -rf $O/stats
      tail -32 $O/pmc_c5b.txt; tail -12 $O/pmc_c4.txt ;;
    # kernel-trace timeline of the last pass: timeline <tag> <walkers> [bench args]
    timeline)
      tag=$1; n=$2; shift; shift
      O=/tmp/trace_$tag; rm -rf $O
      timeout -k 5 500 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $PWD/bench.py --workload c5b --walkers $n --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check --no-end-to-end "$@" > gpurun_out/r06/trace_$tag.json 2> gpurun_out/r06/trace_$tag.err
      python tools/timeline.py $O > gpurun_out/r06/timeline_$tag.txt 2>&1; head -12 gpurun_out/r06/timeline_$tag.txt; tail -12 gpurun_out/r06/timeline_$tag.txt ;;
    # bench lines of the three multi-device paths on the one GPU: default, in-process handle over 0,0, two gloo ranks
    benches)
      O=gpurun_out/r06; tag=${1:-x}
      timeout -k 10 900 python bench.py > $O/bench_default_$tag.json 2> $O/bench_default_$tag.err; echo "default rc=$?"
      DSIM_DEVICES=0,0 timeout -k 10 600 python bench.py --gpus 2 --mode inprocess --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $O/bench_inprocess2_$tag.json 2> $O/bench_inprocess2_$tag.err; echo "inprocess rc=$?"
      timeout -k 10 600 python bench.py --gpus 2 --backend gloo --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $O/bench_gloo2_$tag.json 2> $O/bench_gloo2_$tag.err; echo "gloo rc=$?"
      python - <<PY
import json
for n in ("default","inprocess2","gloo2"):
    try:
        d=json.loads(open("$O/bench_%s_$tag.json"%n).read().strip().splitlines()[-1])
        s=d.get("signal_rel_err",{}).get("sample_of_timed_run",{})
        print(n, "value %.4g"%d["value"], "walk %.4g"%d["walk_kernel_spin_steps_per_s"], "setup", {k: round(v,2) for k,v in d["setup_s"].items()}, "sample", s.get("result") if isinstance(s,dict) else s, "e2e", d.get("end_to_end_s"))
        if "secondary" in d: print("  secondary", {k:(round(v["value"]/1e10,3), v.get("walkers")) for k,v in d["secondary"].items()})
    except Exception as e:
        print(n, "FAILED", e); print(open("$O/bench_%s_$tag.err"%n).read()[-1500:])
PY
      ;;
    pcs)
      tag=$1; lib=$2; wl=$3; n=$4; method=${5:-stochastic}
      O=/tmp/pcs_$tag; rm -rf $O
      unit=cycles; interval=1048576
      if [ "$method" = host_trap ]; then unit=time; interval=100; fi
      DSIM_LIB=$PWD/disimpy_amd/csrc/$lib timeout -k 10 420 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $method --pc-sampling-unit $unit --pc-sampling-interval $interval --kernel-trace --output-format csv -d $O -- python3 $PWD/bench.py --steps 1 --warmup 0 --workload $wl --walkers $n --no-cpu-baseline --no-secondary --no-signal-check > gpurun_out/r06/pcs_$tag.json 2> gpurun_out/r06/pcs_$tag.err
      echo "pcs rc=$?"; tail -5 gpurun_out/r06/pcs_$tag.err
      find $O -type f | head -20; du -sh $O
      python3 tools/pc_hist.py $O gpurun_out/r06/pcs_${tag} ;;
  esac
}
args=()
for a in "$@"; do
  if [ "$a" = "--" ]; then run_one "${args[@]}"; args=(); else args+=("$a"); fi
done
[ ${#args[@]} -gt 0 ] && run_one "${args[@]}"
