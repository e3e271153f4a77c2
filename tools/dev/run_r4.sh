# Round-4 GPU call driver: bash tools/dev/run_r4.sh <what> ...   (one parametrised script instead of one file per call)
#   tests [pytest args]      the GPU suite (or a selection), tail into gpurun_out/r04/pytest_<tag>.txt
#   ab <tag> <workload> <walkers> "<libs>" ["bench args"]   same-box A/B (tools/gpu_ab.sh)
# Several commands may be chained with "--":  run_r4.sh tests -k mesh -- ab a1 c5b 10000000 "libdsim_hip_r03.so libdsim_hip.so"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
run_one() {
  what=$1; shift
  case $what in
    tests) timeout -k 10 1500 python -m pytest tests -m gpu -q -x "$@" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|socket.cpp\|Gloo\|amdgpu.ids" | tail -25 | tee gpurun_out/r04/pytest_last.txt ;;
    ab) bash tools/gpu_ab.sh "$@" | tee -a gpurun_out/r04/ab_log.txt ;;
    sh) "$@" ;;
    # final evidence of the round, part 1: correctness + the headline lines + kernel trace (-> gpurun_out/r04/final)
    final1)
      O=gpurun_out/r04/final; mkdir -p $O
      timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
      timeout -k 10 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|socket.cpp\|Gloo\|amdgpu.ids" | tail -15 > $O/pytest_gpu.txt
      timeout -k 10 700 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err
      timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
      find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/c5b_bench_kernel_stats.csv \;
      rm -rf $O/stats
      timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command_line.json 2> $O/bench_driver_command.err
      timeout -k 10 600 python bench.py --gpus 2 --backend gloo --walkers 2000000 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $O/bench_gloo_2ranks_1gpu_line.json 2> $O/bench_gloo_2ranks.err
      tail -3 $O/pytest_gpu.txt; tail -2 $O/smoke.txt ;;
    # part 2: counters (separate --pmc passes, --kernel-trace only), stage timers, traversal statistics, rates by size and M
    final2)
      O=gpurun_out/r04/final; mkdir -p $O
      timeout -k 10 1200 python3 tools/pmc_profile.py c5b --walkers 10000000 --tag r04 --passes sq,mem,lds,fetch,write > $O/pmc_c5b.txt 2>&1
      for wl in c2 c3 c4; do timeout -k 10 500 python3 tools/pmc_profile.py $wl --walkers 1000000 --tag r04 --passes sq,mem > $O/pmc_$wl.txt 2>&1; done
      cp profiles/kernel_counters.json profiles/r04_*_pmc.json $O/ 2>/dev/null
      DSIM_LIB=$PWD/disimpy_amd/csrc/libdsim_hip_timers.so timeout -k 5 300 python tools/mesh_timers.py c5b 1000000 > $O/c5b_phase_timers.txt 2>&1
      DSIM_LIB=$PWD/disimpy_amd/csrc/libdsim_hip_stats.so timeout -k 5 300 python tools/mesh_stats.py c5b > $O/c5b_traversal_stats.txt 2>&1
      timeout -k 5 500 rocprofv3 --kernel-trace --output-format csv -d $O/trace1250k -- python3 bench.py --workload c5b --walkers 1250000 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check > $O/trace1250k.json 2> $O/trace1250k.err
      python tools/timeline.py $O/trace1250k > $O/two_stream_timeline_1250k.txt 2>&1
      rm -rf $O/trace1250k
      timeout -k 10 700 python bench.py > $O/bench_line_with_counters.json 2> $O/bench_line_with_counters.err
      for cfg in "10000000 4" "10000000 8" "10000000 32" "5000000 1" "2500000 1" "1250000 1" "1250000 32"; do set -- $cfg
        timeout -k 5 400 python bench.py --workload c5b --walkers $1 --measurements $2 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5b walkers $1 M=$2: value %.4g walk-kernel %.4g' % (d['value'], d['walk_kernel_spin_steps_per_s']))"
      done > $O/rates_by_size_and_measurements.txt 2>&1
      # the two headline lines once more, now that profiles/kernel_counters.json belongs to this build
      timeout -k 10 700 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err
      timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command_line.json 2> $O/bench_driver_command.err
      python tools/dev/fillcheck.py 30000 > $O/fill_neuron_30000_points_vs_oracle.txt 2>&1
      cat $O/rates_by_size_and_measurements.txt; tail -25 $O/pmc_c5b.txt ;;
  esac
}
args=()
for a in "$@"; do
  if [ "$a" = "--" ]; then run_one "${args[@]}"; args=(); else args+=("$a"); fi
done
[ ${#args[@]} -gt 0 ] && run_one "${args[@]}"
