# Round-5 GPU call driver: bash tools/dev/run_r5.sh <what> ...  (several commands chained with "--")
#   tests [pytest args]   the GPU suite (or a selection)
#   ab <tag> <workload> <walkers> "<libs>" ["bench args"]   same-box A/B (tools/gpu_ab.sh)
#   pcs <tag> <lib> <workload> <walkers> [method]   PC-sampling histogram of the walk kernel (tools/pc_hist.py)
#   sh <command ...>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export ROUND=r05
mkdir -p gpurun_out/r05
FILTER="^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|socket.cpp\|Gloo\|amdgpu.ids"
run_one() {
  what=$1; shift
  case $what in
    tests) timeout -k 10 1800 python -m pytest tests -m gpu -q -x "$@" 2>&1 | grep -v "$FILTER" | tail -25 | tee gpurun_out/r05/pytest_last.txt ;;
    ab) bash tools/gpu_ab.sh "$@" | tee -a gpurun_out/r05/ab_log.txt ;;
    sh) "$@" ;;
    pcs)
      tag=$1; lib=$2; wl=$3; n=$4; method=${5:-stochastic}
      O=/tmp/pcs_$tag; rm -rf $O
      unit=cycles; interval=1048576
      if [ "$method" = host_trap ]; then unit=time; interval=100; fi
      DSIM_LIB=$PWD/disimpy_amd/csrc/$lib timeout -k 10 420 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $method --pc-sampling-unit $unit --pc-sampling-interval $interval --kernel-trace --output-format csv -d $O -- python3 $PWD/bench.py --steps 1 --warmup 0 --workload $wl --walkers $n --no-cpu-baseline --no-secondary --no-signal-check > gpurun_out/r05/pcs_$tag.json 2> gpurun_out/r05/pcs_$tag.err
      echo "pcs rc=$?"; tail -5 gpurun_out/r05/pcs_$tag.err
      find $O -type f | head -20; du -sh $O
      python3 tools/pc_hist.py $O gpurun_out/r05/pcs_${tag} ;;
  esac
}
args=()
for a in "$@"; do
  if [ "$a" = "--" ]; then run_one "${args[@]}"; args=(); else args+=("$a"); fi
done
[ ${#args[@]} -gt 0 ] && run_one "${args[@]}"
