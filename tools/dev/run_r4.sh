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
  esac
}
args=()
for a in "$@"; do
  if [ "$a" = "--" ]; then run_one "${args[@]}"; args=(); else args+=("$a"); fi
done
[ ${#args[@]} -gt 0 ] && run_one "${args[@]}"
