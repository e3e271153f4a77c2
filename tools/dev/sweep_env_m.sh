# bash tools/dev/sweep_env_m.sh <workload> <walkers> <M> "<ENV=val ...>" ... : like sweep_env.sh with --measurements M
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
WL=$1; N=$2; M=$3; shift; shift; shift
for setting in "$@"; do
  for rep in 1 2; do
  env $setting DSIM_BENCH_POSITIONS_CACHE=/tmp/pos_${WL}_$N.npy timeout -k 5 500 python bench.py --workload $WL --walkers $N --measurements $M --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check --no-end-to-end 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$WL n=$N M=$M [$setting] rep$rep: value %.4g walk-kernel %.4g  S/N %.9f' % (d['value'], d['walk_kernel_spin_steps_per_s'], d['signal_check']['last_pass_mean_S_over_N']))"
  done
done
