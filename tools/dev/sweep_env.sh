# bash tools/dev/sweep_env.sh <workload> <walkers> "<ENV=val ...>" ... : one bench line per environment setting (same box)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
WL=$1; N=$2; shift; shift
for setting in "$@"; do
  for rep in 1 2; do
  env $setting DSIM_BENCH_POSITIONS_CACHE=/tmp/pos_${WL}_$N.npy timeout -k 5 400 python bench.py --workload $WL --walkers $N --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check --no-end-to-end 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$WL n=$N [$setting] rep$rep: value %.4g walk-kernel %.4g' % (d['value'], d['walk_kernel_spin_steps_per_s']))"
  done
done
