# re-sweep of the launch tunables after the round-5 instruction cuts (same box): bash tools/dev/sweep_r5b.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
one() { # workload walkers "env" "bench args"
  for rep in 1 2; do
  env $3 DSIM_BENCH_POSITIONS_CACHE=/tmp/pos_$1_$2.npy timeout -k 5 400 python bench.py --workload $1 --walkers $2 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check --no-end-to-end $4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 n=$2 [$3 $4] rep$rep: value %.4g walk-kernel %.4g' % (d['value'], d['walk_kernel_spin_steps_per_s']))"
  done
}
one c5b 10000000 "A=0" ""
one c5b 10000000 "A=0" "--resort-steps 256"
one c5b 10000000 "A=0" "--resort-steps 384"
one c5b 10000000 "A=0" "--resort-steps 768"
one c5b 10000000 "A=0" "--resort-steps 1024"
one c5b 10000000 "DSIM_XCD_RUN=2" ""
one c5b 10000000 "DSIM_XCD_RUN=4" ""
one c5b 10000000 "DSIM_GROUPS=3" ""
one c5b 10000000 "DSIM_GROUPS=4" ""
one c5b 1250000 "A=0" ""
one c5b 1250000 "A=0" "--resort-steps 64"
one c5b 1250000 "A=0" "--resort-steps 96"
one c5b 1250000 "A=0" "--resort-steps 192"
one c5b 1250000 "A=0" "--resort-steps 256"
one c5b 1250000 "DSIM_GROUPS=3" ""
one c5b 1250000 "DSIM_GROUPS=4" ""
one c5b 1250000 "DSIM_XCD_RUN=2" ""
one c5b 1250000 "DSIM_XCD_RUN=4" ""
