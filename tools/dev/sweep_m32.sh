cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for cfg in "10000000 32 0" "10000000 32 64" "10000000 32 96" "10000000 32 128" "10000000 8 0" "10000000 8 96" "1250000 32 0" "1250000 32 64" "1250000 32 96"; do set -- $cfg
DSIM_BENCH_POSITIONS_CACHE=/tmp/pos_c5b_$1.npy timeout -k 5 500 python bench.py --workload c5b --walkers $1 --measurements $2 --resort-steps $3 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check --no-end-to-end 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5b walkers $1 M=$2 resort $3: value %.4g walk-kernel %.4g launches %d' % (d['value'], d['walk_kernel_spin_steps_per_s'], d['roofline']['launches_per_step']))"
done
