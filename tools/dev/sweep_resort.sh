cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for cfg in "10000000 128" "10000000 256" "10000000 384" "10000000 512" "1250000 64" "1250000 128" "1250000 192" "1250000 256"; do set -- $cfg
DSIM_BENCH_POSITIONS_CACHE=/tmp/pos_c5b_$1.npy timeout -k 5 400 python bench.py --workload c5b --walkers $1 --resort-steps $2 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5b walkers $1 resort $2: value %.4g walk-kernel %.4g' % (d['value'], d['walk_kernel_spin_steps_per_s']))"
done
