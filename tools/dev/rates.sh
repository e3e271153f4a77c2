# bash tools/dev/rates.sh "<walkers> <M>" ... : one bench line per configuration of the headline workload
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for cfg in "$@"; do set -- $cfg
  DSIM_BENCH_POSITIONS_CACHE=/tmp/pos_c5b_$1.npy timeout -k 5 500 python bench.py --workload c5b --walkers $1 --measurements $2 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-signal-check --no-end-to-end 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5b walkers $1 M=$2: value %.4g walk-kernel %.4g' % (d['value'], d['walk_kernel_spin_steps_per_s']))"
done
