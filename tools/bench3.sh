#!/bin/bash
# GPU box helper: full GPU test-suite, then the C2 / C3 / C5 / C4 bench lines (value, kernel rate).
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.txt 2>&1; tail -2 gpurun_out/pytest_gpu.txt
for w in ${@:-c2 c5 c4}; do python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:40], '%.4g'%d['value'], '%.4g'%d['kernel_spin_steps_per_s'])"; done
