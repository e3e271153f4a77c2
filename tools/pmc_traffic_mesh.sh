#!/bin/bash
# HBM traffic of the mesh walk kernel: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (kernel filter),
# summed over the launches of one bench step.  Usage: tools/pmc_traffic_mesh.sh c5 out.json
set -e
W=$1; OUT=$2; P=/tmp/pmc_traffic_$W; mkdir -p $P
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 500 rocprofv3 --kernel-include-regex walk_kernel --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/fetch -- python3 bench.py --steps 1 --warmup 0 --workload $W --no-cpu-baseline > $P/fetch.log 2>&1
timeout 500 rocprofv3 --kernel-include-regex walk_kernel --pmc WRITE_SIZE --kernel-trace --output-format csv -d $P/write -- python3 bench.py --steps 1 --warmup 0 --workload $W --no-cpu-baseline > $P/write.log 2>&1
python3 tools/pmc_summary.py $P/fetch $P/write > $OUT
python3 - "$OUT" "$P" <<PY
import json, sys
d = json.load(open(sys.argv[1])); P = sys.argv[2]
f = d[P + "/fetch"]; w = d[P + "/write"]
n = f["dispatches"]
rd = f["counters"]["FETCH_SIZE"] * 1024 * 2   # gfx950: FETCH_SIZE reports half the bytes (profiles/hbm_traffic.json)
wr = w["counters"]["WRITE_SIZE"] * 1024
print("launches %d: read %.3f GB, written %.3f GB per bench step; per launch %.3f GB" % (n, rd / 1e9, wr / 1e9, (rd + wr) / n / 1e9))
PY
