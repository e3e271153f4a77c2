#!/bin/bash
# SQ counters of the C2 launch (free diffusion, 32 measurements): tools/pmc_c2.sh out.json
set -e
OUT=$1; P=/tmp/pmc_c2; mkdir -p $P
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-include-regex walk_kernel --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $P/sq -- python3 tools/traffic_workload.py > $P/sq.log 2>&1
timeout 300 rocprofv3 --kernel-include-regex walk_kernel --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $P/sq2 -- python3 tools/traffic_workload.py > $P/sq2.log 2>&1
python3 tools/pmc_summary.py $P/sq $P/sq2 > $OUT
python3 - "$OUT" "$P" <<PY
import json, sys
d = json.load(open(sys.argv[1])); P = sys.argv[2]
c = d[P + "/sq"]["counters"]; m = d[P + "/sq2"]["counters"]; ws = 2 * 15628 * 1e3
print("per wave-step: VALU %.0f SALU %.0f LDS %.0f" % (c["SQ_INSTS_VALU"] / ws, m["SQ_INSTS_SALU"] / ws, m["SQ_INSTS_LDS"] / ws))
print("VALU busy %.2f  wait_any %.2f" % (c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (m["GRBM_GUI_ACTIVE"] / 8), c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]))
PY
