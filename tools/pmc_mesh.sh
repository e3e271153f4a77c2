#!/bin/bash
# PMC passes over the mesh walk kernel only (--kernel-include-regex keeps the thousands of tiny
# fill_mesh / sort dispatches out of the counter collection).  Usage: tools/pmc_mesh.sh c5 out.json
set -e
W=$1; OUT=$2; P=/tmp/pmc_mesh_$W; mkdir -p $P
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-include-regex walk_kernel --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $P/sq -- python3 bench.py --steps 1 --warmup 0 --workload $W --walkers 1000000 --no-cpu-baseline > $P/sq.log 2>&1
timeout 400 rocprofv3 --kernel-include-regex walk_kernel --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $P/mem -- python3 bench.py --steps 1 --warmup 0 --workload $W --walkers 1000000 --no-cpu-baseline > $P/mem.log 2>&1
timeout 400 rocprofv3 --kernel-include-regex walk_kernel --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $P/ta -- python3 bench.py --steps 1 --warmup 0 --workload $W --walkers 1000000 --no-cpu-baseline > $P/ta.log 2>&1 || true
python3 tools/pmc_summary.py $P/sq $P/mem $P/ta > $OUT
python3 - "$OUT" "$P" <<PY
import json, sys
d = json.load(open(sys.argv[1])); P = sys.argv[2]
c = d[P + "/sq"]["counters"]; m = d[P + "/mem"]["counters"]; ws = 15628 * 1e4
print(sys.argv[1], d[P + "/sq"]["resources"], "dispatches", d[P + "/sq"]["dispatches"])
print("  per wave-step: VALU %.0f VMEM %.0f LDS %.0f SALU %.0f L2req %.0f LDSconflict %.0f" % (c["SQ_INSTS_VALU"] / ws, m["SQ_INSTS_VMEM_RD"] / ws, m["SQ_INSTS_LDS"] / ws, m["SQ_INSTS_SALU"] / ws, m["TCP_TCC_READ_REQ_sum"] / ws, m["SQ_LDS_BANK_CONFLICT"] / ws))
t = d.get(P + "/ta", {}).get("counters", {})
if t:
    cyc = m["GRBM_GUI_ACTIVE"] / 8  # per-XCD cycles (8 XCDs accumulate)
    print("  TA busy %.2f (of 256 TAs)  addr stalled by TC %.2f  data stalled by TC %.2f  TCP pending stall %.2f  tag conflict %.2f  vmem level/inst %.0f" % (
        t.get("TA_TA_BUSY_sum", 0) / 256 / cyc, t.get("TA_ADDR_STALLED_BY_TC_CYCLES_sum", 0) / 256 / cyc, t.get("TA_DATA_STALLED_BY_TC_CYCLES_sum", 0) / 256 / cyc,
        t.get("TCP_PENDING_STALL_CYCLES_sum", 0) / 256 / cyc, t.get("TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", 0) / 256 / cyc,
        t.get("SQ_INST_LEVEL_VMEM", 0) / max(1.0, m["SQ_INSTS_VMEM_RD"])))
print("  VALU busy %.2f  wait_any %.2f  wait_inst %.2f  active_any %.2f" % (c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (m["GRBM_GUI_ACTIVE"] / 8), c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"]))
PY
