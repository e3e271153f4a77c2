"""rocprofv3 counter passes over the walk kernel of one bench workload (GPU box).

    python3 tools/pmc_profile.py c5b [--walkers 1000000] [--tag r02] [--bench-args ...]

Runs bench.py (one step, no CPU legs) under rocprofv3 four times - two SQ passes, FETCH_SIZE,
WRITE_SIZE: separate passes as MI355X_MICROARCH.md prescribes, --kernel-trace only - restricted to
the walk kernel, writes the raw per-kernel sums to profiles/<tag>_<workload>_pmc.json and the
derived per-wavefront-step figures into profiles/kernel_counters.json (read by bench.py for
`roofline`).  This process never touches the GPU itself; rocprofv3 runs `python3 bench.py`
directly.
"""
import argparse
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PASSES = {
    "sq": "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY "
          "SQ_ACTIVE_INST_ANY",
    "mem": "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT "
           "TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE",
    # LDS: cycles the LDS array was busy and how many of them were bank-conflict replays (MI355X_MICROARCH.md, LDS:
    # SQ_LDS_BANK_CONFLICT = extra cycles, SQ_LDS_IDX_ACTIVE = all LDS-array cycles)
    "lds": "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_INSTS_LDS "
           "SQ_WAVE_CYCLES",
    # instruction mix of the vector ALU: which share of the instructions are fp64 / 64-bit integer (quarter-rate class)
    "mix": "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 "
           "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT",
    "mix2": "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU "
            "SQ_INSTS_SMEM SQ_INSTS_LDS",
    "fetch": "FETCH_SIZE",
    "write": "WRITE_SIZE",
}


def summarise(d, match):
    agg = collections.defaultdict(float)
    disp = set()
    info = {}
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if match in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"])
                disp.add(r["Dispatch_Id"])
                info = {k: r[k] for k in ("VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size")
                        if k in r}
                info["kernel"] = r["Kernel_Name"].split("(")[0]
    ms = 0.0
    calls = 0
    for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if match in r["Kernel_Name"]:
                ms += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
                calls += 1
    return {"counters": dict(agg), "dispatches": len(disp), "resources": info, "kernel_ms_total": ms,
            "kernel_calls": calls}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload")
    ap.add_argument("--walkers", type=int, default=1_000_000)
    ap.add_argument("--tag", default="r05")
    ap.add_argument("--match", default="walk_kernel")
    ap.add_argument("--passes", default="sq,mem,fetch,write")
    ap.add_argument("--bench-args", default="")
    args = ap.parse_args()
    env = dict(os.environ, TMPDIR="/tmp")
    raw = {}
    for name in args.passes.split(","):
        out = "/tmp/pmc_%s_%s" % (args.workload, name)
        subprocess.run(["rm", "-rf", out])
        cmd = (["rocprofv3", "--kernel-include-regex", args.match, "--pmc"] + PASSES[name].split()
               + ["--kernel-trace", "--output-format", "csv", "-d", out, "--", "python3", os.path.join(ROOT, "bench.py"),
                  "--steps", "1", "--warmup", "0", "--workload", args.workload, "--walkers", str(args.walkers),
                  "--no-cpu-baseline", "--no-secondary", "--no-signal-check", "--no-end-to-end"] + args.bench_args.split())
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            print("pass %s failed:\n%s\n%s" % (name, r.stdout[-2000:], r.stderr[-2000:]), file=sys.stderr)
            continue
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        raw[name] = summarise(out, args.match)
        if line:
            d = json.loads(line[-1])
            raw[name]["bench"] = {k: d.get(k) for k in ("value", "config", "walk_kernel_spin_steps_per_s")}
            raw[name]["bench"]["streams"] = (d.get("roofline") or {}).get("streams", 1)
    dst = os.path.join(ROOT, "profiles", "%s_%s_pmc.json" % (args.tag, args.workload))
    json.dump(raw, open(dst, "w"), indent=1)
    T = (raw.get("sq", {}).get("bench", {}).get("config") or {}).get("time_steps") or 10000
    ws = -(-args.walkers // 64) * T
    import hashlib

    sys.path.insert(0, ROOT)
    from disimpy_amd import _lib  # the path only: this process never calls into the library

    M = 1
    for tok_i, tok in enumerate(args.bench_args.split()):
        if tok == "--measurements":
            M = int(args.bench_args.split()[tok_i + 1])
    entry = {"walkers": args.walkers, "time_steps": T, "source": os.path.relpath(dst, ROOT), "measurements": M,
             "bench_args": args.bench_args,
             # the build the counters belong to: bench.py quotes them only when it runs this very library
             "lib_sha256": hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()}
    if "sq" in raw:
        c = raw["sq"]["counters"]
        entry["kernel"] = raw["sq"]["resources"].get("kernel")
        entry["valu_per_wave_step"] = c["SQ_INSTS_VALU"] / ws
        entry["wait_any_frac"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
        entry["kernel_ms_per_pass_profiled"] = raw["sq"]["kernel_ms_total"]
    if "mem" in raw:
        m = raw["mem"]["counters"]
        entry["salu_per_wave_step"] = m["SQ_INSTS_SALU"] / ws
        entry["vmem_rd_per_wave_step"] = m["SQ_INSTS_VMEM_RD"] / ws
        entry["lds_per_wave_step"] = m["SQ_INSTS_LDS"] / ws
        entry["l2_read_req_per_wave_step"] = m["TCP_TCC_READ_REQ_sum"] / ws
        if "sq" in raw:  # SQ_ACTIVE_INST_VALU counts quad-cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs
            entry["valu_busy"] = raw["sq"]["counters"]["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (m["GRBM_GUI_ACTIVE"] / 8)
    if "mix" in raw:
        c = raw["mix"]["counters"]
        entry["valu_mix_per_wave_step"] = {k.replace("SQ_INSTS_VALU_", "").lower(): v / ws for k, v in c.items() if k != "SQ_INSTS_VALU"}
        entry["valu_mix_per_wave_step"]["all"] = c.get("SQ_INSTS_VALU", 0.0) / ws
    if "mix2" in raw:
        c = raw["mix2"]["counters"]
        entry.setdefault("valu_mix_per_wave_step", {}).update(
            {k.replace("SQ_INSTS_VALU_", "").replace("SQ_INSTS_", "").lower(): v / ws for k, v in c.items() if k.startswith("SQ_INSTS")})
        if c.get("SQ_ACTIVE_INST_VALU"):  # lanes active per vector instruction (64 = no divergence)
            entry["valu_lanes_active_mean"] = c.get("SQ_THREAD_CYCLES_VALU", 0.0) / c["SQ_ACTIVE_INST_VALU"]
    if "lds" in raw:
        c = raw["lds"]["counters"]
        if c.get("SQ_LDS_IDX_ACTIVE"):
            entry["lds_bank_conflict_frac_of_lds_cycles"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
            entry["lds_array_busy_frac_of_wave_cycles"] = c["SQ_LDS_IDX_ACTIVE"] / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None
    if "fetch" in raw and "write" in raw:
        nl = max(1, raw["fetch"]["dispatches"])
        rd = raw["fetch"]["counters"]["FETCH_SIZE"] * 1024 * 2  # KB; gfx950: FETCH_SIZE reports half the bytes
        wr = raw["write"]["counters"]["WRITE_SIZE"] * 1024
        entry["hbm_read_bytes_per_launch"] = rd / nl
        entry["hbm_write_bytes_per_launch"] = wr / nl
        # a large mesh shard runs as two halves on two streams: a launch then covers walkers / streams walkers
        streams = (raw["fetch"].get("bench") or {}).get("streams") or 1
        entry["hbm_bytes_per_walker_launch"] = (rd + wr) / nl / (args.walkers / streams)
        entry["launches"] = nl
        entry["streams"] = streams
    kc = os.path.join(ROOT, "profiles", "kernel_counters.json")
    allc = json.load(open(kc)) if os.path.exists(kc) else {}
    allc[args.workload] = entry
    json.dump(allc, open(kc, "w"), indent=1)
    print(json.dumps(entry, indent=1))


if __name__ == "__main__":
    main()
