"""Summarise rocprofv3 --pmc / --kernel-trace CSV output directories (development aid).

    python tools/pmc_summary.py <dir> [<dir> ...] [--match walk_kernel]

Prints, per directory, the per-kernel sum of every counter over all dispatches
whose kernel name contains --match, the number of dispatches and the total
kernel time from the kernel trace."""
import collections
import csv
import glob
import json
import sys


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    match = "walk_kernel"
    for a in sys.argv[1:]:
        if a.startswith("--match="):
            match = a.split("=", 1)[1]
    out = {}
    for d in args:
        entry = {}
        for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
            agg = collections.defaultdict(float)
            disp = set()
            info = {}
            for r in csv.DictReader(open(f)):
                if match in r["Kernel_Name"]:
                    agg[r["Counter_Name"]] += float(r["Counter_Value"])
                    disp.add(r["Dispatch_Id"])
                    info = {k: r[k] for k in ("VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size") if k in r}
            entry["counters"] = dict(agg)
            entry["dispatches"] = len(disp)
            entry["resources"] = info
        for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
            tot = collections.defaultdict(lambda: [0, 0.0])
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0][:60]
                tot[k][0] += 1
                tot[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            entry["kernel_ms"] = {k: {"calls": v[0], "total_ms": v[1]} for k, v in tot.items()}
        out[d] = entry
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
