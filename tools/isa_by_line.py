"""Static instruction counts of one kernel by source line (development aid, no GPU needed).

    hipcc <the library's flags> --cuda-device-only -S -gline-tables-only dsim_hip.hip -o dev_g.s
    python tools/isa_by_line.py dev_g.s '_Z17walk_kernel_coop4ILi1ELi1EEv10WalkParams' [file:lo-hi ...]

Every instruction is attributed to the last `.loc` before it.  Prints VALU / SALU / LDS / VMEM / SMEM counts per
source file and line (only lines with VALU >= 3 unless ranges are given), so that the static weight of a stage of
the traversal can be compared between builds; dynamic weights come from the traversal statistics."""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    path, func = sys.argv[1], sys.argv[2]
    ranges = []
    for a in sys.argv[3:]:
        f, r = a.split(":")
        lo, hi = r.split("-")
        ranges.append((f, int(lo), int(hi)))
    files = {}
    counts = collections.defaultdict(lambda: collections.Counter())
    inside = False
    cur = ("?", 0)
    total = collections.Counter()
    for line in open(path):
        s = line.strip()
        m = re.match(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', s)
        if m:
            files[int(m.group(1))] = m.group(2)
            continue
        if s.startswith(func + ":"):
            inside = True
            continue
        if inside and s.startswith(".Lfunc_end"):
            break
        if not inside:
            continue
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        if not s or s.startswith((".", ";", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        k = classify(op)
        counts[cur][k] += 1
        total[k] += 1
    print("total", dict(total))
    byfile = collections.defaultdict(collections.Counter)
    for (f, l), c in counts.items():
        byfile[f].update(c)
    for f, c in byfile.items():
        print("  %-28s %s" % (f, dict(c)))
    keys = sorted(counts)
    for (f, l) in keys:
        c = counts[(f, l)]
        if ranges:
            if not any(f == rf and lo <= l <= hi for rf, lo, hi in ranges):
                continue
        elif c["valu"] < 3:
            continue
        print("%-22s %5d  valu %4d salu %4d lds %3d vmem %3d smem %3d" % (f, l, c["valu"], c["salu"], c["lds"], c["vmem"], c["smem"]))


if __name__ == "__main__":
    main()
