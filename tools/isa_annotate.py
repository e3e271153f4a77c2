"""The instructions of one kernel, each with the source line it came from (development aid, no GPU needed).

    hipcc <the library's flags> --cuda-device-only -S -gline-tables-only dsim_hip.hip -o dev_g.s
    python tools/isa_annotate.py dev_g.s out.txt ['_Z17walk_kernel_coop4ILi1ELi1EEv10WalkParams']

Writes `file line | instruction` for every instruction and label of the kernel, in program order.  Reading the hot
loops of the mesh kernel this way found what the per-line COUNTS of tools/isa_by_line.py only hint at (round 5): default
values materialised on every path of a function with early returns, registers zeroed for lanes whose value is never
read, a lane mask turned into 0 / 1 per lane and compared again (`v_cndmask_b32 v, 0, 1, s[..]` + `v_cmp_ne_u32`) where
the mask of the comparison was already there.  `grep -n "v_cndmask_b32_e64 v[0-9]*, 0, 1, " out.txt` lists the last kind."""
import re
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    kernel = sys.argv[3] if len(sys.argv) > 3 else "_Z17walk_kernel_coop4ILi1ELi1EEv10WalkParams"
    lines = open(src).read().split("\n")
    start = [i for i, l in enumerate(lines) if l.startswith(kernel + ":")]
    if not start:
        sys.exit("no such kernel in %s: %s" % (src, kernel))
    start = start[0]
    end = [i for i, l in enumerate(lines) if i > start and l.strip().startswith(".Lfunc_end")][0]
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l)
        if m:
            files[int(m.group(1))] = m.group(2)
    cur, out = ("?", 0), []
    for i in range(start, end):
        s = lines[i].strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        if not s or (s.startswith((".", ";", "//")) and not s.startswith(".LBB")):
            continue
        out.append("%-22s %4d | %s" % (cur[0][-20:], cur[1], s.split(";")[0]))
    open(dst, "w").write("\n".join(out) + "\n")
    print("%d lines -> %s" % (len(out), dst))


if __name__ == "__main__":
    main()
