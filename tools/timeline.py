"""Launch timeline of the mesh walk kernel from a rocprofv3 kernel trace (development aid, GPU box):
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --walkers 1250000 --steps 1 --warmup 1 ...
    python tools/timeline.py gpurun_out/trace > profiles/rNN_timeline.txt
Prints, for the LAST pass over the waveform, one line per launch of walk_kernel* (queue, start, end) and, per other
kernel, the number of dispatches inside that window and their summed duration; plus the gaps between consecutive launches
of the same queue."""
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit("no *kernel_trace.csv under %s" % d)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")))
    rows.sort(key=lambda r: r[1])
    walks = [r for r in rows if r[0].startswith("void walk_kernel") or r[0].startswith("walk_kernel")]
    if not walks:
        raise SystemExit("no walk kernel in the trace")
    # a pass over the waveform ends with the signal reduction (one signal_final_kernel per pass): the LAST pass is what lies
    # between the last two of them (round 4 cut passes at gaps longer than three launches - with launches of 9 ms and 3 ms
    # between passes that merged two passes into one)
    finals = [r[1] for r in rows if "signal_final_kernel" in r[0]]
    if len(finals) >= 2:
        last = [w for w in walks if finals[-2] < w[1] < finals[-1]]
    elif finals:
        last = [w for w in walks if w[1] < finals[-1]]
    else:
        last = walks
    if not last:
        last = walks
    t0 = last[0][1]
    t1 = max(w[2] for w in last)
    print("# last pass: %d launches of the walk kernel, %.2f ms from the first start to the last end" % (len(last), (t1 - t0) / 1e6))
    queues = sorted(set(w[3] for w in last))
    prev_end = {}
    busy = []
    for w in last:
        gap = (w[1] - prev_end[w[3]]) / 1e6 if w[3] in prev_end else float("nan")
        print("queue %s  start %9.2f  end %9.2f  duration %8.2f  gap since this queue's previous launch %7.2f" % (
            queues.index(w[3]) + 1, (w[1] - t0) / 1e6, (w[2] - t0) / 1e6, (w[2] - w[1]) / 1e6, gap))
        prev_end[w[3]] = w[2]
        busy.append((w[1], w[2]))
    busy.sort()
    covered, hi = 0, t0
    for a, b in busy:
        if b <= hi:
            continue
        covered += b - max(a, hi)
        hi = b
    print("# at least one walk kernel running: %.2f ms of %.2f ms (%.1f %%)" % (covered / 1e6, (t1 - t0) / 1e6, 100.0 * covered / (t1 - t0)))
    others = {}
    for name, a, b, q in rows:
        if a < t0 or a > t1 or name.startswith("void walk_kernel") or name.startswith("walk_kernel"):
            continue
        k = others.setdefault(name[:70], [0, 0])
        k[0] += 1
        k[1] += b - a
    print("# the other kernels inside the same window: dispatches, summed duration")
    for name, (c, t) in sorted(others.items(), key=lambda kv: -kv[1][1]):
        print("%6d %9.2f ms  %s" % (c, t / 1e6, name))
    n_launch = len(last)
    print("# dispatches between two walk launches (all kernels but the walk kernel / launches): %.1f" % (sum(c for c, _ in others.values()) / max(1, n_launch)))


if __name__ == "__main__":
    main()
