"""Aggregate rocprofv3 PC-sampling output (csv) of one run into a histogram by instruction.

    python3 tools/pc_hist.py <rocprofv3 output dir> <output prefix>

Writes <prefix>_head.txt (header + first rows of every pc-sampling csv, so that the format is on record),
<prefix>_hist.csv (count per distinct (instruction, comment) pair, every column that looks like a stall / issue
reason summed when the stochastic method provides them) and prints the top of it."""
import collections
import csv
import glob
import sys


def main():
    d, prefix = sys.argv[1], sys.argv[2]
    files = [f for f in glob.glob(d + "/**/*.csv", recursive=True) if "pc_sampling" in f]
    head = open(prefix + "_head.txt", "w")
    hist = collections.Counter()
    extra = collections.defaultdict(collections.Counter)
    total = 0
    for f in files:
        with open(f, newline="") as fh:
            rd = csv.DictReader(fh)
            head.write("%s\n%s\n" % (f, rd.fieldnames))
            for i, r in enumerate(rd):
                if i < 5:
                    head.write(repr(r) + "\n")
                key = (r.get("Instruction", "?"), r.get("Instruction_Comment", ""))
                hist[key] += 1
                total += 1
                for k, v in r.items():
                    if k in ("Instruction", "Instruction_Comment", "Sample_Timestamp", "Exec_Mask", "Dispatch_Id",
                             "Correlation_Id", "Wave_Issued_Instruction"):
                        if k == "Wave_Issued_Instruction":
                            extra[key]["issued:" + str(v)] += 1
                        continue
                    if v not in ("", "0", None) and len(str(v)) < 40:
                        extra[key][k + ":" + str(v)] += 1
    head.close()
    with open(prefix + "_hist.csv", "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["count", "instruction", "comment", "extra"])
        for key, c in hist.most_common():
            ex = ";".join("%s=%d" % kv for kv in extra[key].most_common(6))
            w.writerow([c, key[0], key[1], ex])
    print("pc samples:", total, "distinct:", len(hist), "files:", files)
    for key, c in hist.most_common(25):
        print("%7d %5.2f%%  %-60s %s" % (c, 100.0 * c / max(total, 1), key[0][:60], key[1][-60:]))


if __name__ == "__main__":
    main()
