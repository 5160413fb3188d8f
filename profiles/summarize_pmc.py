"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel, mean of each counter over dispatches."""
import csv
import sys
from collections import defaultdict


def main(paths):
    last = None
    if paths and paths[0] == "--last":  # only the last N dispatches of every kernel (the end of an evolving run)
        last = int(paths[1])
        paths = paths[2:]
    acc = defaultdict(lambda: defaultdict(list))
    for path in paths:
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = sorted({c for k in acc.values() for c in k})
    for name, d in sorted(acc.items()):
        if not name.startswith("mpm::"):
            continue
        print(name)
        for c in counters:
            if c in d:
                v = d[c][-last:] if last else d[c]
                print("   %-22s %14.4g   (n=%d)" % (c, sum(v) / len(v), len(v)))


if __name__ == "__main__":
    main(sys.argv[1:])
