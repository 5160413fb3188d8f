"""average duration of the LAST n dispatches of every kernel in a rocprofv3 --kernel-trace CSV (the state a long run
ends in, e.g. bench.py --state evolved, instead of the average over the whole run that --stats gives)
usage: last_calls.py <kernel_trace.csv> [n]"""
import csv, sys, collections
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r["Kernel_Name"].split("(")[0]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows = []
for k, v in d.items():
    v.sort()
    last = v[-n:]
    rows.append((sum(e - s for s, e in last) / len(last) / 1e3, len(v), k))
for us, calls, k in sorted(rows, reverse=True):
    print("%9.1f us  (last %d of %d calls)  %s" % (us, min(n, calls), calls, k[-70:]))
