"""census of ONE window of a rocprofv3 --kernel-trace CSV: every kernel dispatched between the start of the n-th last and the end
of the last dispatch of an anchor kernel (default: the G2P kernel = one per rank and substep), with its count per anchor
dispatch, mean duration and share of the window.  Names every runtime blit (__amd_rocclr_fillBufferAligned / copyBuffer) that a
stepping loop issues, and what the kernels of ONE rank take at the per-rank problem size.
usage: loop_census.py <kernel_trace.csv> [n_anchor_dispatches] [anchor substring] [skip: that many anchor dispatches at the END are left out]"""
import csv, sys, collections
path = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
anchor = sys.argv[3] if len(sys.argv) > 3 else "k_g2p"
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
anc = [i for i, r in enumerate(rows) if anchor in r[2]]
if not anc:
    sys.exit("no dispatch of %r" % anchor)
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0
if skip:
    anc = anc[:-skip]
n = min(n, len(anc) - 1)
# window: from the end of the anchor dispatch BEFORE the first counted one to the end of the last one = n whole substeps
i0, i1 = anc[-n - 1] + 1, anc[-1]
t0, t1 = rows[i0 - 1][1], rows[i1][1]
d = collections.defaultdict(list)
for s, e, k in rows[i0:i1 + 1]:
    d[k].append(e - s)
busy = sum(sum(v) for v in d.values())
print("# window: %d dispatches of '%s' (= rank-substeps), %.1f us wall, %.1f us inside kernels, %.3f us per rank-substep wall / %.3f in kernels"
      % (n, anchor, (t1 - t0) / 1e3, busy / 1e3, (t1 - t0) / 1e3 / n, busy / 1e3 / n))
print("# per rank-substep: calls x mean us = us   kernel")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print("%6.2f x %8.2f us = %8.2f us  %5.1f %%  %s" % (len(v) / n, sum(v) / len(v) / 1e3, sum(v) / 1e3 / n, 100.0 * sum(v) / busy, k[-72:]))
blits = {k: len(v) for k, v in d.items() if "rocclr" in k}
print("# runtime blits inside the window: %s" % (blits if blits else "none"))
