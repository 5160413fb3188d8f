"""HBM traffic per launch of the transfer kernels from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE).

    python profiles/make_traffic.py <pmc_fetch>/p_counter_collection.csv <pmc_write>/p_counter_collection.csv > traffic_c3.json

Units and gfx950 correction (MI355X_MICROARCH.md, "HBM"): both counters are in KiB; on gfx950 FETCH_SIZE reports
half of the bytes of wide (16 B/lane) reads, so reads = FETCH_SIZE * 1024 * 2; WRITE_SIZE is taken as is
(for k_g2p it matches the bytes the kernel stores: 180 B per particle)."""
import csv
import json
import sys
from collections import defaultdict


LAST = None
TAG = ""
LIB = None  # --lib <libmpmhip.so>: the library the passes ran with; its kernels' code hashes go into the summary (bench.py checks them)


def means(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mpm::", "").split("<")[0]].append(float(r["Counter_Value"]))
    acc = {k: (v[-LAST:] if LAST else v) for k, v in acc.items()}
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main(fetch_csv, write_csv):
    f, w = means(fetch_csv, "FETCH_SIZE"), means(write_csv, "WRITE_SIZE")
    out = {"source": "committed profile %s: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), mean over the last "
                     "dispatches; reads = FETCH_SIZE KiB x 1024 x 2 (the counter reports 128-byte requests as 64: factor 1.98 measured "
                     "for sequential 64-byte record gathers, 1.0 for fully shuffled ones, profiles/r02_c_calibrate_fetch.json — "
                     "an upper bound where the particle order has decayed), writes = WRITE_SIZE KiB x 1024" % (TAG or "?"),
           "kernels": {}}
    for k in sorted(set(f) & set(w)):
        rd, wr = f[k] * 1024 * 2, w[k] * 1024
        out["kernels"][k] = {"read_bytes": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr}
    if LIB:  # identity of the code the counters were taken on: bench.py prints `traffic: null` when the loaded library's kernel differs
        import os
        import subprocess
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import kernel_diff
        hashes = kernel_diff.kernel_hashes(LIB)
        try:
            commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip()
        except Exception:
            commit = ""
        out["code"] = {"commit": commit or None, "kernel_hashes": {k: hashes[k] for k in out["kernels"] if k in hashes}}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    args = sys.argv[1:]
    while args and args[0] in ("--last", "--tag", "--lib"):
        if args[0] == "--last":  # only the last N dispatches of every kernel
            LAST = int(args[1])
        elif args[0] == "--lib":
            LIB = args[1]
        else:
            TAG = args[1]
        args = args[2:]
    main(*args[:2])
