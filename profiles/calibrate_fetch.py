"""Calibration of rocprofv3's FETCH_SIZE for the 64-byte record gathers of k_p2g / k_g2p.

Run on the GPU box:  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d DIR -o p -- python profiles/calibrate_fetch.py
then:                python profiles/calibrate_fetch.py --read DIR/p_counter_collection.csv
The probe kernel (mpmhip_debug_gather_bandwidth) reads exactly (64 + 4) n bytes per launch in three index patterns
(identity, shuffled runs of 8 records, fully shuffled); FETCH_SIZE [KiB] x 1024 / that = the factor to apply.
MI355X_MICROARCH.md says x2 for wide reads on gfx950; this states what the counter does for THIS access width."""
import csv
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N = 1 << 24  # 16 Mi records = 1 GiB


def run():
    import taichi_mpm_amd as tm
    sim = tm.create_simulation3("mpm").initialize(dict(res=(32,) * 3))
    sim.add_particles(dict(type="jelly", cube=(10, 12)))
    sim._ensure_ctx()
    out = {}
    for mode, name in ((0, "identity"), (1, "runs_of_8"), (2, "shuffled")):
        g = C.c_double()
        sim._check(sim._L.mpmhip_debug_gather_bandwidth(sim._ctx, N, mode, 3, C.byref(g)))
        out[name] = g.value
    print(json.dumps({"records": N, "known_bytes_per_launch": 68 * N, "GBps": out}))


def read(path):
    vals = []
    for r in csv.DictReader(open(path)):
        if "k_gather_records_probe" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            vals.append(float(r["Counter_Value"]))
    # 4 launches per pattern (1 warm-up + 3), three patterns in order
    names = ("identity", "runs_of_8", "shuffled")
    out = {}
    for i, nm in enumerate(names):
        v = vals[4 * i + 1:4 * i + 4]
        kib = sum(v) / len(v)
        out[nm] = {"FETCH_SIZE_KiB": kib, "known_bytes": 68 * N, "bytes_per_FETCH_SIZE_KiB": 68 * N / kib,
                   "factor_vs_1024": 68 * N / (kib * 1024)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--read":
        read(sys.argv[2])
    else:
        run()
