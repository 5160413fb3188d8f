"""profiles/p2g_block_times.py — per-block durations of k_p2g from wall-clock stamps inside the kernel (variant library
lib/libmpmhip_timing.so = the default sources with -DMPMHIP_TIMING_BUILD: every workgroup stamps the begin and the end of each
block it rasterises with the 100 MHz wall clock).  Answers what the kernel's time is made of: the duration of a block against
its fullest cell, and how the blocks' start / end times spread over the launch (ramp, tail, imbalance).
usage (GPU box): python profiles/p2g_block_times.py [c3|c2] [evolved]     -> one text table on stdout"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MPMHIP_LIB_VARIANT"] = "timing"
import numpy as np  # noqa: E402

import bench  # noqa: E402
import taichi_mpm_amd as tm  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c3"
    evolved = "evolved" in sys.argv[2:]
    cfg = dict(bench.CONFIGS[name])
    sim = bench.build_sim(tm, cfg, 0)
    sim._ensure_ctx()
    L = sim._L
    sim.run_substeps(10)
    if evolved:
        bench.evolve_to_impact(sim, cfg)
    fn = L._lib.mpmhip_timing_p2g_blocks if hasattr(L, "_lib") else L.mpmhip_timing_p2g_blocks
    fn.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
    assert fn(sim._ctx, 1, None, 0) == 0
    sim.run_substeps(1)
    sim.synchronize()
    prof_blocks = sim.profile()["active_blocks"]
    buf = np.zeros((prof_blocks, 3), np.uint64)
    assert fn(sim._ctx, 0, buf.ctypes.data_as(C.c_void_p), prof_blocks) == 0
    ok = buf[:, 1] > 0
    b, e = buf[ok, 0].astype(np.int64), buf[ok, 1].astype(np.int64)
    cmax, csum = (buf[ok, 2] >> np.uint64(32)).astype(np.int64), (buf[ok, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    t0 = b.min()
    dur = (e - b) / 100.0  # us
    print("# k_p2g block times, %s%s: %d blocks stamped of %d active, %d particles; kernel span (first begin -> last end) %.1f us"
          % (name, " evolved" if evolved else "", ok.sum(), prof_blocks, csum.sum(), (e.max() - t0) / 100.0))
    q = np.percentile(dur, [0, 5, 25, 50, 75, 95, 99, 100])
    print("block duration us   min %.1f  p5 %.1f  p25 %.1f  median %.1f  p75 %.1f  p95 %.1f  p99 %.1f  max %.1f   (mean %.2f)" % (*q, dur.mean()))
    print("sum of block durations %.0f us = %.1f us per wave slot at 2 waves / SIMD x 1024 SIMDs" % (dur.sum(), dur.sum() / 2048.0))
    print("\nfullest cell -> blocks, mean / p95 duration us, mean particles per block")
    for c in sorted(set(cmax.tolist())):
        m = cmax == c
        if m.sum() >= max(8, 0.002 * len(dur)):
            print("  %3d  %7d  %6.2f / %6.2f   %6.1f" % (c, m.sum(), dur[m].mean(), np.percentile(dur[m], 95), csum[m].mean()))
    span = (e.max() - t0) / 100.0
    edges = np.linspace(0, span, 11)
    print("\nkernel timeline in tenths of the span: blocks beginning / ending in each tenth, blocks in flight at its middle")
    for i in range(10):
        lo, hi = edges[i], edges[i + 1]
        mid = t0 + 100.0 * 0.5 * (lo + hi)
        print("  %5.1f - %5.1f us   begin %6d   end %6d   in flight %6d" % (lo, hi, (((b - t0) / 100.0 >= lo) & ((b - t0) / 100.0 < hi)).sum(),
                                                                        (((e - t0) / 100.0 > lo) & ((e - t0) / 100.0 <= hi)).sum(), ((b <= mid) & (e > mid)).sum()))


if __name__ == "__main__":
    main()
