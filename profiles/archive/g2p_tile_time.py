"""profiles/g2p_tile_time.py — what share of k_g2p_packed's time a workgroup spends making block tiles resident (the two barriers and
the dependent loads block key -> grid-block slots -> nodes), from wall-clock stamps inside the kernel (variant library
lib/libmpmhip_timing.so = the default sources with -DMPMHIP_TIMING_BUILD).  usage (GPU box): python profiles/g2p_tile_time.py [evolved]"""
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
    evolved = "evolved" in sys.argv[1:]
    cfg = dict(bench.CONFIGS["c3"])
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
    wgs = 4096
    buf = np.zeros((wgs * 4 // 3 + 2, 3), np.uint64)
    assert fn(sim._ctx, 0, buf.ctypes.data_as(C.c_void_p), len(buf)) == 0
    d = buf.reshape(-1)[:wgs * 4].reshape(wgs, 4).astype(np.float64)
    ok = d[:, 3] > 0
    tot, tile, ntile, nchunk = d[ok, 0] / 100.0, d[ok, 1] / 100.0, d[ok, 2], d[ok, 3]  # 100 MHz clock -> us
    print("# k_g2p_packed, c3 %s: %d workgroups with work; per workgroup: %.1f chunks, %.1f tiles loaded" %
          ("after impact" if evolved else "lattice", ok.sum(), nchunk.mean(), ntile.mean()))
    print("time in the kernel per workgroup   mean %.1f us  (p5 %.1f  p95 %.1f)" % (tot.mean(), *np.percentile(tot, [5, 95])))
    print("  of which making tiles resident   mean %.1f us = %.1f %%   (%.2f us per tile)" %
          (tile.mean(), 100 * tile.sum() / tot.sum(), tile.sum() / max(ntile.sum(), 1)))


if __name__ == "__main__":
    main()
