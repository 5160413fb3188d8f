"""profiles/g2p_split_store.py — timing experiment for VERDICT r3 item 4(b): does k_g2p get faster when it writes a 16-byte
position + a 48-byte G2P record + a 48-byte P2G record (116 B per particle) instead of two 64-byte records (132 B), with the
48-byte records packed in the LDS transpose so that the store path keeps its registers?  The variant library
lib/libmpmhip_splitstore.so (-DMPMHIP_EXP_SPLIT_STORE) carries ONLY the changed store path of k_g2p — nothing reads the split
records — so one launch from a saved state is timed at a time: load snapshot -> one substep with k_g2p bracketed by events.
usage (GPU box):  python profiles/g2p_split_store.py make     (default library: writes /tmp/g2p_*.snap, prints its own times)
                  MPMHIP_LIB_VARIANT=splitstore python profiles/g2p_split_store.py time"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
import taichi_mpm_amd as tm  # noqa: E402

REPS = 8


def fresh(cfg):
    sim = bench.build_sim(tm, cfg, 0)
    sim._ensure_ctx()
    return sim


def time_state(cfg, path):
    sim = fresh(cfg)
    out = []
    for _ in range(REPS):
        sim.load_snapshot(path)
        sim.set_profiling(2)
        sim.profile(reset=True)
        sim.substep()
        out.append(sim.profile(reset=True)["phases"]["g2p"])
        sim.set_profiling(0)
    sim.close()
    return np.array(out)


def main():
    cfg = dict(bench.CONFIGS["c3"])
    tag = os.environ.get("MPMHIP_LIB_VARIANT", "") or "default"
    if sys.argv[1] == "make":
        sim = fresh(cfg)
        sim.run_substeps(10)
        sim.save_snapshot("/tmp/g2p_lattice.snap")
        bench.evolve_to_impact(sim, cfg)
        sim.save_snapshot("/tmp/g2p_evolved.snap")
        sim.close()
    for state in ("lattice", "evolved"):
        t = time_state(cfg, "/tmp/g2p_%s.snap" % state)
        print("%-11s %-8s k_g2p ms: median %.4f  min %.4f  max %.4f  (first substep after a snapshot load, %d loads)" % (
            tag, state, np.median(t), t.min(), t.max(), REPS))


if __name__ == "__main__":
    main()
