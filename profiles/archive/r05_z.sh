#!/bin/bash
# round 5, step z: k_p2g with two waves per block splitting every cell's particles (tuning-variant library) at 1 M particles and on a rank
# of 8 bricks — is half a block half the chain?  (decides whether splitting only the TAIL blocks of a small launch is worth building)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
line() { grep '^{' | tail -1; }
for rep in 1 2; do
for V in 0 12; do
  MPMHIP_LIB_VARIANT=tuning MPMHIP_P2G_SPLIT=$V python bench.py --config c2 --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | line > $O/r05_z_c2_s${V}_$rep.json
  MPMHIP_LIB_VARIANT=tuning MPMHIP_P2G_SPLIT=$V MPMHIP_TILE_OVERLAP=0 python bench.py --virtual 8 --steps 24 --warmup 8 2>/dev/null | line > $O/r05_z_v8_s${V}_$rep.json
done
done
python - <<'P'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
r = lambda p: {k: round(v * 1e3, 1) for k, v in p.items()}
for f in sorted(glob.glob(O + "/r05_z_*_[12].json")):
    d = json.load(open(f))
    if "K" in d:
        print("%-26s per rank %.4f ms %s" % (os.path.basename(f), d["per_rank_ms_serial_no_events"], r(d["rank0_phases_ms"])))
    else:
        ev = d.get("evolved") or {}
        print("%-26s %.4f %s | evolved %.4f %s" % (os.path.basename(f), d["ms_per_step"], r(d["phases_ms_per_step"]), ev.get("ms_per_step", 0), r(ev.get("phases_ms_per_step", {}))))
P
