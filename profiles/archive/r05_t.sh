#!/bin/bash
# round 5, step t: 16 blocks per chunk at 8 M — does the cap of the scans' grids (3/8 of the resident workgroups) still fit?
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
line() { grep '^{' | tail -1; }
for rep in 1 2; do
for G in 0 512 1024 1536 2048 4096; do
  MPMHIP_CT_BLOCKS=16 MPMHIP_SCAN_GRID=$G python bench.py --config c3 --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | line > $O/r05_t_c3_g${G}_$rep.json
done
done
python - <<'P'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
r = lambda p: {k: round(v * 1e3, 1) for k, v in p.items()}
for f in sorted(glob.glob(O + "/r05_t_*_[12].json")):
    d = json.load(open(f))
    ev = d.get("evolved") or {}
    print("%-26s %.4f %s | evolved %.4f %s" % (os.path.basename(f), d["ms_per_step"], r(d["phases_ms_per_step"]), ev.get("ms_per_step", 0), r(ev.get("phases_ms_per_step", {}))))
P
