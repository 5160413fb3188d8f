#!/bin/bash
# round 6, step h: upper bound of "union tiles for 2x2x1 quads of blocks"  before building it: the ablation library
# writes 150 of the 216 nodes of every P2G tile (600 / 4) and lets the grid pass read 5 of the 8 overlapping tiles — results invalid,
# only the timing means anything; the 4-wave workgroup's coupling of four blocks is NOT modelled (it can only cost).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
line() { grep '^{' | tail -1; }
for rep in 1 2 3; do
for A in 0 192 64 128; do
  MPMHIP_LIB_VARIANT=ablate MPMHIP_ABLATE=$A python bench.py --no-evolved --no-virtual --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | line > $O/r06_h_quad_${A}_$rep.json
done
done
for A in 0 192; do
  MPMHIP_LIB_VARIANT=ablate MPMHIP_ABLATE=$A python bench.py --state evolved --no-evolved --no-virtual --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | line > $O/r06_h_quad_ev_${A}.json
done
python - <<'P'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
r = lambda p: {k: round(v * 1e3, 1) for k, v in p.items()}
for f in sorted(glob.glob(O + "/r06_h_quad_*.json")):
    d = json.load(open(f))
    print("%-28s %.4f %s" % (os.path.basename(f), d["ms_per_step"], r(d["phases_ms_per_step"])))
P
