#!/bin/bash
# round 5, step ad: launch sizes once more on the final build — k_p2g one workgroup per block at C3 (17 576 blocks against the cap of
# 16 384), k_grid_blocks' workgroups
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
line() { grep '^{' | tail -1; }
for rep in 1 2; do
for V in p16384 p20480 p32768 g2048 g8192 g16384; do
  case $V in p*) E="MPMHIP_P2G_WGS=${V#p}";; g*) E="MPMHIP_GRID_WGS=${V#g}";; esac
  env $E python bench.py --config c3 --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | line > $O/r05_ad_c3_${V}_$rep.json
done
done
python - <<'P'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
r = lambda p: {k: round(v * 1e3, 1) for k, v in p.items()}
for f in sorted(glob.glob(O + "/r05_ad_*_[12].json")):
    d = json.load(open(f))
    ev = d.get("evolved") or {}
    print("%-26s %.4f %s | evolved %.4f %s" % (os.path.basename(f), d["ms_per_step"], r(d["phases_ms_per_step"]), ev.get("ms_per_step", 0), r(ev.get("phases_ms_per_step", {}))))
P
