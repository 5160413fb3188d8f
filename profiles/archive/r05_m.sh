#!/bin/bash
# round 5, step m: (1) C3 with the one-launch sort front is 8 us slower although its sort is 7 us faster: is it the 537 MB counter
# table in front of the other buffers?  (2) upper bound of any balancing of k_p2g's cells: the ablation library with every cell capped
# at 8 / 6 particles (results invalid, timing only) on the state after impact
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
line() { grep '^{' | tail -1; }
for rep in 1 2; do
for V in v2 v1 v1tab v2last; do
  case $V in v2) E="X=1";; v1) E="MPMHIP_SORT_V1=1";; v1tab) E="MPMHIP_SORT_V1=2";; v2last) E="MPMHIP_SORT_KEYED_LAST=1";; esac
  env $E python bench.py --config c3 --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | line > $O/r05_m_c3_${V}_$rep.json
done
done
for A in 0 16 32; do
  MPMHIP_LIB_VARIANT=ablate MPMHIP_ABLATE=$A python bench.py --state evolved --no-evolved --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | line > $O/r05_m_cap_$A.json
  MPMHIP_LIB_VARIANT=ablate MPMHIP_ABLATE=$A python bench.py --no-evolved --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | line > $O/r05_m_capl_$A.json
done
python - <<'P'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
r = lambda p: {k: round(v * 1e3, 1) for k, v in p.items()}
for f in sorted(glob.glob(O + "/r05_m_*.json")):
    d = json.load(open(f))
    ev = d.get("evolved") or {}
    print("%-24s %.4f %s | evolved %.4f %s" % (os.path.basename(f), d["ms_per_step"], r(d["phases_ms_per_step"]), ev.get("ms_per_step", 0), r(ev.get("phases_ms_per_step", {}))))
P
