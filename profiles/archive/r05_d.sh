#!/bin/bash
# round 5, step d: the evolved leg of the default bench flow lost 75 us in G2P with the owner-list build although the kernel itself
# (rocprof, --state evolved) did not: which ingredient?
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
run() { tag=$1; shift; env "$@" python bench.py --config c3 --steps 30 --warmup 8 --no-cpu-baseline $EXTRA 2>/dev/null | grep '^{' | tail -1 > $O/r05_d_$tag.json; }
run plain X=1
run nostats MPMHIP_NO_STATS_STORE=1
run scan480 MPMHIP_SCAN_GRID=480
run both MPMHIP_NO_STATS_STORE=1 MPMHIP_SCAN_GRID=480
run head MPMHIP_LIB_VARIANT=head
EXTRA="--state evolved --no-evolved" run state_evolved X=1
EXTRA="--state evolved --no-evolved" run state_evolved_head MPMHIP_LIB_VARIANT=head
python - <<'P'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
for f in sorted(glob.glob(O + "/r05_d_*.json")):
    d = json.load(open(f))
    e = d.get("evolved") or {}
    r = lambda p: {k: round(v * 1e3, 1) for k, v in p.items()}
    print("%-28s %.4f %s | evolved %.4f %s" % (os.path.basename(f), d["ms_per_step"], r(d["phases_ms_per_step"]), e.get("ms_per_step", 0), r(e.get("phases_ms_per_step", {}))))
P
