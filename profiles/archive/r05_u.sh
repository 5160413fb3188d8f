#!/bin/bash
# round 5, step u: the rank role of the sort front with the next batch's keys in flight: workgroups of the role (MPMHIP_RANK_WGS; 8192 = one
# batch per workgroup at 8 M = the form until here)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sort or crowded or every_form or overflow" > $O/r05_u_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r05_u_pytest.log
tail -2 $O/r05_u_pytest.log
line() { grep '^{' | tail -1; }
for rep in 1 2; do
for G in 8192 4096 2048 1024; do
  MPMHIP_RANK_WGS=$G python bench.py --config c3 --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | line > $O/r05_u_c3_g${G}_$rep.json
done
for G in 1024 512 256; do
  MPMHIP_RANK_WGS=$G python bench.py --config c2 --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | line > $O/r05_u_c2_g${G}_$rep.json
done
done
python - <<'P'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
r = lambda p: {k: round(v * 1e3, 1) for k, v in p.items()}
for f in sorted(glob.glob(O + "/r05_u_*_[12].json")):
    d = json.load(open(f))
    ev = d.get("evolved") or {}
    print("%-26s %.4f %s | evolved %.4f %s" % (os.path.basename(f), d["ms_per_step"], r(d["phases_ms_per_step"]), ev.get("ms_per_step", 0), r(ev.get("phases_ms_per_step", {}))))
P
