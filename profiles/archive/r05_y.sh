#!/bin/bash
# round 5, step y: k_p2g with the next block's first records requested before the merge (one wave per block, fewer workgroups than
# blocks): workgroups of the launch (MPMHIP_P2G_WGS; 16384 = default until here: one block per workgroup at C3's 17.5 k blocks, so nothing to prefetch for)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/r05_y_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r05_y_pytest.log
tail -2 $O/r05_y_pytest.log
line() { grep '^{' | tail -1; }
for rep in 1 2; do
for G in 16384 8192 4096 3072 2048; do
  MPMHIP_P2G_WGS=$G python bench.py --config c3 --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | line > $O/r05_y_c3_g${G}_$rep.json
done
for G in 16384 2048 1024; do
  MPMHIP_P2G_WGS=$G python bench.py --config c2 --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | line > $O/r05_y_c2_g${G}_$rep.json
done
done
python - <<'P'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
r = lambda p: {k: round(v * 1e3, 1) for k, v in p.items()}
for f in sorted(glob.glob(O + "/r05_y_*_[12].json")):
    d = json.load(open(f))
    ev = d.get("evolved") or {}
    print("%-26s %.4f %s | evolved %.4f %s" % (os.path.basename(f), d["ms_per_step"], r(d["phases_ms_per_step"]), ev.get("ms_per_step", 0), r(ev.get("phases_ms_per_step", {}))))
P
