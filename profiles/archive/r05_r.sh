#!/bin/bash
# round 5, step r: the fixed overflow test; blocks per chunk of the cell table at 8 M on the keyed sort (64 = default there, 32, 16)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "overflow or every_form" > $O/r05_r_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r05_r_pytest.log
tail -4 $O/r05_r_pytest.log
line() { grep '^{' | tail -1; }
for rep in 1 2; do
for V in 64 32 16; do
  MPMHIP_CT_BLOCKS=$V python bench.py --config c3 --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | line > $O/r05_r_c3_ct${V}_$rep.json
done
done
python - <<'P'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
r = lambda p: {k: round(v * 1e3, 1) for k, v in p.items()}
for f in sorted(glob.glob(O + "/r05_r_*_[12].json")):
    d = json.load(open(f))
    ev = d.get("evolved") or {}
    print("%-26s %.4f %s | evolved %.4f %s" % (os.path.basename(f), d["ms_per_step"], r(d["phases_ms_per_step"]), ev.get("ms_per_step", 0), r(ev.get("phases_ms_per_step", {}))))
P
