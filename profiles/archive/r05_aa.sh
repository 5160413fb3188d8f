#!/bin/bash
# round 5, step aa: k_p2g_mixed (full rounds one block per wave, the tail's blocks split over two waves) against k_p2g
# (MPMHIP_P2G_MIXED=0) at 1 M particles and on a rank of 8 / 4 bricks
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref.py -m gpu -q -x > $O/r05_aa_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r05_aa_pytest.log
tail -2 $O/r05_aa_pytest.log
line() { grep '^{' | tail -1; }
for rep in 1 2; do
for V in mixed plain; do
  case $V in mixed) E="X=1";; plain) E="MPMHIP_P2G_MIXED=0";; esac
  env $E python bench.py --config c2 --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | line > $O/r05_aa_c2_${V}_$rep.json
  env $E MPMHIP_TILE_OVERLAP=0 python bench.py --virtual 8 --steps 24 --warmup 8 2>/dev/null | line > $O/r05_aa_v8_${V}_$rep.json
  env $E MPMHIP_TILE_OVERLAP=0 python bench.py --virtual 4 --steps 24 --warmup 8 2>/dev/null | line > $O/r05_aa_v4_${V}_$rep.json
done
done
python - <<'P'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
r = lambda p: {k: round(v * 1e3, 1) for k, v in p.items()}
for f in sorted(glob.glob(O + "/r05_aa_*_[12].json")):
    d = json.load(open(f))
    if "K" in d:
        print("%-26s per rank %.4f ms %s" % (os.path.basename(f), d["per_rank_ms_serial_no_events"], r(d["rank0_phases_ms"])))
    else:
        ev = d.get("evolved") or {}
        print("%-26s %.4f %s | evolved %.4f %s" % (os.path.basename(f), d["ms_per_step"], r(d["phases_ms_per_step"]), ev.get("ms_per_step", 0), r(ev.get("phases_ms_per_step", {}))))
P
