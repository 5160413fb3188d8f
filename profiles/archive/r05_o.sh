#!/bin/bash
# round 5, step o: the hand-out of the ranks by k_g2p with its finish one chunk later (the atomic has the next chunk's arithmetic to return)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref.py -m gpu -q -x > $O/r05_o_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r05_o_pytest.log
tail -3 $O/r05_o_pytest.log
line() { grep '^{' | tail -1; }
for rep in 1 2; do
for V in ranks front; do
  case $V in ranks) E="X=1";; front) E="MPMHIP_G2P_RANKS=0";; esac
  env $E python bench.py --config c3 --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | line > $O/r05_o_c3_${V}_$rep.json
  env $E python bench.py --config c2 --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | line > $O/r05_o_c2_${V}_$rep.json
  env $E MPMHIP_TILE_OVERLAP=0 python bench.py --virtual 8 --steps 24 --warmup 8 2>/dev/null | line > $O/r05_o_v8_${V}_$rep.json
done
done
python - <<'P'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
r = lambda p: {k: round(v * 1e3, 1) for k, v in p.items()}
for f in sorted(glob.glob(O + "/r05_o_*_[12].json")):
    d = json.load(open(f))
    if "K" in d:
        print("%-26s per rank %.4f ms %s" % (os.path.basename(f), d["per_rank_ms_serial_no_events"], r(d["rank0_phases_ms"])))
    else:
        ev = d.get("evolved") or {}
        print("%-26s %.4f %s | evolved %.4f %s %s" % (os.path.basename(f), d["ms_per_step"], r(d["phases_ms_per_step"]), ev.get("ms_per_step", 0), r(ev.get("phases_ms_per_step", {})), (ev.get("roofline") or {}).get("kernel")))
P
