#!/bin/bash
# round 5, step s: blocks per chunk of the cell table again — one rank of 2 / 4 bricks (4 M / 2 M particles: 32 today) with 16, C3 a second
# time on another box, C5 (64 M, four-launch sort: 64 today) with 16
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "overflow or every_form" > $O/r05_s_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r05_s_pytest.log
tail -2 $O/r05_s_pytest.log
line() { grep '^{' | tail -1; }
for rep in 1 2; do
for V in 16 32 64; do
  MPMHIP_CT_BLOCKS=$V python bench.py --config c3 --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | line > $O/r05_s_c3_ct${V}_$rep.json
done
for V in 16 32; do
  for K in 2 4; do
    MPMHIP_CT_BLOCKS=$V MPMHIP_TILE_OVERLAP=0 python bench.py --virtual $K --steps 24 --warmup 8 2>/dev/null | line > $O/r05_s_v${K}_ct${V}_$rep.json
  done
done
done
for V in 16 64; do
  MPMHIP_CT_BLOCKS=$V python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | line > $O/r05_s_c5_ct${V}_1.json
done
python - <<'P'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
r = lambda p: {k: round(v * 1e3, 1) for k, v in p.items()}
for f in sorted(glob.glob(O + "/r05_s_*_[12].json")):
    d = json.load(open(f))
    if "K" in d:
        print("%-26s per rank %.4f ms %s" % (os.path.basename(f), d["per_rank_ms_serial_no_events"], r(d["rank0_phases_ms"])))
    else:
        ev = d.get("evolved") or {}
        print("%-26s %.4f %s | evolved %.4f %s" % (os.path.basename(f), d["ms_per_step"], r(d["phases_ms_per_step"]), ev.get("ms_per_step", 0), r(ev.get("phases_ms_per_step", {}))))
P
