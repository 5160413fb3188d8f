#!/bin/bash
# round 5, step g: list walk by size (k_grid_list / k_grid_blocks, k_cell_table / _plain), reductions + energy of a tiled job; full GPU tests; perf against the previous library
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r05_g_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r05_g_pytest.log
tail -5 $O/r05_g_pytest.log
line() { grep '^{' | tail -1; }
for rep in 1 2; do
for V in new head; do
  case $V in new) E="X=1";; head) E="MPMHIP_LIB_VARIANT=head";;  esac
  env $E python bench.py --config c2 --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | line > $O/r05_g_c2_${V}_$rep.json
  env $E python bench.py --config c3 --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | line > $O/r05_g_c3_${V}_$rep.json
  for K in 2 4 8; do
    env $E MPMHIP_TILE_OVERLAP=0 python bench.py --virtual $K --steps 24 --warmup 8 2>/dev/null | line > $O/r05_g_v${K}_${V}_$rep.json
  done
done
done
python - <<'P'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
r = lambda p: {k: round(v * 1e3, 1) for k, v in p.items()}
for f in sorted(glob.glob(O + "/r05_g_*_[12].json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    if "K" in d:
        print("%-24s per rank %.4f ms %s" % (os.path.basename(f), d["per_rank_ms_serial_no_events"], r(d["rank0_phases_ms"])))
    else:
        ev = d.get("evolved") or {}
        print("%-24s %.4f %s | evolved %.4f %s" % (os.path.basename(f), d["ms_per_step"], r(d["phases_ms_per_step"]), ev.get("ms_per_step", 0), r(ev.get("phases_ms_per_step", {}))))
P
