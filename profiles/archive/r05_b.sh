#!/bin/bash
# round 5, step b: the grid pass over k_cell_table's owner list (MPMHIP_GRID_WALK=2, default) against the pre-round-5 walks in the
# same library (=0) and against the previous library (lib variant `head`), on C2 / C3 / 2, 4, 8 virtual ranks; GPU tests first.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/r05_b_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r05_b_pytest.log
tail -4 $O/r05_b_pytest.log
line() { grep '^{' | tail -1; }
for rep in 1 2; do
for V in new old head; do
  case $V in new) E="MPMHIP_GRID_WALK=2";; old) E="MPMHIP_GRID_WALK=0";; head) E="MPMHIP_LIB_VARIANT=head";; esac
  env $E python bench.py --config c2 --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | line > $O/r05_b_c2_${V}_$rep.json
  env $E python bench.py --config c3 --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | line > $O/r05_b_c3_${V}_$rep.json
  for K in 2 4 8; do
    env $E MPMHIP_TILE_OVERLAP=0 python bench.py --virtual $K --steps 24 --warmup 8 2>/dev/null | line > $O/r05_b_v${K}_${V}_$rep.json
  done
done
done
for K in 2 4 8; do bash profiles/virtual_census.sh r05_b_v$K $K > /dev/null; done
cd /tmp && export TMPDIR=/tmp
for C in c2 c3; do
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$C -o t -- python $R/bench.py --config $C --steps 20 --warmup 5 --no-cpu-baseline --no-evolved > $O/r05_b_${C}_trace.log 2>&1
  python $R/profiles/loop_census.py /tmp/tr_$C/t_kernel_trace.csv 12 > $O/r05_b_${C}_census.txt 2>&1
done
python - <<'P'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
for f in sorted(glob.glob(O + "/r05_b_*_[12].json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    if "K" in d:
        print(os.path.basename(f), "per rank %.4f ms" % d["per_rank_ms_serial_no_events"], {k: round(v * 1e3, 1) for k, v in d["rank0_phases_ms"].items()})
    else:
        ev = d.get("evolved") or {}
        print(os.path.basename(f), "%.4f ms" % d["ms_per_step"], {k: round(v * 1e3, 1) for k, v in (d.get("phases_ms") or d.get("phase_ms") or {}).items()}, "evolved %.4f" % ev.get("ms_per_step", 0))
P
