#!/bin/bash
# round 5, step e: neighbour rows written by k_rank + owner list compacted by k_cell_table + grid pass over the list; one-sided
# refinement of ill-conditioned F in LDS.  GPU tests, the error table, then this library against the previous one (variant head).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/r05_e_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r05_e_pytest.log
tail -4 $O/r05_e_pytest.log
python profiles/illcond_table.py > $O/r05_e_illcond_default.txt 2>&1; cat $O/r05_e_illcond_default.txt | grep -v Warning | head -60
line() { grep '^{' | tail -1; }
for rep in 1 2; do
for V in new head; do
  case $V in new) E="X=1";; head) E="MPMHIP_LIB_VARIANT=head";; esac
  env $E python bench.py --config c2 --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | line > $O/r05_e_c2_${V}_$rep.json
  env $E python bench.py --config c3 --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | line > $O/r05_e_c3_${V}_$rep.json
  for K in 2 4 8; do
    env $E MPMHIP_TILE_OVERLAP=0 python bench.py --virtual $K --steps 24 --warmup 8 2>/dev/null | line > $O/r05_e_v${K}_${V}_$rep.json
  done
done
done
for K in 2 4 8; do bash profiles/virtual_census.sh r05_e_v$K $K > /dev/null; done
cd /tmp && export TMPDIR=/tmp
for C in c2 c3; do
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$C -o t -- python $R/bench.py --config $C --steps 20 --warmup 5 --no-cpu-baseline --no-evolved > $O/r05_e_${C}_trace.log 2>&1
  python $R/profiles/loop_census.py /tmp/tr_$C/t_kernel_trace.csv 12 > $O/r05_e_${C}_census.txt 2>&1
done
cd $R; bash profiles/evolved_trace.sh r05_e_ev > /dev/null
python - <<'P'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
r = lambda p: {k: round(v * 1e3, 1) for k, v in p.items()}
for f in sorted(glob.glob(O + "/r05_e_*_[12].json")):
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
