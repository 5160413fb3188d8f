#!/bin/bash
# round 5, step l: the front of the sort as one launch (k_sort_front: block table + ranks on key-indexed counters; MPMHIP_SORT_V1=1 =
# the four-launch sort in the same library): GPU suite, then A/B on C2 / C3 / 2, 4, 8 virtual ranks, census of the new loop
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -q -x > $O/r05_l_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r05_l_pytest.log
tail -4 $O/r05_l_pytest.log
line() { grep '^{' | tail -1; }
for rep in 1 2; do
for V in v2 v1; do
  case $V in v2) E="X=1";; v1) E="MPMHIP_SORT_V1=1";; esac
  env $E python bench.py --config c2 --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | line > $O/r05_l_c2_${V}_$rep.json
  env $E python bench.py --config c3 --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | line > $O/r05_l_c3_${V}_$rep.json
  for K in 2 4 8; do
    env $E MPMHIP_TILE_OVERLAP=0 python bench.py --virtual $K --steps 24 --warmup 8 2>/dev/null | line > $O/r05_l_v${K}_${V}_$rep.json
  done
done
done
bash profiles/virtual_census.sh r05_l_v8 8 > /dev/null
cd /tmp && export TMPDIR=/tmp
for C in c2 c3; do
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$C -o t -- python $R/bench.py --config $C --steps 20 --warmup 5 --no-cpu-baseline --no-evolved > $O/r05_l_${C}_trace.log 2>&1
  python $R/profiles/loop_census.py /tmp/tr_$C/t_kernel_trace.csv 12 > $O/r05_l_${C}_census.txt 2>&1
done
cd $R; bash profiles/evolved_trace.sh r05_l_ev > /dev/null
python - <<'P'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
r = lambda p: {k: round(v * 1e3, 1) for k, v in p.items()}
for f in sorted(glob.glob(O + "/r05_l_*_[12].json")):
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
cat $O/r05_l_c2_census.txt $O/r05_l_c3_census.txt $O/r05_l_v8_census.txt | grep -v "^#"; grep -v "stream_copy\|k_affine\|build_keys\|gather_records" $O/r05_l_ev_last_calls.txt | head -9
