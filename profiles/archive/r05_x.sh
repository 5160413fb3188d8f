#!/bin/bash
# round 5, step x: k_cell_table (the form with the owner list: below 2 M slots and on tiled contexts) with its counters requested in front of
# the neighbour rows' lookups
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sort or crowded or every_form or overflow" > $O/r05_x_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r05_x_pytest.log
tail -2 $O/r05_x_pytest.log
line() { grep '^{' | tail -1; }
for rep in 1 2; do
  python bench.py --config c2 --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | line > $O/r05_x_c2_$rep.json
  MPMHIP_TILE_OVERLAP=0 python bench.py --virtual 8 --steps 24 --warmup 8 2>/dev/null | line > $O/r05_x_v8_$rep.json
  MPMHIP_TILE_OVERLAP=0 python bench.py --virtual 2 --steps 24 --warmup 8 2>/dev/null | line > $O/r05_x_v2_$rep.json
done
python - <<'P'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
r = lambda p: {k: round(v * 1e3, 1) for k, v in p.items()}
for f in sorted(glob.glob(O + "/r05_x_*_[12].json")):
    d = json.load(open(f))
    if "K" in d:
        print("%-26s per rank %.4f ms %s" % (os.path.basename(f), d["per_rank_ms_serial_no_events"], r(d["rank0_phases_ms"])))
    else:
        ev = d.get("evolved") or {}
        print("%-26s %.4f %s | evolved %.4f %s" % (os.path.basename(f), d["ms_per_step"], r(d["phases_ms_per_step"]), ev.get("ms_per_step", 0), r(ev.get("phases_ms_per_step", {}))))
P
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_c2 -o t -- python $R/bench.py --config c2 --steps 20 --warmup 5 --no-cpu-baseline --no-evolved > $O/r05_x_c2_trace.log 2>&1
python $R/profiles/loop_census.py /tmp/tr_c2/t_kernel_trace.csv 12 | tee $O/r05_x_c2_census.txt
