#!/bin/bash
# round 5, step v: k_perm_keyed with four slots of a thread side by side (one grid stride apart) against one at a time
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sort or crowded or every_form or overflow" > $O/r05_v_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r05_v_pytest.log
tail -2 $O/r05_v_pytest.log
line() { grep '^{' | tail -1; }
for rep in 1 2 3; do
for V in batch serial; do
  case $V in batch) E="X=1";; serial) E="MPMHIP_PERM_SERIAL=1";; esac
  env $E python bench.py --config c3 --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | line > $O/r05_v_c3_${V}_$rep.json
  env $E python bench.py --config c2 --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | line > $O/r05_v_c2_${V}_$rep.json
done
done
python - <<'P'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
r = lambda p: {k: round(v * 1e3, 1) for k, v in p.items()}
for f in sorted(glob.glob(O + "/r05_v_*_[123].json")):
    d = json.load(open(f))
    ev = d.get("evolved") or {}
    print("%-26s %.4f %s | evolved %.4f %s" % (os.path.basename(f), d["ms_per_step"], r(d["phases_ms_per_step"]), ev.get("ms_per_step", 0), r(ev.get("phases_ms_per_step", {}))))
P
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_c3 -o t -- python $R/bench.py --config c3 --steps 20 --warmup 5 --no-cpu-baseline --no-evolved > $O/r05_v_c3_trace.log 2>&1
python $R/profiles/loop_census.py /tmp/tr_c3/t_kernel_trace.csv 12 | tee $O/r05_v_c3_census.txt
