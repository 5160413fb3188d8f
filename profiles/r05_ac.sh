#!/bin/bash
# round 5, step ac: the GPU suite twice more on a fresh box (flakiness check), smoke, the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
for i in 1 2; do
  timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > $O/r05_ac_pytest_$i.txt; cat $O/r05_ac_pytest_$i.txt | tail -2
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py 2>/dev/null | grep '^{' | tail -1 > $O/r05_ac_bench.json
python - <<'P'
import json, os
d = json.load(open(os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r05_ac_bench.json"))
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["traffic_check"], d["evolved"]["ms_per_step"], d["cpu_baseline"]["value"])
P
