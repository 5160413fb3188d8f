#!/bin/bash
# round 5, step af: the last library build (host-side change only: the counter table's allocation may fail softly): smoke, the parity file, the default line
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q > $O/r05_af_pytest.txt 2>&1; echo "pytest rc=$?" >> $O/r05_af_pytest.txt
grep -E "passed|failed|rc=" $O/r05_af_pytest.txt | tail -3
python bench.py 2>/dev/null | grep '^{' | tail -1 > $O/r05_af_bench.json
python - <<'P'
import json, os
d = json.load(open(os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r05_af_bench.json"))
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["traffic_check"], d["evolved"]["ms_per_step"])
P
