#!/bin/bash
bash profiles/run_profile.sh r02_c > /dev/null 2>&1
bash profiles/run_profile.sh r02_c_evolved --state evolved > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/cal -o p -- python $R/profiles/calibrate_fetch.py > $R/gpurun_out/r02_c/calibrate_run.json 2>/dev/null
python $R/profiles/calibrate_fetch.py --read /tmp/cal/p_counter_collection.csv > $R/gpurun_out/r02_c/calibrate_fetch.json 2>&1
cat $R/gpurun_out/r02_c/calibrate_run.json $R/gpurun_out/r02_c/calibrate_fetch.json
grep -A30 "k_g2p" $R/gpurun_out/r02_c/pmc_summary.txt | head -34
grep -A30 "k_g2p" $R/gpurun_out/r02_c_evolved/pmc_summary.txt | head -34
cat $R/gpurun_out/r02_c_evolved/bench_under_trace.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phases_ms_per_step'])"
