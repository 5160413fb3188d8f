#!/bin/bash
# round 5, step p: the new sort tests + the parity files on the final library, then trace + PMC passes of the default workload in both
# states (profiles/run_profile.sh) so that the committed stats list the kernels of the final sort
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref.py -m gpu -q -x > $O/r05_p_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r05_p_pytest.log
tail -4 $O/r05_p_pytest.log
bash profiles/run_profile.sh r05_p > /dev/null 2>&1
bash profiles/run_profile.sh r05_p_evolved --state evolved > /dev/null 2>&1
cat $O/r05_p/bench_under_trace.json | head -c 600; echo
head -12 $O/r05_p/kernel_stats.csv
cat $O/r05_p/traffic.json | head -c 800; echo
cat $O/r05_p_evolved/traffic.json | head -c 800; echo
