#!/bin/bash
# round 5, step ac: the GPU suite on a fresh box, twice, without -x (flakiness check) with summary lines and exit codes kept
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
for i in 1 2; do
  timeout 1500 python -m pytest tests -m gpu -q > $O/r05_ac_pytest_full_$i.txt 2>&1; echo "pytest rc=$?" >> $O/r05_ac_pytest_full_$i.txt
  grep -E "passed|failed|error|rc=|^FAILED" $O/r05_ac_pytest_full_$i.txt | tail -6
done
