#!/bin/bash
# round 5, step ac: the GPU suite once more on a fresh box (flakiness check) with its summary line and exit code kept
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/r05_ac_pytest_full.txt 2>&1; echo "pytest rc=$?" >> $O/r05_ac_pytest_full.txt
grep -E "passed|failed|error|rc=" $O/r05_ac_pytest_full.txt | tail -5
