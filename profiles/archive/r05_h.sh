#!/bin/bash
# round 5, step h: the whole GPU suite on the current build (no -x: every failure at once)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -q > $O/r05_h_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r05_h_pytest.log
tail -15 $O/r05_h_pytest.log
