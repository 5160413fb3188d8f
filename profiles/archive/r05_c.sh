#!/bin/bash
# round 5, step c: (1) why the evolved C3 state lost 100 us with the owner-list build: per-kernel times of the evolved state, this
# library against the previous one (variant head), with the statistics copy of round 4, with the scans' round-4 grid;
# (2) the constitutive path on ill-conditioned F: error tables of the three libraries + the new test file
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
bash profiles/evolved_trace.sh r05_c_ev_new > /dev/null
bash profiles/evolved_trace.sh r05_c_ev_head MPMHIP_LIB_VARIANT=head > /dev/null
bash profiles/evolved_trace.sh r05_c_ev_nostats MPMHIP_NO_STATS_STORE=1 > /dev/null
bash profiles/evolved_trace.sh r05_c_ev_scan480 MPMHIP_SCAN_GRID=480 > /dev/null
for t in new head nostats scan480; do echo "== $t"; head -9 $O/r05_c_ev_${t}_last_calls.txt; grep '^{' $O/r05_c_ev_${t}_trace.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phases_ms_per_step'])"; done
cd $R
for V in "" head onesided; do MPMHIP_LIB_VARIANT=$V python profiles/illcond_table.py > $O/r05_c_illcond_${V:-default}.txt 2>&1; done
paste -d'|' $O/r05_c_illcond_head.txt $O/r05_c_illcond_default.txt $O/r05_c_illcond_onesided.txt | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_illcond.py -q -s > $O/r05_c_pytest_illcond.log 2>&1; tail -5 $O/r05_c_pytest_illcond.log
