#!/bin/bash
# profiles/r06_s_det.sh — the deterministic mode's cost, per kernel: kernel trace of the C3 bench with MPMHIP_DETERMINISTIC=1, both forms of the
# ordering launch (MPMHIP_CELL_ORDER=1: a wave per block through LDS; 0: a lane per cell), ids from the compact array (Params::pidc)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for form in 1 0; do
  W=/tmp/prof_det_$form
  MPMHIP_DETERMINISTIC=1 MPMHIP_CELL_ORDER=$form rocprofv3 --kernel-trace --stats --output-format csv -d $W -o t -- \
    python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-virtual --no-evolved --no-c5 > $O/r06_s_det_form$form.log 2>&1
  cp $W/t_kernel_stats.csv $O/r06_s_det_form${form}_kernel_stats.csv
  for st in lattice evolved; do
    MPMHIP_DETERMINISTIC=1 MPMHIP_CELL_ORDER=$form python $R/bench.py --state $st --no-cpu-baseline --no-virtual --no-evolved 2>/dev/null | grep '^{' | tail -1 > $O/r06_s_det_form${form}_$st.json
  done
done
python $R/bench.py --no-cpu-baseline --no-virtual 2>/dev/null | grep '^{' | tail -1 > $O/r06_s_bench.json
