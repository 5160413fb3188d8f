#!/bin/bash
# profiles/r06_w_co_ab.sh — A/B on ONE box of k_cell_order_blocks builds (LDS padding on / off, waves per SIMD asked of the compiler):
# the deterministic mode's extra ms per C3 substep, three rounds
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_w_co_ab.txt
: > $O
for round in 1 2 3; do
  for v in "" co_p0w6 co_p1w4 co_p1w5; do
    MPMHIP_LIB_VARIANT=$v python $R/bench.py --no-cpu-baseline --no-virtual --no-evolved 2>/dev/null | grep '^{' | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('round $round variant [%s]' % '$v', 'ms_per_step %.4f' % d['ms_per_step'], 'deterministic extra %.4f' % d['deterministic']['extra_ms_per_step'])" >> $O
  done
done
cat $O
