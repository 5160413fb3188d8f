#!/bin/bash
# A/B on one box: ab.sh variantA variantB ...   ("" = default library); two alternating rounds + SQ_INSTS_VALU of k_g2p
run() { MPMHIP_LIB_VARIANT=$1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['evolved']
print('%-10s lattice %.4f g2p %.4f p2g %.4f sort %.4f | evolved %.4f g2p %.4f p2g %.4f sort %.4f'%('$1' or 'default', d['ms_per_step'], d['phases_ms_per_step']['g2p'], d['phases_ms_per_step']['p2g'], d['phases_ms_per_step']['sort'], e['ms_per_step'], e['phases_ms_per_step']['g2p'], e['phases_ms_per_step']['p2g'], e['phases_ms_per_step']['sort']))"; }
for round in 1 2; do for v in "$@"; do run "$v"; done; done
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  for st in lattice evolved; do
    rm -rf /tmp/abpmc; MPMHIP_LIB_VARIANT=$v rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d /tmp/abpmc -o p -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-evolved --state $st > /dev/null 2>&1
    python $R/profiles/summarize_pmc.py --last 8 /tmp/abpmc/p_counter_collection.csv 2>/dev/null | grep -A3 -E "k_g2p|k_p2g" | grep -E "k_g2p|k_p2g|SQ_INSTS_VALU" | tr '\n' ' '; echo " [$v $st]"
  done
done
