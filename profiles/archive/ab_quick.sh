#!/bin/bash
# quick A/B on one box: ab_quick.sh variantA variantB ...   ("" = default library); N alternating rounds of bench.py
# (lattice + evolved phase tables), no PMC passes.  ROUNDS=3 by default.
run() { MPMHIP_LIB_VARIANT=$1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d.get('evolved',{})
p=d['phases_ms_per_step']; q=e.get('phases_ms_per_step',{})
print('%-10s lattice %.4f g2p %.4f p2g %.4f grid %.4f sort %.4f | evolved %.4f g2p %.4f p2g %.4f grid %.4f sort %.4f'%('$1' or 'default', d['ms_per_step'], p['g2p'], p['p2g'], p['grid'], p['sort'], e.get('ms_per_step',0), q.get('g2p',0), q.get('p2g',0), q.get('grid',0), q.get('sort',0)))"; }
for round in $(seq 1 ${ROUNDS:-3}); do for v in "$@"; do run "$v"; done; done
