#!/bin/bash
# A/B of library variants on the three problem sizes: C3 (8 M), C2 (1 M), 8 virtual ranks of C3 (1 M per ctx)
for v in "$@"; do
for cfg in c3 c2; do MPMHIP_LIB_VARIANT=$v python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d.get('evolved',{}); p=d['phases_ms_per_step']; q=e.get('phases_ms_per_step',{})
print('%-8s %s lattice %.4f grid %.4f sort %.4f p2g %.4f g2p %.4f | evolved %.4f grid %.4f sort %.4f'%('$v' or 'default','$cfg', d['ms_per_step'], p['grid'], p['sort'], p['p2g'], p['g2p'], e.get('ms_per_step',0), q.get('grid',0), q.get('sort',0)))"; done
MPMHIP_LIB_VARIANT=$v python bench.py --virtual 8 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-8s virtual8 per-rank %s phases %s'%('$v' or 'default', [round(x,4) for x in d['per_rank_compute_ms']], {k:round(v,4) for k,v in d['rank0_phases_ms'].items()}))"
done
