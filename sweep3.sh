#!/bin/bash
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['evolved']
print('$1 lattice %.4f g2p %.4f p2g %.4f | evolved %.4f g2p %.4f p2g %.4f sort %.4f'%(d['ms_per_step'], d['phases_ms_per_step']['g2p'], d['phases_ms_per_step']['p2g'], e['ms_per_step'], e['phases_ms_per_step']['g2p'], e['phases_ms_per_step']['p2g'], e['phases_ms_per_step']['sort']))"; }
run base
MPMHIP_G2P_MINW=14 run minw4
MPMHIP_G2P_MINW=12 run minw2
MPMHIP_G2P_MINW=23 run nt128
MPMHIP_G2P_WGS=8192 run wgs8192
MPMHIP_G2P_WGS=2048 run wgs2048
