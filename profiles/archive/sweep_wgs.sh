#!/bin/bash
# workgroup-count sweep of k_g2p / k_p2g on one box: env knobs MPMHIP_G2P_WGS / MPMHIP_P2G_WGS, bench.py phase tables
run() { env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d.get('evolved',{})
p=d['phases_ms_per_step']; q=e.get('phases_ms_per_step',{})
print('%-28s lattice %.4f g2p %.4f p2g %.4f | evolved %.4f g2p %.4f p2g %.4f'%('$1', d['ms_per_step'], p['g2p'], p['p2g'], e.get('ms_per_step',0), q.get('g2p',0), q.get('p2g',0)))"; }
for v in MPMHIP_G2P_WGS=4096 MPMHIP_G2P_WGS=2304 MPMHIP_G2P_WGS=3072 MPMHIP_G2P_WGS=6144 MPMHIP_G2P_WGS=8192 MPMHIP_G2P_WGS=16384 MPMHIP_G2P_WGS=4096 MPMHIP_P2G_WGS=8192 MPMHIP_P2G_WGS=32768; do run $v; done
