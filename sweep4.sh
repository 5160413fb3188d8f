#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref.py -x -q 2>&1 | tail -3
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['evolved']
print('$1 lattice %.4f g2p %.4f p2g %.4f | evolved %.4f g2p %.4f p2g %.4f sort %.4f'%(d['ms_per_step'], d['phases_ms_per_step']['g2p'], d['phases_ms_per_step']['p2g'], e['ms_per_step'], e['phases_ms_per_step']['g2p'], e['phases_ms_per_step']['p2g'], e['phases_ms_per_step']['sort']))"; }
run minw2
MPMHIP_G2P_MINW=13 run minw3
