#!/bin/bash
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['evolved']
print('lattice %.4f'%d['ms_per_step'], {k: round(v,4) for k,v in d['phases_ms_per_step'].items()}, 'frac %.3f'%d['roofline']['frac'])
print('evolved %.4f'%e['ms_per_step'], {k: round(v,4) for k,v in e['phases_ms_per_step'].items()}, 'frac %.3f'%e['roofline']['frac'], 'both %.3f'%e['p2g_plus_g2p_hbm_frac_algorithmic'])"
