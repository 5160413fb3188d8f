#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref.py -x -q 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('lattice ms/step %.4f'%d['ms_per_step'], {k: round(v,4) for k,v in d['phases_ms_per_step'].items()}, 'frac %.3f'%d['roofline']['frac'])
e=d['evolved']
print('evolved ms/step %.4f'%e['ms_per_step'], {k: round(v,4) for k,v in e['phases_ms_per_step'].items()}, 'frac %.3f'%e['roofline']['frac'])"
