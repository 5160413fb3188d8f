#!/bin/bash
# A/B of run-time knobs on one box: env_ab.sh <config> "VAR=value" ...   (A=0 = defaults)
cfg=$1; shift
run() { env $1 timeout 120 python bench.py --config $cfg --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['evolved']; p=d['phases_ms_per_step']; q=e['phases_ms_per_step']
f=lambda t,m: '%.4f = '%t+' '.join('%s %.4f'%(k,m[k]) for k in ('sort','p2g','grid','g2p'))
print('%-22s %s lattice %s | evolved %s'%('$1', '$cfg', f(d['ms_per_step'],p), f(e['ms_per_step'],q)))"; }
for round in 1 2; do for v in "$@"; do run "$v"; done; done
