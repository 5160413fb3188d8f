#!/bin/bash
# kernel times (mean of the last 20 launches) of the 8 M CPIC scene under rocprofv3, per library variant:  cpic_trace.sh "" v1 v2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  rm -rf gpurun_out/cpt; mkdir -p gpurun_out/cpt
  MPMHIP_LIB_VARIANT=$v rocprofv3 --kernel-trace -d gpurun_out/cpt -o t -- python profiles/cpic_scene_8m.py > gpurun_out/cpt/out.txt 2>&1
  echo "== variant '$v': $(grep ms/substep gpurun_out/cpt/out.txt | cut -c1-20)"
  python - <<'PY'
import sqlite3
from collections import defaultdict
c = sqlite3.connect('gpurun_out/cpt/t_results.db').cursor()
d = defaultdict(list)
for n, s, e in c.execute("select name, start, end from kernels order by start"):
    d[n.split('(')[0].replace('void ', '').replace('mpm::', '')].append((e - s) / 1e3)
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1][-20:])):
    if len(v) >= 100 and sum(v[-20:]) / 20 > 20: print("  %8.1f us  %s" % (sum(v[-20:]) / 20, n[:60]))
PY
done
