#!/bin/bash
# round 6: kernel census of the K-brick virtual job (one rank's kernels, per rank-substep) and of the one-ctx C3 / C2 loops
# usage (on the GPU box, from the repo root): bash profiles/r06_census.sh <tag>
tag=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for K in 8 2 4; do
  rm -rf /tmp/prof_v$K
  MPMHIP_TILE_OVERLAP=0 rocprofv3 --kernel-trace -d /tmp/prof_v$K -o v$K --output-format csv -- python $R/bench.py --virtual $K --steps 40 --warmup 10 > /dev/null 2> /tmp/prof_v$K.err
  f=$(find /tmp/prof_v$K -name "*kernel_trace.csv" | head -1)
  python $R/profiles/loop_census.py $f 80 k_g2p > $R/gpurun_out/${tag}_v${K}_census.txt 2>&1
done
