#!/bin/bash
# round 6: kernel census of the 8-brick virtual job WITH the boundary / interior split (MPMHIP_TILE_OVERLAP=1), per rank-substep
# (anchor: TWO k_g2p launches per rank-substep, boundary + interior: the printed per-'rank-substep' numbers are per HALF; the last 160
# dispatches — bench.py's per-phase pass, which runs unsplit — are left out)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_vs8
MPMHIP_TILE_OVERLAP=1 rocprofv3 --kernel-trace -d /tmp/prof_vs8 -o v8 --output-format csv -- python $R/bench.py --virtual 8 --steps 40 --warmup 10 > /dev/null 2> /tmp/prof_vs8.err
f=$(find /tmp/prof_vs8 -name "*kernel_trace.csv" | head -1)
python $R/profiles/loop_census.py $f 320 k_g2p 200 > $R/gpurun_out/r06_ab_v8_split_census.txt 2>&1
