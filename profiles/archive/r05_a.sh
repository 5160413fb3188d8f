#!/bin/bash
# round-5 start: where one rank's time goes at 2 / 4 / 8 bricks (per kernel), and the one-ctx C2 / C3 loops, before any change
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
for K in 2 4 8; do
  MPMHIP_TILE_OVERLAP=0 python $R/bench.py --virtual $K --steps 24 --warmup 8 2>/dev/null | grep '^{' | tail -1 > $O/r05_a_virtual$K.json
  bash $R/profiles/virtual_census.sh r05_a_v$K $K > /dev/null
done
cd /tmp && export TMPDIR=/tmp
for C in c2 c3; do
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$C -o t -- python $R/bench.py --config $C --steps 20 --warmup 5 --no-cpu-baseline --no-evolved > $O/r05_a_${C}_trace.log 2>&1
  python $R/profiles/loop_census.py /tmp/tr_$C/t_kernel_trace.csv 12 > $O/r05_a_${C}_census.txt 2>&1
done
python $R/bench.py --config c2 --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $O/r05_a_bench_c2.json
head -30 $O/r05_a_v2_census.txt $O/r05_a_v8_census.txt $O/r05_a_c2_census.txt
