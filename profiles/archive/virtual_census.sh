#!/bin/bash
# profiles/virtual_census.sh TAG K [env...] — on the GPU box: rocprofv3 kernel trace of `bench.py --virtual K` (no overlap split), then
# loop_census.py over the last 10 substeps of all K ranks and last_calls.py.  Outputs under gpurun_out/.
TAG=${1:-vc}; K=${2:-8}; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
W=/tmp/vc_$TAG
mkdir -p $O $W
cd /tmp && export TMPDIR=/tmp
env MPMHIP_TILE_OVERLAP=0 "$@" rocprofv3 --kernel-trace --output-format csv -d $W/trace -o t -- python $R/bench.py --virtual $K --steps 12 --warmup 6 > $O/${TAG}_trace.log 2>&1
python $R/profiles/loop_census.py $W/trace/t_kernel_trace.csv $((K * 10)) > $O/${TAG}_census.txt 2>&1
python $R/profiles/last_calls.py $W/trace/t_kernel_trace.csv $((K * 10)) > $O/${TAG}_last_calls.txt 2>&1
grep '^{' $O/${TAG}_trace.log | tail -1 > $O/${TAG}_under_trace.json
cat $O/${TAG}_census.txt
