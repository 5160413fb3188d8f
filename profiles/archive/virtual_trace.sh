#!/bin/bash
# profiles/virtual_trace.sh TAG K [bench flags] — on the GPU box (via gpurun): rocprofv3 kernel trace of `bench.py --virtual K`
# (K bricks of the workload as K ctx on one GPU, native loop) and the mean duration of every kernel's LAST dispatches =
# what ONE rank's kernels take at the per-rank problem size.  Outputs under gpurun_out/.
TAG=${1:-vt}; K=${2:-8}; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
W=/tmp/vt_$TAG
mkdir -p $O $W
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $W/trace -o t -- python $R/bench.py --virtual $K --steps 12 --warmup 6 $* > $O/${TAG}_trace.log 2>&1
python $R/profiles/last_calls.py $W/trace/t_kernel_trace.csv $((K * 10)) > $O/${TAG}_last_calls.txt 2>&1
cat $O/${TAG}_last_calls.txt | head -24
