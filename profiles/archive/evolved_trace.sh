#!/bin/bash
# profiles/evolved_trace.sh TAG [env ...] — on the GPU box: rocprofv3 kernel trace of `bench.py --state evolved` (C3, 400 substeps
# after the block has hit the floor) and the mean duration of every kernel's LAST 20 dispatches = the evolved state's kernels
TAG=${1:-ev}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
W=/tmp/ev_$TAG
mkdir -p $O $W
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --output-format csv -d $W/trace -o t -- python $R/bench.py --state evolved --no-evolved --steps 20 --warmup 5 --no-cpu-baseline > $O/${TAG}_trace.log 2>&1
python $R/profiles/last_calls.py $W/trace/t_kernel_trace.csv 20 > $O/${TAG}_last_calls.txt 2>&1
head -12 $O/${TAG}_last_calls.txt
