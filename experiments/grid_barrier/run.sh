#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R/experiments/grid_barrier
{
for n in 1048576 4194304 8388608; do for w in 256 512 768 1024; do timeout 60 ./bar $n $w; done; done
} > $O/grid_barrier.txt 2>&1
cat $O/grid_barrier.txt
