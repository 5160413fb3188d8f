#!/bin/bash
# profiles/run_profile.sh TAG [bench flags] — run on the GPU box (via gpurun): kernel-trace stats + separate PMC passes
# of the default bench.py workload (C3: 256^3 grid, 8M sand particles).  Outputs under gpurun_out/$TAG/.
# e.g.  run_profile.sh r02_c_evolved --state evolved     (the same scene 400 substeps after the block hit the floor;
# the per-kernel means then mix ~2400 evolving substeps with the timed ones: read the bench line under trace for the
# timed region, the PMC passes for per-launch counters of the LAST dispatches — summarize_pmc.py --last N)
TAG=${1:-prof}
shift
X="--no-evolved --no-virtual --no-c5 $*"
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
W=/tmp/prof_$TAG   # raw per-dispatch CSVs stay on the box (an evolving run writes tens of MB per pass); summaries go to $O
mkdir -p $O $W
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline $X"
rocprofv3 --kernel-trace --stats --output-format csv -d $W/trace -o t -- $B > $O/trace.log 2>&1
S="python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline $X"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $W/pmc_fetch -o p -- $S > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $W/pmc_write -o p -- $S > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $W/pmc_sq1 -o p -- $S > $O/pmc_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS --output-format csv -d $W/pmc_sq2 -o p -- $S > $O/pmc_sq2.log 2>&1
python $R/profiles/summarize_pmc.py --last 12 $W/pmc_*/p_counter_collection.csv > $O/pmc_summary.txt 2>&1
python $R/profiles/make_traffic.py --tag $TAG --last 12 --lib $R/taichi_mpm_amd/lib/libmpmhip.so $W/pmc_fetch/p_counter_collection.csv $W/pmc_write/p_counter_collection.csv > $O/traffic.json 2>> $O/pmc_summary.txt
grep '^{' $O/trace.log | tail -1 > $O/bench_under_trace.json
cp $W/trace/t_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
ls $O $W/trace
