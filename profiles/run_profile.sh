#!/bin/bash
# profiles/run_profile.sh TAG — run on the GPU box (via gpurun): kernel-trace stats + separate PMC passes
# of the default bench.py workload (C3: 256^3 grid, 8M sand particles).  Outputs under gpurun_out/$TAG/.
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- $B > $O/trace.log 2>&1
S="python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- $S > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- $S > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $O/pmc_sq1 -o p -- $S > $O/pmc_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_sq2 -o p -- $S > $O/pmc_sq2.log 2>&1
python $R/profiles/summarize_pmc.py $O/pmc_*/p_counter_collection.csv > $O/pmc_summary.txt 2>&1
python $R/profiles/make_traffic.py $O/pmc_fetch/p_counter_collection.csv $O/pmc_write/p_counter_collection.csv > $O/traffic_c3.json 2>> $O/pmc_summary.txt
grep '^{' $O/trace.log | tail -1 > $O/bench_under_trace.json
ls $O $O/trace
