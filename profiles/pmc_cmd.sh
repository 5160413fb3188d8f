cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmc1 -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc2 -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc3 -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc4 -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc4.log 2>&1
ls $R/gpurun_out/pmc1 $R/gpurun_out/pmc3; tail -3 $R/gpurun_out/pmc1.log
