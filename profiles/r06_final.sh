#!/bin/bash
# profiles/r06_final.sh — everything the round-6 numbers rest on, on ONE box, from the final build:
#   the GPU suite, the default bench line (what the driver runs), the profile passes of the lattice and the evolved state
#   (kernel trace + PMC: traffic JSONs with code hashes), the kernel census of an 8-brick rank, the deterministic mode per kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
python -m pytest tests -m gpu -q --durations=10 > $O/r06_final_pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/r06_final_pytest_gpu.txt
( time python bench.py ) > $O/r06_final_bench_c3.json 2> $O/r06_final_bench_c3.err
bash profiles/run_profile.sh r06_z > /dev/null 2>&1
bash profiles/run_profile.sh r06_z_evolved --state evolved > /dev/null 2>&1
bash profiles/r06_census.sh r06_z > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
MPMHIP_DETERMINISTIC=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_det -o t -- \
  python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-virtual --no-evolved --no-c5 > $O/r06_z_det.log 2>&1
cp /tmp/prof_det/t_kernel_stats.csv $O/r06_z_det_kernel_stats.csv
tail -n 3 $O/r06_final_pytest_gpu.txt
