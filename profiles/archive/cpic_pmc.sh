#!/bin/bash
# SQ counters of the transfer kernels of the 8 M CPIC scene (one rocprofv3 --pmc pass; mean of the last 12 dispatches)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/cpp; mkdir -p gpurun_out/cpp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
  --output-format csv -d gpurun_out/cpp -o p -- python profiles/cpic_scene_8m.py > gpurun_out/cpp/out.txt 2>&1
python profiles/summarize_pmc.py --last 12 gpurun_out/cpp/*/p_counter_collection.csv gpurun_out/cpp/p_counter_collection.csv 2>/dev/null | grep -A12 -E "rigid|k_g2p<|k_p2g<|gather" | head -120
