#!/bin/bash
# round 5, final numbers on one box (step q = step j again on the last build, plus the last dispatches of every kernel in both states): the whole GPU suite + smoke, the default bench line as the driver runs it (C3, CPU baseline included),
# C2, C5, the K-brick virtual-rank lines with and without the overlap split, and the per-kernel census of the final build
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
( timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -6; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ) > $O/r05_final_pytest_gpu.txt
cat $O/r05_final_pytest_gpu.txt
line() { grep '^{' | tail -1; }
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | line > $O/r05_final_bench_c3.json
python bench.py --config c2 --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | line > $O/r05_final_bench_c2.json
python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | line > $O/r05_final_bench_c5.json
for K in 2 4 8; do
  MPMHIP_TILE_OVERLAP=0 python bench.py --virtual $K --steps 24 --warmup 8 2>/dev/null | line > $O/r05_virtual${K}_nosplit.json
  MPMHIP_TILE_OVERLAP=1 python bench.py --virtual $K --steps 24 --warmup 8 2>/dev/null | line > $O/r05_virtual${K}.json
  bash profiles/virtual_census.sh r05_v$K $K > /dev/null
done
cd /tmp && export TMPDIR=/tmp
for C in c2 c3; do
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$C -o t -- python $R/bench.py --config $C --steps 20 --warmup 5 --no-cpu-baseline --no-evolved > $O/r05_final_${C}_trace.log 2>&1
  python $R/profiles/loop_census.py /tmp/tr_$C/t_kernel_trace.csv 12 > $O/r05_final_${C}_census.txt 2>&1
  python $R/profiles/last_calls.py /tmp/tr_$C/t_kernel_trace.csv 20 > $O/r05_final_${C}_last_calls.txt 2>&1
done
bash $R/profiles/evolved_trace.sh r05_final_c3_evolved > /dev/null 2>&1
cd $R
python - <<'P'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
r = lambda p: {k: round(v * 1e3, 1) for k, v in p.items()}
for f in sorted(glob.glob(O + "/r05_final_bench_*.json")) + sorted(glob.glob(O + "/r05_virtual*.json")):
    d = json.load(open(f))
    if "K" in d:
        print("%-28s per rank %.4f ms (split %s) %s" % (os.path.basename(f), d["per_rank_ms_serial_no_events"], d["overlap_split"], r(d["rank0_phases_ms"])))
    else:
        ev = d.get("evolved") or {}
        print("%-28s %.4f %s frac %.3f traffic %s | evolved %.4f %s" % (os.path.basename(f), d["ms_per_step"], r(d["phases_ms_per_step"]), d["roofline"]["frac"], d["roofline"].get("traffic"), ev.get("ms_per_step", 0), r(ev.get("phases_ms_per_step", {}))))
P
# trace + PMC passes of the same build (profiles/run_profile.sh): the committed kernel statistics and roofline.traffic sources
bash $R/profiles/run_profile.sh r05_q > /dev/null 2>&1
bash $R/profiles/run_profile.sh r05_q_evolved --state evolved > /dev/null 2>&1
head -8 $O/r05_q/kernel_stats.csv | cut -c1-60,200-400
