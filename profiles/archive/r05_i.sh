#!/bin/bash
# round 5, step i: tiled tests again (totals fix), k_cell_table<32> between 2 M and 6 M slots (2 / 4 bricks of C3), then the PMC + trace
# passes of the final kernels on both states (-> profiles/r05_k_*, traffic_c3*.json with the kernels' code hashes) and k_rank's outlier
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_tiled.py -m gpu -q -x -k "energy or rccl_binding or plan_equals" > $O/r05_i_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r05_i_pytest.log
tail -3 $O/r05_i_pytest.log
line() { grep '^{' | tail -1; }
for rep in 1 2; do
for V in ct32 ct64; do
  case $V in ct32) E="X=1";; ct64) E="MPMHIP_CT_BLOCKS=64";; esac
  for K in 2 4; do
    env $E MPMHIP_TILE_OVERLAP=0 python bench.py --virtual $K --steps 24 --warmup 8 2>/dev/null | line > $O/r05_i_v${K}_${V}_$rep.json
  done
done
done
python - <<'P'
import json, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
r = lambda p: {k: round(v * 1e3, 1) for k, v in p.items()}
for f in sorted(glob.glob(O + "/r05_i_v*_[12].json")):
    d = json.load(open(f))
    print("%-24s per rank %.4f ms %s" % (os.path.basename(f), d["per_rank_ms_serial_no_events"], r(d["rank0_phases_ms"])))
P
bash profiles/run_profile.sh r05_k > /dev/null 2>&1
bash profiles/run_profile.sh r05_k_evolved --state evolved > /dev/null 2>&1
ls $O/r05_k $O/r05_k_evolved
cat $O/r05_k/pmc_summary.txt | head -40
python - <<'P'
import csv, os
p = "/tmp/prof_r05_k/trace/t_kernel_trace.csv"
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(p)) if "k_rank" in r["Kernel_Name"]]
rows.sort()
d = [x[1] / 1e3 for x in rows]
print("k_rank dispatches %d: first five %s us; max %.1f us at dispatch %d; median %.1f us" % (len(d), [round(x, 1) for x in d[:5]], max(d), d.index(max(d)), sorted(d)[len(d) // 2]))
P
