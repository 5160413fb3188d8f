R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_big -o t -- python $R/cpic_big.py > /tmp/big.log 2>&1
python3 - <<PY
import csv
for r in list(csv.DictReader(open("/tmp/prof_big/t_kernel_stats.csv")))[:9]:
    n=r["Name"].split("(")[0].replace("void ","")
    print("%-44s calls %5s avg %9.2f us"%(n[:44], r["Calls"], float(r["AverageNs"])/1e3))
PY
