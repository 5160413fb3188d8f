#!/bin/bash
# profiles/bench_ab.sh CONFIG "ENV=VAL ..." ... — on the GPU box: one default-style bench.py run per argument (environment
# switches), one line each: substep, phase table, evolved substep + phases
CFG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "$@"; do
  out=$(env $cfg python $R/bench.py --config $CFG --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1)
  python - "$cfg" "$out" <<'PY'
import json, sys
cfg, line = sys.argv[1], sys.argv[2]
try:
    d = json.loads(line)
    f = lambda ph: "sort %.1f p2g %.1f grid %.1f g2p %.1f" % tuple(1e3 * ph[k] for k in ("sort", "p2g", "grid", "g2p"))
    e = d.get("evolved", {})
    print("%-40s %.4f ms [%s] | evolved %.4f ms [%s]" % (cfg, d["ms_per_step"], f(d["phases_ms_per_step"]), e.get("ms_per_step", 0), f(e["phases_ms_per_step"]) if "phases_ms_per_step" in e else ""))
except Exception as ex:
    print("%-40s FAILED %r %s" % (cfg, ex, line[-300:]))
PY
done
