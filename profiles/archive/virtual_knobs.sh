#!/bin/bash
# profiles/virtual_knobs.sh K "ENV=VAL ..." ... — on the GPU box: one `bench.py --virtual K` run per argument (each a set of
# environment switches), printing the per-rank substep (no events anywhere) and the per-phase table of rank 0
K=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "$@"; do
  out=$(env $cfg python $R/bench.py --virtual $K --steps 30 --warmup 6 2>/dev/null | tail -1)
  python - "$cfg" "$out" <<'PY'
import json, sys
cfg, line = sys.argv[1], sys.argv[2]
try:
    d = json.loads(line)
    ph = d["rank0_phases_ms"]
    print("%-60s per-rank %.4f ms | lvl4 %.4f | sort %.1f p2g+pack %.1f grid %.1f g2p %.1f us" % (
        cfg, d["per_rank_ms_serial_no_events"], sum(d["per_rank_compute_ms"]) / len(d["per_rank_compute_ms"]),
        1e3 * ph["sort"], 1e3 * ph["p2g"], 1e3 * ph["grid"], 1e3 * ph["g2p"]))
except Exception as e:
    print("%-60s FAILED %r %s" % (cfg, e, line[-300:]))
PY
done
