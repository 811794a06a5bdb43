#!/bin/bash
# Round 6, GPU session 26: C3 (BASELINE.json configs[2]: 10M chunks, DiskANN-style PQ traversal W = 64 + one deferred recompute rerank) again, round 5's command,
# on the round's last kernels (k_pq_traverse without its three serial load chains, survivors placed by counting).
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6s26; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 1500 python scripts/bench_c3.py --chunks 10000000 --steps 3 --warmup 1 --M 32 --efc 200 --rerank-expanded 0 --cpu-baseline-queries 3 > $OUT/bench_c3_10M.json 2> $OUT/bench_c3_10M_log.txt; echo "c3 rc=$?"
python - <<'PY'
import json
try:
    r = json.load(open("gpurun_out/r6s26/bench_c3_10M.json"))
    print("value", r["value"], "ms_per_step", r["ms_per_step"], "recall", r.get("recall_at_10"), "roofline", r["roofline"]["frac"])
    print("traversal", json.dumps(r.get("roofline_traversal"))[:700]); print(json.dumps(r.get("parity_check"))[:600]); print(json.dumps(r.get("config"))[:400])
except Exception as e:
    print("no json:", e)
PY
tail -3 $OUT/bench_c3_10M_log.txt | cut -c1-300
