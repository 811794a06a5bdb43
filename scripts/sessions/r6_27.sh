#!/bin/bash
# Round 6, GPU session 27: ONE rank's work of C4 (BASELINE.json configs[3]: a 7.5M-chunk shard of the 60M corpus, all 256 queries, exchange + merge in the timed region
# at world size 1) again, round 5's command, on the round's last kernels.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6s27; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 1500 python scripts/bench_c4.py --chunks 7500000 --batch 256 --steps 3 --warmup 1 --cpu-baseline-seconds 15 > $OUT/bench_c4_one_shard_7p5M.json 2> $OUT/bench_c4_one_shard_7p5M_log.txt; echo "c4 rc=$?"
python - <<'PY'
import json
try:
    r = json.load(open("gpurun_out/r6s27/bench_c4_one_shard_7p5M.json"))
    print("value", r["value"], "ms_per_step", r["ms_per_step"], "recall", r.get("recall_at_10"), "roofline", r["roofline"]["frac"], "scaling", r.get("scaling"))
    print(json.dumps(r.get("parity_check"))[:600]); print(json.dumps(r.get("cpu_baseline"))[:300])
except Exception as e:
    print("no json:", e)
PY
tail -3 $OUT/bench_c4_one_shard_7p5M_log.txt | cut -c1-300
