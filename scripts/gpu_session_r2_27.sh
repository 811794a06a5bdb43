#!/bin/bash
# Round 2, GPU session 27: short bench with the fused layer tail as the default (1 timed step, no CPU baseline / latency rows / table-mode lines).
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/s27; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 5 60 leann_amd/lib/bin/kbench 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
timeout -k 10 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-latency-rows --no-min-ef-step --no-table-roofline > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?"
python - <<'PY'
import json
r = json.loads(open("gpurun_out/s27/bench.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "recall_at_10", "roofline", "roofline_encoder", "parity_check", "with_per_call_recompute_memo", "with_hub_cache", "with_two_level_search", "extras_errors", "setup_s"):
    print(k, json.dumps(r.get(k))[:900])
PY
tail -5 $OUT/bench.err
