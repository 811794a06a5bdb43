#!/bin/bash
# Round 3, GPU session 15 (the round's last ~100 GPU-seconds): the C5 shape (bge-base 768-d) at a corpus small enough to fit them --
# 100k chunks, 64 queries per step -- for the small-batch latency rows of the general-width library-side provider
# (lm_recompute_create_general / lm_bert_forward_packed) against the Python provider on the same index.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/s15; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 3 100 python bench.py --config c5 --chunks 100000 --batch 64 --steps 1 --warmup 1 --no-cpu-baseline --no-table-roofline --no-parity-check \
    --no-min-ef-step > $OUT/bench_c5_100k.json 2> $OUT/bench_c5_100k.err
echo "rc=$?"; tail -2 $OUT/bench_c5_100k.err | cut -c1-600
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/s15/bench_c5_100k.json").read().strip().splitlines()[-1])
    print(json.dumps({k: d.get(k) for k in ("value", "recall_at_10", "ms_per_step", "without_call_memo", "small_batch_latency", "small_batch_latency_python_provider",
                                            "full_step_over_the_library_side_provider", "roofline_encoder", "extras_errors")})[:3500])
except Exception as ex:
    print("bench json:", ex)
PY
