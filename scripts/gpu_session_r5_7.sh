#!/bin/bash
# Round 5, GPU session 7 (~21 GPU-minutes): (1) s_memtime stamps of the attention kernel (where a wave's 20 k cycles go), (2) BASELINE configs[4]
# (C5) at its stated size -- 10M chunks, bge-base shape, 1024 queries per step -- WITH the extras round 4's run lost to its time limit: memo-off step,
# parity block on the run's own index, CPU baseline, and the per-rank batch of an 8-GPU run (128 queries) as an extra row.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r5s7; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
timeout -k 5 90 $KB 262107 10 attnstamp > $OUT/kbench_attnstamp.jsonl 2>&1; cut -c1-900 $OUT/kbench_attnstamp.jsonl
timeout -k 5 90 $KB 1048000 5 attnstamp > $OUT/kbench_attnstamp_1M.jsonl 2>&1; tail -1 $OUT/kbench_attnstamp_1M.jsonl | cut -c1-900
timeout -k 10 1500 python bench.py --config c5 --steps 2 --warmup 1 --no-latency-rows --no-min-ef-step --no-table-roofline --no-provider-ab --cpu-baseline-seconds 10 --extra-batches 128 > $OUT/bench_c5_10M.json 2> $OUT/bench_c5_10M.log; echo "c5 rc=$?"
tail -4 $OUT/bench_c5_10M.log | cut -c1-600
python - <<'PY'
import json
try:
    r = json.load(open("gpurun_out/r5s7/bench_c5_10M.json"))
    for k in ("value", "ms_per_step", "recall_at_10", "without_call_memo", "roofline", "roofline_encoder", "parity_check", "cpu_baseline", "extra_batch_rows", "setup_s", "extras_errors"):
        print(k, json.dumps(r.get(k))[:700])
except Exception as e:
    print("no c5 json:", e)
PY
