#!/bin/bash
# Round 6, GPU session 22: C5 (BASELINE.json configs[4]: 10M chunks, bge-base shape, fp16 recompute, 1024 queries per step) again on the round's last kernels
# (residual rows of lm_gemm_f16 requested ahead; counting merge in k_update).
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6s22; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 1700 python bench.py --config c5 --steps 2 --warmup 1 --no-latency-rows --no-min-ef-step --no-table-roofline --no-provider-ab --cpu-baseline-seconds 10 --extra-batches 128 > $OUT/bench_c5_10M.json 2> $OUT/bench_c5_10M_log.txt; echo "c5 rc=$?"
python - <<'PY'
import json
try:
    r = json.load(open("gpurun_out/r6s22/bench_c5_10M.json"))
    print("value", r["value"], "ms_per_step", r["ms_per_step"], "recall", r.get("recall_at_10"), "roofline", r["roofline"]["frac"], r["roofline"]["achieved"], "encoder", r.get("roofline_encoder", {}).get("frac"))
    print(json.dumps(r.get("encoder_kernels_profiled_step"))[:600]); print(json.dumps(r.get("parity_check"))[:400]); print(json.dumps(r["roofline"].get("box_probe"))[:600])
except Exception as e:
    print("no c5 json:", e)
PY
tail -3 $OUT/bench_c5_10M_log.txt | cut -c1-400
