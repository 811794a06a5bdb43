#!/bin/bash
# Round 3, GPU session 11: sub-batching / small-forward tests after the preamble trim, B = 1 latency again, then C3 at 10M chunks with the
# complexity sweep extended to 2048 (session 10: recall@10 0.83 at L = 512 -- below the metric's bar).
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/s11; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 400 python -m pytest tests/test_gpu_encoder_kernels.py -m gpu -q -x -k "sub_batching or one_call or general_gemm or hidden_768" > $OUT/pytest.log 2>&1; echo "tests rc=$? $(tail -1 $OUT/pytest.log)"; grep -E "^(FAILED|ERROR)|assert " $OUT/pytest.log | head -5
timeout -k 10 300 python scripts/latency_bench.py 2> /dev/null | tail -1 | cut -c1-400
timeout -k 10 1300 python scripts/bench_c3.py --chunks 10000000 --steps 2 --warmup 1 --cpu-baseline-queries 4 > $OUT/bench_c3_10M.json 2> $OUT/bench_c3.err; echo "c3 rc=$?"; grep -E "complexity" $OUT/bench_c3.err | cut -c1-700; python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/s11/bench_c3_10M.json"))
    print(json.dumps({k: d[k] for k in ("value", "recall_at_10", "ms_per_step", "roofline", "per_query", "cpu_baseline", "setup_s")}))
except Exception as ex:
    print("c3 json:", ex)
PY
