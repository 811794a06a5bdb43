#!/bin/bash
# Round 3, GPU session 12 (closing): the whole `pytest -m gpu` suite on the final tree, a short run of the driver's bench command, and C4
# at its real shard size (one rank = one 7.5M-chunk shard of the 60M / 8 configuration).
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/s12; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$? $(tail -1 $OUT/pytest_gpu.log)"; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log | head
timeout -k 10 600 python bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_c2_short.json 2> $OUT/bench_c2.err; echo "bench rc=$?"; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/s12/bench_c2_short.json").read().strip().splitlines()[-1])
    print(json.dumps({k: d.get(k) for k in ("value", "recall_at_10", "ms_per_step", "small_batch_latency", "extras_errors")})[:1500])
except Exception as ex:
    print("bench json:", ex)
PY
timeout -k 10 1000 python scripts/bench_c4.py --chunks 7500000 --steps 3 --warmup 1 > $OUT/bench_c4_one_shard_7p5M.json 2> $OUT/bench_c4.err; echo "c4 rc=$?"; tail -3 $OUT/bench_c4.err | cut -c1-300; cut -c1-2600 $OUT/bench_c4_one_shard_7p5M.json
