#!/bin/bash
# Round 4, GPU session 12 (last): the GPU suite on the tree with the small-forward limit at 16384 tokens, then the latency rows at the default.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r4s12; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 230 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$? $(tail -1 $OUT/pytest_gpu.log)"; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log | head -20
LAT_BATCHES=1,4,16,64,256 timeout -k 10 85 python scripts/latency_bench.py > $OUT/latency_default.json 2> $OUT/latency_default.err; echo "latency rc=$?"; cut -c1-1800 $OUT/latency_default.json
