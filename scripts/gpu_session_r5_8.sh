#!/bin/bash
# Round 5, GPU session 8 (~2 GPU-minutes): s_memtime stamps of the attention kernel, one record per wave (session 7's shared counters serialised the
# launch: 7 ms instead of 0.25 -- ~12 ns per same-address atomic x 700 k), and the GPU tests added since session 6.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r5s8; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
timeout -k 5 90 $KB 262107 10 a3stamps > $OUT/kbench_a3stamps.jsonl 2>&1; cut -c1-1000 $OUT/kbench_a3stamps.jsonl
timeout -k 5 90 $KB 1048000 5 a3stamps > $OUT/kbench_a3stamps_1M.jsonl 2>&1; grep stamps $OUT/kbench_a3stamps_1M.jsonl | cut -c1-1000
timeout -k 10 300 python -m pytest tests/test_gpu_pq.py -m gpu -q -x > $OUT/pytest_pq.log 2>&1; echo "pytest pq rc=$? $(tail -1 $OUT/pytest_pq.log)"; grep -E "^E  |^FAILED" $OUT/pytest_pq.log | head -10 | cut -c1-300
