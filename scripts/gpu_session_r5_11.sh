#!/bin/bash
# Round 5, GPU session 11 (~8 GPU-minutes): (1) attention: the kernel against itself with a scheduling barrier / a discarded s_memtime at its phase
# boundaries (why is the stamped diagnosis build 10 % faster?); (2) the whole `pytest -m gpu` suite on the tree as it is.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r5s11; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
timeout -k 5 120 $KB 262107 20 attn > $OUT/kbench_attn.jsonl 2>&1; cut -c1-260 $OUT/kbench_attn.jsonl
timeout -k 5 60 $KB 262107 5 a3stamps > $OUT/kbench_a3stamps.jsonl 2>&1; grep '"waves_with_query_blocks": 2' $OUT/kbench_a3stamps.jsonl | cut -c1-800
timeout -k 10 600 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$? $(tail -1 $OUT/pytest_gpu.log)"; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log | head -20 | cut -c1-250
