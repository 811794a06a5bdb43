#!/bin/bash
# Round 5, GPU session 12 (~2 GPU-minutes): attention tile loop aligned to 64 / 128 bytes (LEANN_MI355X_ATTN3 = 1 / 2) against the kernel as it is
# and its stamped build -- is the stamped build's 10 % a matter of where the loop sits in the instruction stream?
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r5s12; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
timeout -k 5 120 $KB 262107 20 attn > $OUT/kbench_attn.jsonl 2>&1; cut -c1-260 $OUT/kbench_attn.jsonl
timeout -k 5 60 $KB 262107 5 a3stamps > $OUT/kbench_a3stamps.jsonl 2>&1; grep '"waves_with_query_blocks": 2' $OUT/kbench_a3stamps.jsonl | cut -c1-300
timeout -k 10 120 python -m pytest tests/test_gpu_encoder_kernels.py -m gpu -q -k "attention" > $OUT/pytest_attention.log 2>&1; echo "pytest attention rc=$? $(tail -1 $OUT/pytest_attention.log)"
