#!/bin/bash
# Round 5, GPU session 10 (~2.5 GPU-minutes): attention generation 3 with the dependent VALU chains of a tile broken up (row maximum as a tree, two
# accumulation chains for the row sums): kbench of all variants, phase stamps, tests.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r5s10; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
timeout -k 10 300 python -m pytest tests/test_gpu_encoder_kernels.py -m gpu -q -k "attention" > $OUT/pytest_attention.log 2>&1; rc=$?; echo "pytest attention rc=$rc $(tail -1 $OUT/pytest_attention.log)"
if [ $rc -ne 0 ]; then grep -E "^E  |^FAILED" $OUT/pytest_attention.log | head -30 | cut -c1-300; fi
timeout -k 5 120 $KB 262107 20 attn > $OUT/kbench_attn.jsonl 2>&1; cut -c1-260 $OUT/kbench_attn.jsonl
for v in 4 7; do KBENCH_A3_STAMP_VARIANT=$v timeout -k 5 60 $KB 262107 5 a3stamps > $OUT/kbench_a3stamps_$v.jsonl 2>&1; grep '"waves_with_query_blocks": 2' $OUT/kbench_a3stamps_$v.jsonl | cut -c1-800; done
