#!/bin/bash
# Round 5, GPU session 9 (~3.5 GPU-minutes): attention generation 3 with K / V staged through registers (bit 0) and row sums on the matrix pipe (bit 1):
# tests of every variant, kbench, phase stamps of variants 0 and 3, whole-encoder A/B; the PQ parity test at C3's shape.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r5s9; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
timeout -k 10 300 python -m pytest tests/test_gpu_encoder_kernels.py tests/test_gpu_pq.py -m gpu -q -k "attention or pq" > $OUT/pytest_attention_pq.log 2>&1; rc=$?; echo "pytest attention + pq rc=$rc $(tail -1 $OUT/pytest_attention_pq.log)"
if [ $rc -ne 0 ]; then grep -E "^E  |^FAILED" $OUT/pytest_attention_pq.log | head -30 | cut -c1-300; fi
timeout -k 5 120 $KB 262107 20 attn > $OUT/kbench_attn.jsonl 2>&1; cut -c1-260 $OUT/kbench_attn.jsonl
for v in 4 7; do KBENCH_A3_STAMP_VARIANT=$v timeout -k 5 60 $KB 262107 5 a3stamps > $OUT/kbench_a3stamps_$v.jsonl 2>&1; grep stamps $OUT/kbench_a3stamps_$v.jsonl | cut -c1-800; done
timeout -k 10 300 python scripts/encoder_switch_ab.py sentence-transformers/all-MiniLM-L6-v2 22000 1048576 "-" "LEANN_MI355X_ATTN3=0" "LEANN_MI355X_ATTN3=1" "LEANN_MI355X_ATTN3=9" > $OUT/encoder_switch_ab.jsonl 2> $OUT/encoder_switch_ab.err; echo "encoder ab rc=$?"; cut -c1-300 $OUT/encoder_switch_ab.jsonl; tail -2 $OUT/encoder_switch_ab.err | cut -c1-300
