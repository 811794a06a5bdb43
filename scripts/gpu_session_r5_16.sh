#!/bin/bash
# Round 5, GPU session 16 (~3 GPU-minutes): attention generation 3 on an instruction diet (packed-fp16 Q prescale, un-canonicalised max tree, packed rescale / epilogue;
# LEANN_MI355X_ATTN3=1) on hardware: tests, kbench, whole encoder.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r5s16; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
timeout -k 10 200 python -m pytest tests/test_gpu_encoder_kernels.py -m gpu -q -k "attention" > $OUT/pytest_attention.log 2>&1; rc=$?; echo "pytest attention rc=$rc $(tail -1 $OUT/pytest_attention.log)"
if [ $rc -ne 0 ]; then grep -E "^E  |^FAILED" $OUT/pytest_attention.log | head -30 | cut -c1-300; fi
timeout -k 5 120 $KB 262107 20 attn > $OUT/kbench_attn.jsonl 2>&1; cut -c1-260 $OUT/kbench_attn.jsonl
timeout -k 10 200 python scripts/encoder_switch_ab.py sentence-transformers/all-MiniLM-L6-v2 22000 1048576 "-" "LEANN_MI355X_ATTN3=1" > $OUT/encoder_switch_ab.jsonl 2> $OUT/encoder_switch_ab.err; echo "encoder ab rc=$?"; cut -c1-300 $OUT/encoder_switch_ab.jsonl; tail -2 $OUT/encoder_switch_ab.err | cut -c1-300
