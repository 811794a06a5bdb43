#!/bin/bash
# Round 6, GPU session 15: lm_gemm_f16 -- the rotated main loop (barrier in front of a K-tile's LAST k-step, the next K-tile's first fragments and DMA under that step's
# MFMAs), bias as the accumulators' start value, residual rows requested ahead with one wait, and the four-wave 128 x 128-per-wave shape, each against the round-5 form
# and the vendor library's bare product on the encoder's shapes; then the GEMM / hidden-768 GPU tests on the new default.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6s15; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
timeout -k 5 300 $KB 65536 10 gemmf16 2>&1 | grep -v '"kbench"' | tee $OUT/kbench_gemm_forms.jsonl | cut -c1-260
timeout -k 10 400 python -m pytest tests/test_gpu_encoder_kernels.py -m gpu -q -k "gemm or 768 or general or small" > $OUT/pytest_gemm.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/pytest_gemm.log)"; grep -E "^(FAILED|ERROR)" $OUT/pytest_gemm.log | head -20 | cut -c1-250
