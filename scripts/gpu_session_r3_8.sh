#!/bin/bash
# Round 3, GPU session 8: lm_gemm_f16 with 256-wide tiles and a partial last column tile on the QKV width of the 384-wide models
# (N = 1152) against the weight-stationary kernel; encoder-level A/B (LEANN_MI355X_GEMM=1: QKV projection through lm_gemm_f16).
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/s8; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
timeout -k 5 60 $KB 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
timeout -k 10 300 python -m pytest tests/test_gpu_encoder_kernels.py -m gpu -q -x -k "general_gemm" > $OUT/pytest_gemm.log 2>&1; echo "gemm tests rc=$? $(tail -1 $OUT/pytest_gemm.log)"; grep -E "^(FAILED|ERROR)|assert " $OUT/pytest_gemm.log | head -5
timeout -k 5 150 $KB 262107 10 gemmf16 > $OUT/kbench_gemm.jsonl 2>> $OUT/kbench.err; echo "== gemmf16 rc=$?"; grep '"round": 1' $OUT/kbench_gemm.jsonl | cut -c1-200
timeout -k 10 300 python scripts/encoder_switch_ab.py all-MiniLM-L6-v2 11264 1048576 - LEANN_MI355X_GEMM=1 2> /dev/null | cut -c1-240
