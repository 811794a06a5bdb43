#!/bin/bash
# Round 6, GPU session 18: s_memtime stamps of lm_gemm_f16 (diagnosis variant 7) for the 8-wave and the four-wave shape: main loop / epilogue phases per tile.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6s18; rm -rf "$OUT"; mkdir -p "$OUT"
LEANN_MI355X_GEMM_VARIANT=7 timeout -k 5 200 leann_amd/lib/bin/kbench 65536 3 gemmstamp 2>&1 | grep -v '"kbench"' | tee $OUT/kbench_gemm_stamps_two_shapes.jsonl | cut -c1-600
