#!/bin/bash
# Round 6, GPU session 19: the four-wave shape with L2 touches three K-tiles ahead: stamps (where the waits went), then timings against the 8-wave shape and the vendor library.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6s19; rm -rf "$OUT"; mkdir -p "$OUT"
LEANN_MI355X_GEMM_VARIANT=7 timeout -k 3 60 leann_amd/lib/bin/kbench 65536 3 gemmstamp 2>&1 | grep -v '"kbench"' | tee $OUT/kbench_gemm_stamps_two_shapes.jsonl | cut -c1-700
KBENCH_GEMM_WIDE=1 timeout -k 3 120 leann_amd/lib/bin/kbench 65536 10 gemmf16 2>&1 | grep -v '"kbench"' | tee $OUT/kbench_gemm_wide_l2_touch.jsonl | cut -c1-250
