#!/bin/bash
# Round 6, GPU session 17: lm_gemm_f16's four-wave shape (128 x 128 per wave, one wave per SIMD) with every fragment read and DMA piece behind an MFMA,
# against the 8-wave shape and the vendor library's bare product.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6s17; rm -rf "$OUT"; mkdir -p "$OUT"
KBENCH_GEMM_WIDE=1 timeout -k 5 300 leann_amd/lib/bin/kbench 65536 10 gemmf16 2>&1 | grep -v '"kbench"' | tee $OUT/kbench_gemm_wide_interleaved.jsonl | cut -c1-250
