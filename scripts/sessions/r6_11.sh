#!/bin/bash
# Round 6, GPU session 11: what the layer's kernels cost at the sizes of a one-query search round under dynamic batching (12 k ... 50 k tokens per forward:
# 50 - 200 workgroups on a 256-CU chip), large-forward kernels and the general GEMM.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6s11; rm -rf "$OUT"; mkdir -p "$OUT"
for T in 12288 24576 49152; do
  KBENCH_TAIL4_ONLY=1 timeout -k 5 60 leann_amd/lib/bin/kbench $T 20 tail4 2>/dev/null | grep '"round": 2' | sed "s/^/{\"tokens\": $T, \"row\": /; s/$/}/" | tee -a $OUT/kbench_small_forward_sizes.jsonl | cut -c1-220
  timeout -k 5 60 leann_amd/lib/bin/kbench $T 20 qkv 2>/dev/null | grep '"round": 2' | sed "s/^/{\"tokens\": $T, \"row\": /; s/$/}/" | tee -a $OUT/kbench_small_forward_sizes.jsonl | cut -c1-220
  timeout -k 5 60 leann_amd/lib/bin/kbench $T 20 fusedqa 2>/dev/null | grep '"round": 2' | sed "s/^/{\"tokens\": $T, \"row\": /; s/$/}/" | tee -a $OUT/kbench_small_forward_sizes.jsonl | cut -c1-260
  timeout -k 5 90 leann_amd/lib/bin/kbench $T 10 gemmf16 2>/dev/null | grep -E '"round": 1' | sed "s/^/{\"tokens\": $T, \"row\": /; s/$/}/" | tee -a $OUT/kbench_small_forward_sizes.jsonl | cut -c1-300
done
