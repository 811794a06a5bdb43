#!/bin/bash
# Round 5, GPU session 20 (the round's last ~80 GPU-seconds): kernels that use lm_dma16_sv after its s_nop 0 -> s_nop 3 (VALU-written SGPR -> VMEM wait
# states inside the asm block): the general GEMM against its reference on the encoder's shapes, the attention kernel against its reference.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r5s20; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 3 14 leann_amd/lib/bin/kbench 65536 5 attn > $OUT/kbench_attn.jsonl 2>&1; cut -c1-230 $OUT/kbench_attn.jsonl
timeout -k 3 28 leann_amd/lib/bin/kbench 65536 3 gemmf16 > $OUT/kbench_gemmf16.jsonl 2>&1; cut -c1-230 $OUT/kbench_gemmf16.jsonl
