#!/bin/bash
# Round 5, GPU session 19 (the round's last ~2 GPU-minutes): the FINAL attention kernel (max tree through fmaxf again) in kbench, two rounds beside generation 2.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r5s19; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 5 60 leann_amd/lib/bin/kbench 262107 20 attn > $OUT/kbench_attn.jsonl 2>&1; cut -c1-260 $OUT/kbench_attn.jsonl
