#!/bin/bash
# Round 6, GPU session 6: why generation 2 of the fused kernel does not overlap its two phases -- per-wave stamps by slot part and wave group; fragment
# ring depth 8 (this build) vs 4 (kbench_rd4), both generations.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6s6; rm -rf "$OUT"; mkdir -p "$OUT"
for kb in kbench kbench_rd4; do
  for fl in 256 ""; do
    if [ -n "$fl" ]; then export KBENCH_FIXED_LEN=$fl; else unset KBENCH_FIXED_LEN; fi
    KBENCH_QB_STAMPS=1 timeout -k 5 120 leann_amd/lib/bin/$kb 262107 10 fusedqa 2>&1 | grep -E "stamps|round\": 2" | sed "s/^/{\"build\": \"$kb\", \"lengths\": \"${fl:-N(180,50)}\", \"row\": /; s/$/}/" | tee -a $OUT/kbench_fusedqa_gen2_stamps.jsonl | cut -c1-1000
  done
done
