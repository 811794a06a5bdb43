#!/bin/bash
# Round 6, GPU session 7: generation 2 of the fused kernel with the attending wave at raised priority (s_setprio 2 / 3) against none.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6s7; rm -rf "$OUT"; mkdir -p "$OUT"
for kb in kbench_prio0 kbench kbench_prio3; do
  for fl in 256 ""; do
    if [ -n "$fl" ]; then export KBENCH_FIXED_LEN=$fl; else unset KBENCH_FIXED_LEN; fi
    KBENCH_QB_STAMPS=1 timeout -k 5 120 leann_amd/lib/bin/$kb 262107 10 fusedqa 2>&1 | grep -E "stamps|round\": [12]" | sed "s/^/{\"build\": \"$kb\", \"lengths\": \"${fl:-N(180,50)}\", \"row\": /; s/$/}/" | tee -a $OUT/kbench_fusedqa_gen2_priority.jsonl | cut -c1-1000
  done
done
