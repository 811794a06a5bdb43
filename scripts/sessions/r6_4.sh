#!/bin/bash
# Round 6, GPU session 4: phase stamps of the fused QKV + attention kernel (where a wave's cycles go), FETCH_SIZE pass of the stored-embedding search at
# 32768 distinct queries in flight.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6s4; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
for fl in "" 256 128; do
  if [ -n "$fl" ]; then export KBENCH_FIXED_LEN=$fl; else unset KBENCH_FIXED_LEN; fi
  KBENCH_QA_STAMPS=1 timeout -k 5 120 $KB 262107 10 fusedqa 2>&1 | grep -E "stamps|round\": 2" | sed "s/^/{\"lengths\": \"${fl:-N(180,50)}\", \"row\": /; s/$/}/" | tee -a $OUT/kbench_fusedqa_stamps.jsonl | cut -c1-900
done
unset KBENCH_FIXED_LEN
bash scripts/pmc_table_mode.sh r6s4 32768 2>&1 | tail -3 | cut -c1-2500
