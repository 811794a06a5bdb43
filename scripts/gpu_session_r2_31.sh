#!/bin/bash
# Round 2, GPU session 31: PMC passes (HBM-side bytes) of the fused layer-tail kernel.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/s31; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 5 60 leann_amd/lib/bin/kbench 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
timeout -k 10 300 bash scripts/pmc_tail.sh s31
