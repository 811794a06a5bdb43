#!/bin/bash
# Round 4, GPU session 11: where is the crossover between the small-forward form of a hidden-384 layer (general kernels) and the fused
# form?  Session 10 saw B = 1 p50 44.5 -> 39.2 ms with LEANN_MI355X_SMALL_TOKENS=16384 instead of the default 6144.  Same script, B = 1, 4,
# 16, at 6144 / 16384 / 32768.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r4s11; rm -rf "$OUT"; mkdir -p "$OUT"
for lim in 6144 16384 32768; do
  LAT_BATCHES=1,4,16 LEANN_MI355X_SMALL_TOKENS=$lim timeout -k 10 110 python scripts/latency_bench.py > $OUT/latency_small_$lim.json 2> $OUT/latency_small_$lim.err; echo "limit $lim rc=$?"; cut -c1-1200 $OUT/latency_small_$lim.json
done
