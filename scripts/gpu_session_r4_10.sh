#!/bin/bash
# Round 4, GPU session 10 (the round's last GPU-minutes): B = 1 latency with the speculative prefetch swept on the SAME queries
# (scripts/latency_bench.py, 200k-chunk index), at the default small-forward limit (6144 tokens) and at 16384.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r4s10; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 170 python scripts/latency_bench.py --speculate 0,4,8,16,32,64 > $OUT/latency_speculate.json 2> $OUT/latency_speculate.err; echo "latency rc=$?"; cut -c1-2500 $OUT/latency_speculate.json; tail -2 $OUT/latency_speculate.err | cut -c1-300
LEANN_MI355X_SMALL_TOKENS=16384 timeout -k 10 120 python scripts/latency_bench.py --speculate 0,16,32,64 > $OUT/latency_speculate_small16k.json 2> $OUT/latency_speculate_small16k.err; echo "latency (small-forward limit 16384) rc=$?"; cut -c1-2000 $OUT/latency_speculate_small16k.json
