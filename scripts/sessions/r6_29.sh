#!/bin/bash
# Round 6, GPU session 29: the token budget of ONE forward of the library-side provider under sustained load (scripts/forward_size_ab.py):
# does a forward whose activations stay inside the 256 MB Infinity Cache (32 k ... 131 k tokens) buy clocks at the power cap, and does that
# outweigh the worse fill of the chip?
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6s29; rm -rf "$OUT"; mkdir -p "$OUT"
T0=$(date +%s)
timeout -k 10 600 python scripts/forward_size_ab.py > $OUT/forward_size_ab.jsonl 2> $OUT/forward_size_ab.err; echo "ab rc=$? in $(( $(date +%s) - T0 )) s"
tail -3 $OUT/forward_size_ab.err
cat $OUT/forward_size_ab.jsonl | cut -c1-420
