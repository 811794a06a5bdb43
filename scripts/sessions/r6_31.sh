#!/bin/bash
# Round 6, GPU session 31: the benchmark step's encoder time by size of the round's forward (scripts/round_size_profile.py): how much of a step is spent in the thin
# tail of small rounds, where a forward fills a fraction of the chip?
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6s31; rm -rf "$OUT"; mkdir -p "$OUT"
T0=$(date +%s)
timeout -k 10 600 python scripts/round_size_profile.py > $OUT/round_size_profile.jsonl 2> $OUT/round_size_profile.err; echo "rc=$? in $(( $(date +%s) - T0 )) s"
tail -3 $OUT/round_size_profile.err
cat $OUT/round_size_profile.jsonl | cut -c1-600
