#!/bin/bash
# Round 2, GPU session 30: sub-batch token budget of the recompute provider, fused layer tail on / off, one round's worth of chunks.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/s30; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 5 60 leann_amd/lib/bin/kbench 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
timeout -k 10 300 python scripts/subbatch_bench.py > $OUT/subbatch.jsonl 2> $OUT/subbatch.err; echo "rc=$?"; cat $OUT/subbatch.jsonl; tail -3 $OUT/subbatch.err
