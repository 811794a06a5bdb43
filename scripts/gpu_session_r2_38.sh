#!/bin/bash
# Round 2, GPU session 38: SURVEY 8(d) fixed-length variant of C2 (every chunk 256 tokens), one timed step.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/s38; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 150 python bench.py --fixed-len 256 --steps 1 --warmup 1 --no-cpu-baseline --no-latency-rows --no-min-ef-step --no-table-roofline --no-parity-check > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?"
tail -c 2500 $OUT/bench.json | cut -c1-2500; tail -2 $OUT/bench.err | cut -c1-600
