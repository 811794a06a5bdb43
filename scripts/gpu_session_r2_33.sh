#!/bin/bash
# Round 2, GPU session 33: BASELINE configs[4] (C5: bge-base 768-d fp16 recompute) at a REDUCED corpus (500k chunks instead of 10M), 256 queries per step.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/s33; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 5 60 leann_amd/lib/bin/kbench 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
timeout -k 10 330 python bench.py --config c5 --chunks 500000 --batch 256 --steps 1 --warmup 1 --no-min-ef-step --no-latency-rows --no-table-roofline --cpu-baseline-seconds 5 > $OUT/c5.json 2> $OUT/c5.err; echo "c5 rc=$?"; tail -c 3500 $OUT/c5.json; tail -4 $OUT/c5.err | cut -c1-1500
