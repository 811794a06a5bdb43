#!/bin/bash
# Round 2, GPU session 29: BASELINE configs[2] (C3, DiskANN-style PQ traversal + deferred rerank) and configs[4] (C5, bge-base 768-d fp16 recompute)
# at REDUCED corpus sizes (2M / 500k chunks instead of 10M: the full-size set-up alone is ~10 GPU-minutes each) -- the lines say so.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/s29; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 5 60 leann_amd/lib/bin/kbench 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
timeout -k 10 400 python scripts/bench_c3.py --chunks 2000000 --steps 5 --warmup 2 > $OUT/c3.json 2> $OUT/c3.err; echo "c3 rc=$?"; tail -c 2500 $OUT/c3.json; tail -4 $OUT/c3.err
timeout -k 10 400 python bench.py --config c5 --chunks 500000 --batch 128 --steps 1 --warmup 1 --no-min-ef-step --no-latency-rows --no-table-roofline --cpu-baseline-seconds 5 > $OUT/c5.json 2> $OUT/c5.err; echo "c5 rc=$?"; tail -c 3000 $OUT/c5.json; tail -4 $OUT/c5.err
