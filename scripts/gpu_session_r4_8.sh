#!/bin/bash
# Round 4, GPU session 8: BASELINE configs[2] (10M chunks, PQ traversal + deferred rerank) with the ground truth computed in blocks
# (leann_amd/exact.py), full bench line with diagnosis; then BASELINE configs[4] at its STATED size (10M chunks, bge-base shape, B = 1024) once.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r4s8; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
timeout -k 5 60 $KB 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
KBENCH_TAIL4_ONLY=1 timeout -k 5 120 $KB 262107 10 tail4 > $OUT/box_probe_tail4.jsonl 2>/dev/null; echo "box probe: gen4 $(grep '"variant": "0", "round": 2' $OUT/box_probe_tail4.jsonl | grep -o '"us": [0-9.]*') gen3 $(grep 'generation 3)", "round": 2' $OUT/box_probe_tail4.jsonl | grep -o '"us": [0-9.]*')"
timeout -k 10 700 python scripts/bench_c3.py --diagnose --steps 3 --warmup 1 --cpu-baseline-queries 4 > $OUT/bench_c3_10M.json 2> $OUT/bench_c3_10M.err; echo "c3 rc=$?"
grep -E "^\[c3\]" $OUT/bench_c3_10M.err | cut -c1-1200
cut -c1-1500 $OUT/bench_c3_10M.json
timeout -k 10 1300 python bench.py --config c5 --steps 2 --warmup 1 --no-latency-rows --no-min-ef-step --no-table-roofline --no-provider-ab --cpu-baseline-seconds 10 > $OUT/bench_c5_10M.json 2> $OUT/bench_c5_10M.err; echo "c5 rc=$?"
tail -25 $OUT/bench_c5_10M.err | cut -c1-400
cut -c1-3000 $OUT/bench_c5_10M.json
