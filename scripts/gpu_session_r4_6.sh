#!/bin/bash
# Round 4, GPU session 6: BASELINE configs[2] at 10M chunks with the corpus density of the headline config (topic count scaled with the
# corpus: 1000 chunks per topic); a 5-second probe of the layer tail first (is this box in the slow mode of sessions 3 / 4?).
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r4s6; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
timeout -k 5 60 $KB 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
KBENCH_TAIL4_ONLY=1 timeout -k 5 120 $KB 262107 10 tail4 > $OUT/box_probe_tail4.jsonl 2>/dev/null; echo "box probe: gen4 $(grep '"variant": "0", "round": 2' $OUT/box_probe_tail4.jsonl | grep -o '"us": [0-9.]*') gen3 $(grep 'generation 3)", "round": 2' $OUT/box_probe_tail4.jsonl | grep -o '"us": [0-9.]*')"
timeout -k 10 1100 python scripts/bench_c3.py --diagnose --steps 3 --warmup 1 --cpu-baseline-queries 4 > $OUT/bench_c3_10M.json 2> $OUT/bench_c3_10M.err; echo "c3 rc=$?"
grep -E "^\[c3\]" $OUT/bench_c3_10M.err | cut -c1-1500
cut -c1-2500 $OUT/bench_c3_10M.json
