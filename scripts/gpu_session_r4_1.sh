#!/bin/bash
# Round 4, GPU session 1: generation 4 of the fused layer tail (lm_layer_tail_h384.hip) -- every schedule variant of the diagnosis
# library against generation 3 in one process (interleaved rounds), reference check of each, s_memtime stamps of five of them.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r4s1; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
timeout -k 5 60 $KB 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
timeout -k 5 300 $KB 262107 20 tail4 > $OUT/kbench_tail4.jsonl 2> $OUT/kbench_tail4.err; echo "== tail4 rc=$?"
cat $OUT/kbench_tail4.jsonl | cut -c1-700; tail -3 $OUT/kbench_tail4.err
