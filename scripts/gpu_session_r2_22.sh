#!/bin/bash
# Round 2, GPU session 22: where the cycles of the fused layer tail's prologue go (ablation stamps) and of the weight-stationary GEMM's tiles.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/s22; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
timeout -k 5 60 $KB 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
timeout -k 5 200 $KB 262107 20 tailstampsprolog > $OUT/kbench_tail.jsonl 2> $OUT/kbench_tail.err; echo "== tail rc=$?"; cut -c1-1000 $OUT/kbench_tail.jsonl; tail -3 $OUT/kbench_tail.err
timeout -k 5 200 $KB 262107 20 wsgemm > $OUT/kbench_ws.jsonl 2> $OUT/kbench_ws.err; echo "== ws rc=$?"; cut -c1-600 $OUT/kbench_ws.jsonl; tail -3 $OUT/kbench_ws.err
