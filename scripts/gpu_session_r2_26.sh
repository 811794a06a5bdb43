#!/bin/bash
# Round 2, GPU session 26: weight-stationary GEMM with the next tile's rows requested in front of the stores; fused layer tail (stagger 40 default).
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/s26; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
timeout -k 5 60 $KB 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
timeout -k 5 200 $KB 262107 20 linear > $OUT/kbench_linear.jsonl 2> $OUT/kbench.err; echo "== linear rc=$?"; cut -c1-300 $OUT/kbench_linear.jsonl; tail -3 $OUT/kbench.err
timeout -k 5 200 $KB 262107 20 tail > $OUT/kbench_tail.jsonl 2>> $OUT/kbench.err; echo "== tail rc=$?"; cut -c1-400 $OUT/kbench_tail.jsonl
