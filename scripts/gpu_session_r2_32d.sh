#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/s32d; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 5 60 leann_amd/lib/bin/kbench 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
timeout -k 10 150 python scripts/diag_768.py > $OUT/diag.jsonl 2> $OUT/diag.err; echo "rc=$?"; cat $OUT/diag.jsonl; tail -3 $OUT/diag.err
