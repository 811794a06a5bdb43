#!/bin/bash
# Round 6, GPU session 37 (session 36 plus the operand-reuse setting; session 34 plus the mixing kernels: the MFMA stream with the layer tail's side traffic added piece by piece): the matrix-pipe rate the chip sustains under its power cap with nothing else in the way (scripts/mfma_sustained.cpp): the ceiling the
# 1400 W cap leaves under the 2.5 PFLOP/s figure of the roofline rows.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6s37; rm -rf "$OUT"; mkdir -p "$OUT"
T0=$(date +%s)
timeout -k 10 300 python scripts/mfma_sustained.py 4 > $OUT/mfma_sustained.jsonl 2> $OUT/mfma_sustained.err; echo "rc=$? in $(( $(date +%s) - T0 )) s"
tail -3 $OUT/mfma_sustained.err
cat $OUT/mfma_sustained.jsonl | cut -c1-700
