#!/bin/bash
# Round 2, GPU session 28: weight DMA issued inside the slot stream (one piece behind every fourth MFMA) vs all twelve at the top of the iteration.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/s28; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
timeout -k 5 60 $KB 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
for i in 1 2; do
  LEANN_MI355X_ABLATE=2048 timeout -k 5 200 $KB 262107 20 tail > $OUT/kbench_tail_top_$i.jsonl 2>> $OUT/kbench.err; echo "== top-of-iteration DMA rc=$?"; grep one.launch $OUT/kbench_tail_top_$i.jsonl | cut -c1-330
  timeout -k 5 200 $KB 262107 20 tail > $OUT/kbench_tail_spread_$i.jsonl 2>> $OUT/kbench.err; echo "== in-slot DMA rc=$?"; grep one.launch $OUT/kbench_tail_spread_$i.jsonl | cut -c1-330
done
timeout -k 5 200 $KB 262107 20 tailstamps > $OUT/kbench_tail_stamps.jsonl 2>> $OUT/kbench.err; grep stamps $OUT/kbench_tail_stamps.jsonl | cut -c1-900
timeout -k 5 200 $KB 262107 20 mlp > $OUT/kbench_mlp.jsonl 2>> $OUT/kbench.err; grep "variant 3" $OUT/kbench_mlp.jsonl | cut -c1-300
tail -3 $OUT/kbench.err
