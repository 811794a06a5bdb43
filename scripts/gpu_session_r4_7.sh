#!/bin/bash
# Round 4, GPU session 7: (1) the layer tail with the two-stage W_o ring back as the product form (the six-stage ring ran 1085 us instead of
# 675 us on the boxes of sessions 3 / 4 / 6), next to the six-stage build and a build with non-temporal row DMA / output stores, on whatever
# kind of box this is; (2) the QKV kernels on the same box; (3) what the true top-10 of the C3 corpus are made of (scripts/c3_corpus_probe.py).
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r4s7; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
timeout -k 5 60 $KB 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
for round in 1 2; do
for v in diag diag_W6 diag_NT; do
  LD_LIBRARY_PATH=$PWD/leann_amd/lib/$v KBENCH_TAIL4_ONLY=1 KBENCH_TAIL4_STAMP=1 timeout -k 5 120 $KB 262107 10 tail4 > $OUT/tail_${v}_$round.jsonl 2> $OUT/tail_${v}_$round.err
  echo "tail $v (round $round): gen4 $(grep '"variant": "0", "round": 2' $OUT/tail_${v}_$round.jsonl | grep -o '"us": [0-9.]*') gen3 $(grep 'generation 3)", "round": 2' $OUT/tail_${v}_$round.jsonl | grep -o '"us": [0-9.]*')"
done
done
grep -h stamps $OUT/tail_diag_1.jsonl $OUT/tail_diag_W6_1.jsonl | cut -c1-700
timeout -k 5 120 $KB 262107 10 qkv > $OUT/kbench_qkv.jsonl 2> $OUT/kbench_qkv.err; echo "== qkv rc=$?"; cut -c1-260 $OUT/kbench_qkv.jsonl
timeout -k 5 60 $KB 262107 10 bw > $OUT/kbench_bw.jsonl 2>&1; grep -E "copy|read " $OUT/kbench_bw.jsonl | head -4 | cut -c1-200
timeout -k 10 420 python scripts/c3_corpus_probe.py --chunks 4000000 > $OUT/c3_corpus_probe.jsonl 2> $OUT/c3_corpus_probe.err; echo "== corpus probe rc=$?"; cat $OUT/c3_corpus_probe.jsonl; tail -3 $OUT/c3_corpus_probe.err
