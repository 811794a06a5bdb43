#!/bin/bash
# Round 4, GPU session 4: (1) bisect of the layer-tail regression seen in session 3 -- four builds of the diagnosis library that differ
# in lm_layer_tail_h384.hip only: A = HEAD, B = HEAD with the two-pass fp16-parameter LayerNorms of session 2, C = session 2's
# prologue / two-stage W_o ring with HEAD's LayerNorms, D = session 2's file (control); (2) BASELINE configs[2] at its stated size:
# 10M chunks, DiskANN-style, with the graph / quantiser diagnosis, both rerank sets and a second quantiser size.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r4s4; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
timeout -k 5 60 $KB 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
for round in 1 2; do
for v in D A B C; do
  LD_LIBRARY_PATH=$PWD/leann_amd/lib/diag_$v KBENCH_TAIL4_ONLY=1 KBENCH_TAIL4_STAMP=1 timeout -k 5 120 $KB 262107 10 tail4 > $OUT/bisect_${v}_$round.jsonl 2> $OUT/bisect_${v}_$round.err
  echo "== variant $v round $round rc=$?"; grep -E '"round": [12]|stamps' $OUT/bisect_${v}_$round.jsonl | grep -v "three launches" | cut -c1-560
done
done
timeout -k 10 1100 python scripts/bench_c3.py --diagnose --pq-bytes-extra 128 --steps 3 --warmup 1 --cpu-baseline-queries 4 > $OUT/bench_c3_10M.json 2> $OUT/bench_c3_10M.err; echo "c3 rc=$?"
grep -E "^\[c3\]" $OUT/bench_c3_10M.err | cut -c1-1200
cut -c1-3000 $OUT/bench_c3_10M.json
