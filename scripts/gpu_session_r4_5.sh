#!/bin/bash
# Round 4, GPU session 5: is the layer-tail slow mode (sessions 3 / 4: the SAME source 690 us in one process and 950 us in another)
# the lock-step of the 256 workgroups (bursts on HBM / L2) that the first-round stagger is there to break?  Builds A (HEAD), D (session 2's
# file, slow in session 4), C (fast in session 4), E (A + non-temporal row DMA and output stores) at several stagger spreads; then the
# weight-streaming QKV kernel with its counted waits corrected.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r4s5; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
timeout -k 5 60 $KB 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
for v in D A E C; do
for sg in 40 0 100 137; do
  LEANN_MI355X_STAGGER=$sg LD_LIBRARY_PATH=$PWD/leann_amd/lib/diag_$v KBENCH_TAIL4_ONLY=1 timeout -k 5 120 $KB 262107 10 tail4 > $OUT/st_${v}_$sg.jsonl 2> $OUT/st_${v}_$sg.err
  echo "variant $v stagger $sg: $(grep '"variant": "0", "round": 2' $OUT/st_${v}_$sg.jsonl | grep -o '"us": [0-9.]*') gen3 $(grep 'generation 3)", "round": 2' $OUT/st_${v}_$sg.jsonl | grep -o '"us": [0-9.]*')"
done
done
timeout -k 5 200 $KB 262107 20 qkv > $OUT/kbench_qkv.jsonl 2> $OUT/kbench_qkv.err; echo "== qkv rc=$?"; cat $OUT/kbench_qkv.jsonl | cut -c1-300; tail -2 $OUT/kbench_qkv.err
