#!/bin/bash
# First GPU session of the next round: the opt-in kernel that has only run in emulation so far.
#   1. its GPU tests (marker gpu_next: NOT part of -m gpu)          2. its timing against the two launches it replaces
#   3. the encoder-level A/B (one search round's worth of chunks)   -> make LEANN_MI355X_QKV_IN_TAIL=1 the default only if 2 and 3 win,
#      then move tests/test_gpu_next.py's tests under -m gpu and re-run the reference session (scripts/gpu_session_r2_34.sh).
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/next; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
timeout -k 5 60 $KB 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
timeout -k 5 120 $KB 4000 5 tailqkv > $OUT/kbench_tailqkv_small.jsonl 2> $OUT/kbench.err; echo "== small rc=$?"; grep -E "tail_qkv|ws_h384" $OUT/kbench_tailqkv_small.jsonl | cut -c1-300
timeout -k 5 200 $KB 262107 20 tailqkv > $OUT/kbench_tailqkv.jsonl 2>> $OUT/kbench.err; echo "== 262k rc=$?"; grep -E "tail_qkv|ws_h384" $OUT/kbench_tailqkv.jsonl | cut -c1-300; tail -3 $OUT/kbench.err
timeout -k 10 300 python -m pytest tests/test_gpu_next.py -m gpu_next -q > $OUT/pytest_gpu_next.log 2>&1; echo "rc=$? $(tail -1 $OUT/pytest_gpu_next.log)"; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu_next.log | head
# 4. opt-in slot order of the feed-forward loop (the two products alternate, single first-product chain): interleaved A/B on one box
for i in 1 2; do
  timeout -k 5 200 $KB 262107 20 tail > $OUT/kbench_tail_plain_$i.jsonl 2>> $OUT/kbench.err; echo "== plain order"; grep one.launch $OUT/kbench_tail_plain_$i.jsonl | cut -c1-220
  LEANN_MI355X_ABLATE=4096 timeout -k 5 200 $KB 262107 20 tail > $OUT/kbench_tail_alternating_$i.jsonl 2>> $OUT/kbench.err; echo "== alternating products"; grep one.launch $OUT/kbench_tail_alternating_$i.jsonl | cut -c1-220
done
for v in 0 1; do LEANN_MI355X_QKV_IN_TAIL=$v timeout -k 10 200 python scripts/subbatch_bench.py 2> /dev/null | grep '"fused_layer_tail": "1", "sub_batch_tokens": 524160' | sed "s/^/QKV_IN_TAIL=$v /"; done
# 5. one library call per forward (csrc/lm_encoder_forward.cpp) vs the per-kernel calls: small-batch latency on a 200k-chunk index
for v in 0 1; do LEANN_MI355X_ONECALL=$v timeout -k 10 300 python scripts/latency_bench.py 2> /dev/null | tail -1 | cut -c1-400; done
