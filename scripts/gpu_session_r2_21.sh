#!/bin/bash
# Round 2, GPU session 21: the fused layer tail (out-projection + LN + MLP + LN in one kernel) against the three launches it
# replaces (timing, stamps, reference check), then the whole -m gpu suite without -x.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/s21; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
timeout -k 5 60 $KB 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
timeout -k 5 200 $KB 262107 20 tailstamps > $OUT/kbench_tail.jsonl 2> $OUT/kbench_tail.err; echo "== tail rc=$?"; cut -c1-900 $OUT/kbench_tail.jsonl; tail -3 $OUT/kbench_tail.err
timeout -k 5 100 $KB 4000 5 tail > $OUT/kbench_tail_small.jsonl 2>> $OUT/kbench_tail.err; echo "== tail small rc=$?"; cut -c1-400 $OUT/kbench_tail_small.jsonl
echo "== pytest -m gpu"
timeout -k 10 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "rc=$? $(tail -1 $OUT/pytest_gpu.log)"; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log | head -20
