#!/bin/bash
# Round 2, GPU session 35: LayerNorm-1 gamma / beta prefetch in the fused layer tail: stamps + the kernel's tests.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/s35; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
timeout -k 5 60 $KB 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
timeout -k 5 200 $KB 262107 20 tailstamps > $OUT/kbench_tail.jsonl 2> $OUT/kbench.err; echo "== tail rc=$?"; cut -c1-700 $OUT/kbench_tail.jsonl; tail -3 $OUT/kbench.err
timeout -k 10 300 python -m pytest tests/test_gpu_encoder_kernels.py -m gpu -q -k "fused_attention_output or layer_tail or every_kernel" > $OUT/pytest_gpu.log 2>&1; echo "rc=$? $(tail -1 $OUT/pytest_gpu.log)"; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log | head
