#!/bin/bash
# Round 3, GPU session 2: the general GEMM (lm_gemm_f16) and head_dim-64 attention on hardware for the first time:
# tests first, then kernel timings against rocBLAS / the weight-stationary kernel, then encoder-level A/B (MiniLM, bge-base shapes).
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/s2; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
timeout -k 5 60 $KB 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
timeout -k 10 600 python -m pytest tests/test_gpu_encoder_kernels.py -m gpu -q -x -k "general_gemm or head_dim_64 or hidden_768 or general_gemm_switches or one_call" > $OUT/pytest_new.log 2>&1; echo "new tests rc=$? $(tail -1 $OUT/pytest_new.log)"; grep -E "^(FAILED|ERROR)|Error|assert " $OUT/pytest_new.log | head -20
timeout -k 5 300 $KB 262107 10 gemmf16 > $OUT/kbench_gemmf16.jsonl 2> $OUT/kbench.err; echo "== gemmf16 rc=$?"; cut -c1-330 $OUT/kbench_gemmf16.jsonl; tail -3 $OUT/kbench.err
timeout -k 5 200 $KB 262107 10 attn64 > $OUT/kbench_attn64.jsonl 2>> $OUT/kbench.err; echo "== attn64 rc=$?"; cut -c1-300 $OUT/kbench_attn64.jsonl
timeout -k 10 300 python scripts/encoder_switch_ab.py all-MiniLM-L6-v2 11264 524160 - LEANN_MI355X_GEMM=1 LEANN_MI355X_GEMM=2 2> $OUT/ab_minilm.err | tee $OUT/ab_minilm.jsonl | cut -c1-260; tail -2 $OUT/ab_minilm.err
timeout -k 10 400 python scripts/encoder_switch_ab.py bge-base-en-v1.5 4096 524160 - LEANN_MI355X_GEMM=0 LEANN_MI355X_GEMM=0,LEANN_MI355X_ATTN=0 2> $OUT/ab_bge.err | tee $OUT/ab_bge.jsonl | cut -c1-260; tail -2 $OUT/ab_bge.err
