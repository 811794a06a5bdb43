#!/bin/bash
# Round 3, GPU session 10: lm_gemm_f16 with the de-phased DMA issue (diagnosis variant 8) against the default, interleaved; then C3 at its
# full 10M chunks (set-up ~8 min: corpus, 10M bge-small forwards, GPU graph build, PQ).
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/s10; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
timeout -k 5 60 $KB 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
for v in 0 8 0 8; do
  LEANN_MI355X_GEMM_VARIANT=$v timeout -k 5 150 $KB 262107 10 gemmf16 > $OUT/kbench_gemm_var${v}.jsonl 2>> $OUT/kbench.err
  echo "== variant $v rc=$?"; grep '"round": 1' $OUT/kbench_gemm_var${v}.jsonl | grep lm_gemm_f16 | cut -c1-150
done
timeout -k 10 1100 python scripts/bench_c3.py --chunks 10000000 --steps 3 --warmup 1 > $OUT/bench_c3_10M.json 2> $OUT/bench_c3.err; echo "c3 rc=$?"; grep -E "complexity sweep|setup|embedded|flat graph" $OUT/bench_c3.err | cut -c1-400; cut -c1-2500 $OUT/bench_c3_10M.json
