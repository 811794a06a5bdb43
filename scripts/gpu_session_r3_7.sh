#!/bin/bash
# Round 3, GPU session 7: persistent lm_gemm_f16 with the bias slice staged through LDS (timing vs one tile per workgroup, stamps),
# bge-base encoder A/B, the table-provider variant with the corrected noise scale, the bench-script tests.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/s7; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
timeout -k 5 60 $KB 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
timeout -k 10 300 python -m pytest tests/test_gpu_encoder_kernels.py -m gpu -q -x -k "general_gemm or hidden_768" > $OUT/pytest_gemm.log 2>&1; echo "gemm tests rc=$? $(tail -1 $OUT/pytest_gemm.log)"
for grid in cus tiles cus; do
  LEANN_MI355X_GEMM_GRID=$grid timeout -k 5 150 $KB 262107 10 gemmf16 > $OUT/kbench_gemm_$grid.jsonl 2>> $OUT/kbench.err
  echo "== grid $grid rc=$?"; grep '"round": 1' $OUT/kbench_gemm_$grid.jsonl | grep lm_gemm_f16 | cut -c1-200
done
LEANN_MI355X_GEMM_VARIANT=7 timeout -k 5 120 $KB 262107 3 gemmstamp > $OUT/kbench_gemm_stamps.jsonl 2>> $OUT/kbench.err; echo "== stamps rc=$?"; grep stamps $OUT/kbench_gemm_stamps.jsonl | cut -c1-400
timeout -k 10 400 python scripts/encoder_switch_ab.py bge-base-en-v1.5 4096 1048576 - LEANN_MI355X_GEMM=0 2> /dev/null | cut -c1-240
timeout -k 10 400 python scripts/bench_table_provider.py > $OUT/bench_table_provider_1M.json 2> $OUT/bench_table.err; echo "table rc=$?"; cut -c1-2200 $OUT/bench_table_provider_1M.json
timeout -k 10 900 python -m pytest tests/test_gpu_bench_scripts.py -m gpu -q > $OUT/pytest_scripts.log 2>&1; echo "script tests rc=$? $(tail -1 $OUT/pytest_scripts.log)"; grep -E "^(FAILED|ERROR)|Error|assert " $OUT/pytest_scripts.log | head
