#!/bin/bash
# Round 3, GPU session 4: GEMM phase stamps; the reworked PQ traversal on hardware (tests + C3 at 2M chunks with the complexity sweep);
# small-batch latency with the layer on the general kernels (many small workgroups) against the fused kernels (one 77 us workgroup chain).
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/s4; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
timeout -k 5 60 $KB 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
LEANN_MI355X_GEMM_VARIANT=7 timeout -k 5 120 $KB 262107 3 gemmstamp > $OUT/kbench_gemm_stamps.jsonl 2> $OUT/kbench.err; echo "== stamps rc=$?"; grep stamps $OUT/kbench_gemm_stamps.jsonl | cut -c1-400
for v in 2 6; do
  LEANN_MI355X_GEMM_VARIANT=$v timeout -k 5 150 $KB 262107 10 gemmf16 > $OUT/kbench_gemm_var$v.jsonl 2>> $OUT/kbench.err
  echo "== variant $v rc=$?"; grep '"round": 1' $OUT/kbench_gemm_var$v.jsonl | grep lm_gemm_f16 | cut -c1-230
done
timeout -k 10 500 python -m pytest tests/test_gpu_pq.py tests/test_gpu_plugin_callers.py tests/test_distributed.py -m gpu -q > $OUT/pytest.log 2>&1; echo "tests rc=$? $(tail -1 $OUT/pytest.log)"; grep -E "^(FAILED|ERROR)|Error" $OUT/pytest.log | head
for v in default 2; do
  if [ $v = default ]; then timeout -k 10 300 python scripts/latency_bench.py 2> /dev/null | tail -1 | cut -c1-400
  else LEANN_MI355X_GEMM=$v timeout -k 10 300 python scripts/latency_bench.py 2> /dev/null | tail -1 | cut -c1-400; fi
done
timeout -k 10 600 python scripts/bench_c3.py --chunks 2000000 --steps 3 --warmup 1 > $OUT/bench_c3_2M.json 2> $OUT/bench_c3.err; echo "c3 rc=$?"; grep "complexity sweep" $OUT/bench_c3.err | cut -c1-400; cut -c1-1500 $OUT/bench_c3_2M.json
