#!/bin/bash
# Round 3, GPU session 3: where does lm_gemm_f16's time go?  Loop variants of the diagnosis build (DMA placement, ablations, the
# four-slot ring with counted waits) on the encoder's GEMM shapes; plus the new plugin / reader tests.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/s3; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
timeout -k 5 60 $KB 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
for v in 0 5 1 2 3 4 6 0 5; do
  LEANN_MI355X_GEMM_VARIANT=$v timeout -k 5 200 $KB 262107 10 gemmf16 > $OUT/kbench_gemm_var$v.jsonl 2>> $OUT/kbench.err
  echo "== variant $v rc=$?"; grep '"round": 1' $OUT/kbench_gemm_var$v.jsonl | grep lm_gemm_f16 | sed 's/"TFLOPs": 0.0, //; s/"rocblas_TFLOPs": 0.0//' | cut -c1-200
done
timeout -k 10 600 python -m pytest tests/test_gpu_plugin_callers.py tests/test_distributed.py tests/test_gpu_pq.py -m gpu -q > $OUT/pytest.log 2>&1; echo "tests rc=$? $(tail -1 $OUT/pytest.log)"; grep -E "^(FAILED|ERROR)|Error" $OUT/pytest.log | head
