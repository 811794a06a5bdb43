#!/bin/bash
# Round 2, GPU session 37: last check of the tree as committed -- encoder kernel tests, end-to-end pipeline tests, smoke().
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/s37; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 5 60 leann_amd/lib/bin/kbench 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
timeout -k 10 300 python -m pytest tests/test_gpu_encoder_kernels.py tests/test_gpu_pipeline.py tests/test_config1_golden.py -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "rc=$? $(tail -1 $OUT/pytest_gpu.log)"; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log | head
timeout -k 10 100 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "rc=$? $(tail -1 $OUT/smoke.log | cut -c1-100)"
