#!/bin/bash
# Round 6, GPU session 14: the fused QKV + attention kernel with the second-round blocks of 5- and 6-block sequences shared among the waves that would idle
# (helpers project K / V and attend a share of the key tiles; split-K merge by the owner) against the plain form and the pair; its GPU tests.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6s14; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
for fl in "" 150 180 256; do
  if [ -n "$fl" ]; then export KBENCH_FIXED_LEN=$fl; else unset KBENCH_FIXED_LEN; fi
  timeout -k 5 120 $KB 262107 10 fusedqa 2>&1 | grep -v '"kbench"' | sed "s/^/{\"lengths\": \"${fl:-N(180,50)}\", \"row\": /; s/$/}/" | tee -a $OUT/kbench_fusedqa_role_split.jsonl | cut -c1-300
done
unset KBENCH_FIXED_LEN
timeout -k 10 400 python -m pytest tests/test_gpu_encoder_kernels.py tests/test_gpu_native_provider.py tests/test_gpu_plugin_callers.py -m gpu -q > $OUT/pytest_encoder.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/pytest_encoder.log)"; grep -E "^(FAILED|ERROR)" $OUT/pytest_encoder.log | head -20 | cut -c1-250
