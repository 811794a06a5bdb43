#!/bin/bash
# Round 4, GPU session 3: the layer tail with the one-pass LayerNorms and the six-stage W_o ring (residual rows deferred), the
# weight-streaming QKV kernel (two waves per SIMD) against the weight-stationary one, the whole `pytest -m gpu` suite, a short bench.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r4s3; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
timeout -k 5 60 $KB 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
timeout -k 5 200 $KB 262107 20 qkv > $OUT/kbench_qkv.jsonl 2> $OUT/kbench_qkv.err; echo "== qkv rc=$?"; cat $OUT/kbench_qkv.jsonl | cut -c1-300; tail -2 $OUT/kbench_qkv.err
timeout -k 5 300 $KB 262107 20 tail4 > $OUT/kbench_tail4.jsonl 2> $OUT/kbench_tail4.err; echo "== tail4 rc=$?"
grep -v '"round": 0' $OUT/kbench_tail4.jsonl | cut -c1-700; tail -3 $OUT/kbench_tail4.err
timeout -k 10 500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$? $(tail -1 $OUT/pytest_gpu.log)"; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log | head -20
timeout -k 10 400 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-table-roofline > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r4s3/bench_c2.json").read().strip().splitlines()[-1])
    keep = ("value", "recall_at_10", "ms_per_step", "roofline", "roofline_encoder", "without_call_memo", "encoder_kernels_profiled_step", "small_batch_latency", "extras_errors", "parity_check")
    print(json.dumps({k: d.get(k) for k in keep})[:5000])
except Exception as ex:
    print("bench json:", ex)
PY
tail -3 $OUT/bench_c2.err | cut -c1-300
