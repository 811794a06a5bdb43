#!/bin/bash
# Round 3, GPU session 13 (the round's last 9.7 GPU-minutes: ONE call).  In order of importance, each step under its own timeout and
# with its own log under gpurun_out/s13, so that a call cut short still leaves what ran:
#   1. the new code on hardware: the built-in recompute provider (tests/test_gpu_native_provider.py), the tests whose default path
#      changed with it (CLS pooling kernel, hidden-768 one-call forward, default forward), the recompute parity tests (per-call memo default);
#   2. the driver's bench command, short (1 timed step): the JSON line of HEAD incl. the provider A/B latency rows;
#   3. whatever time is left: the rest of `pytest -m gpu`.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/s13; rm -rf "$OUT"; mkdir -p "$OUT"
T0=$(date +%s)
timeout -k 5 200 python -m pytest tests/test_gpu_native_provider.py tests/test_gpu_encoder_kernels.py tests/test_gpu_parity.py -m gpu -q \
    -k "native or meanpool or hidden_768 or default_forward or recompute or memo" > $OUT/pytest_new_code.log 2>&1
echo "step1 rc=$? $(tail -1 $OUT/pytest_new_code.log) [$(( $(date +%s) - T0 )) s]"; grep -E "^(FAILED|ERROR)|Error|assert" $OUT/pytest_new_code.log | head -12
timeout -k 5 260 python bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_c2_short.json 2> $OUT/bench_c2.err
echo "step2 bench rc=$? [$(( $(date +%s) - T0 )) s]"; tail -2 $OUT/bench_c2.err | cut -c1-400
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/s13/bench_c2_short.json").read().strip().splitlines()[-1])
    keys = ("value", "recall_at_10", "ms_per_step", "without_call_memo", "small_batch_latency", "small_batch_latency_python_provider",
            "full_step_over_the_library_side_provider", "extras_errors")
    print(json.dumps({k: d.get(k) for k in keys})[:3000])
    print("parity:", json.dumps(d.get("parity_check"))[:600])
except Exception as ex:
    print("bench json:", ex)
PY
timeout -k 5 240 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_native_provider.py > $OUT/pytest_gpu_rest.log 2>&1
echo "step3 rc=$? $(tail -1 $OUT/pytest_gpu_rest.log) [$(( $(date +%s) - T0 )) s]"; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu_rest.log | head
