#!/bin/bash
# Round 5, GPU session 18 (~8 GPU-minutes): session 17 again after the attention fix (session 17: 9 "same bits" tests failed -- inline-asm max tree on MFMA results without hazard wait states) -- the whole `pytest -m gpu` suite, smoke(), then the driver's command
# (python bench.py --gpus 1 --steps 20 --warmup 5, no profiler).
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r5s18; rm -rf "$OUT"; mkdir -p "$OUT"
T0=$(date +%s)  # the whole session must end inside the 10 GPU-minutes that are left: the bench gets what the tests leave

timeout -k 10 330 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$? $(tail -1 $OUT/pytest_gpu.log)"; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log | head -20 | cut -c1-250
timeout -k 10 90 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $OUT/smoke.log | cut -c1-120)"
LEFT=$((540 - $(date +%s) + T0)); echo "bench gets ${LEFT}s"; STEPS="--steps 20 --warmup 5"; [ "$LEFT" -lt 290 ] && STEPS="--steps 8 --warmup 2"; [ "$LEFT" -gt 180 ] && timeout -k 10 $LEFT python bench.py --gpus 1 $STEPS > $OUT/bench_c2_driver_command.json 2> $OUT/bench_c2_driver_command.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    r = json.load(open("gpurun_out/r5s18/bench_c2_driver_command.json"))
    print("value", r["value"], "ms_per_step", r["ms_per_step"], "recall", r["recall_at_10"], "no-memo", r["without_call_memo"]["value"], "roofline", r["roofline"]["frac"], "encoder", r["roofline_encoder"]["frac"])
    print(json.dumps(r.get("encoder_kernels_profiled_step"))[:800])
    print(json.dumps(r.get("value_by_batch"))[:900])
    print(json.dumps(r.get("with_hub_cache"))[:300], json.dumps(r.get("with_two_level_search"))[:300], json.dumps(r.get("at_min_ef"))[:200])
    print(r.get("extras_errors"))
except Exception as e:
    print("no bench json:", e)
PY
tail -2 $OUT/bench_c2_driver_command.err | cut -c1-300
