#!/bin/bash
# Round 6, GPU session 8: the whole GPU suite on HEAD, smoke(), then the driver's command (python bench.py --gpus 1 --steps 20 --warmup 5, no profiler).
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6s8; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 420 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$? $(tail -1 $OUT/pytest_gpu.log)"; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log | head -20 | cut -c1-250
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $OUT/smoke.log | cut -c1-160)"
T0=$(date +%s)
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_c2_driver_command.json 2> $OUT/bench_c2_driver_command.err; echo "bench rc=$? in $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
try:
    r = json.load(open("gpurun_out/r6s8/bench_c2_driver_command.json"))
    print("value", r["value"], "ms_per_step", r["ms_per_step"], "recall", r["recall_at_10"], "no-memo", r["without_call_memo"]["value"], "roofline", r["roofline"]["frac"], r["roofline"]["avg_launch_us"], "encoder", r["roofline_encoder"]["frac"])
    print("box_probe", json.dumps(r["roofline"].get("box_probe"))[:1800])
    print("b1 best", r["roofline"].get("b1_best_at_recall_0.9"))
    f = r.get("small_batch_latency_frontier") or {}
    print("b256", f.get("batch_256"))
    print("table", json.dumps({k: v for k, v in (r.get("roofline_table_mode") or {}).items() if "auto" in k or k in ("traffic", "queries_in_flight")})[:1800])
    print(json.dumps(r.get("small_batch_latency"))[:500]); print(json.dumps(r.get("parity_check"))[:400]); print(r.get("cpu_baseline", {}).get("value"), r.get("extras_errors"))
except Exception as e:
    print("no bench json:", e)
PY
tail -2 $OUT/bench_c2_driver_command.err | cut -c1-300
