#!/bin/bash
# Round 6, GPU session 35 (the reference run repeated on the tree as the round ends: sessions 29-34 added an A/B switch to the provider and context fields to the bench line): the whole GPU suite, smoke(), the driver's bench command under rocprofv3
# --kernel-trace --stats (per-kernel table of the same run), then the driver's command as the driver runs it (no profiler).
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6s35; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 420 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$? $(tail -1 $OUT/pytest_gpu.log)"; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log | head -20 | cut -c1-250
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $OUT/smoke.log | cut -c1-160)"
( cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}" && timeout -k 10 700 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o c2 -- python bench.py --gpus 1 --steps 6 --warmup 2 > $OUT/bench_c2_under_rocprofv3.json 2> $OUT/bench_c2_under_rocprofv3.err; echo "bench under rocprofv3 rc=$?" )
find $OUT/prof -name "*kernel_trace.csv" -size +8M -delete; find $OUT/prof -name "*.db" -delete
python - <<'PY'
import csv, glob, json
try:
    r = json.load(open("gpurun_out/r6s35/bench_c2_under_rocprofv3.json"))
    print("under rocprofv3: value", r["value"], "recall", r["recall_at_10"], "roofline", r["roofline"]["frac"], r["roofline"]["avg_launch_us"], r["roofline"]["all_launches_of_the_process"])
except Exception as e:
    print("no bench json:", e)
f = glob.glob("gpurun_out/r6s35/prof/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    for x in rows[:10]:
        print(f'{x["Name"][:90]:90s} calls={x["Calls"]:>7s} avg_us={float(x["AverageNs"])/1e3:9.2f} pct={x["Percentage"]}')
PY
T0=$(date +%s)
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_c2_driver_command.json 2> $OUT/bench_c2_driver_command.err; echo "bench rc=$? in $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
try:
    r = json.load(open("gpurun_out/r6s35/bench_c2_driver_command.json"))
    print("value", r["value"], "ms_per_step", r["ms_per_step"], "recall", r["recall_at_10"], "no-memo", r["without_call_memo"]["value"], "roofline", r["roofline"]["frac"], r["roofline"]["avg_launch_us"], "encoder", r["roofline_encoder"]["frac"])
    print("box_probe", json.dumps(r["roofline"].get("box_probe"))[:1500])
    f = r.get("small_batch_latency_frontier") or {}
    print("best", f.get("best_at_recall_0.9")); print("b256", f.get("batch_256"))
    for c in f.get("cells", []):
        if c["ef"] == 64 and c["beam"] == 1: print(c)
    print(json.dumps(r.get("small_batch_latency"))[:600]); print(json.dumps(r.get("parity_check"))[:400]); print(r.get("cpu_baseline", {}).get("value"), r.get("extras_errors"))
except Exception as e:
    print("no bench json:", e)
PY
