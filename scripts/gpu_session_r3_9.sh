#!/bin/bash
# Round 3, GPU session 9 (the reference session of the round): the whole `pytest -m gpu` suite, then the driver's bench command under
# rocprofv3 --kernel-trace --stats (fewer steps than the driver's 20 to stay inside the session budget).
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/s9; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$? $(tail -1 $OUT/pytest_gpu.log)"; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log | head
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $OUT/smoke.log | cut -c1-200)"
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o c2 -- python bench.py --gpus 1 --steps 6 --warmup 2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "bench rc=$?"
find $OUT/prof -name "*kernel_trace.csv" -size +8M -delete; find $OUT/prof -name "*.db" -delete
python - <<'PY'
import csv, glob, json
try:
    d = json.loads(open("gpurun_out/s9/bench_c2.json").read().strip().splitlines()[-1])
    print(json.dumps({k: d.get(k) for k in ("value", "recall_at_10", "ms_per_step", "roofline", "roofline_encoder", "small_batch_latency", "parity_check", "cpu_baseline", "extras_errors")})[:3500])
except Exception as ex:
    print("bench json:", ex)
f = glob.glob("gpurun_out/s9/prof/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    for r in rows[:14]:
        print(f'{r["Name"][:110]:110s} calls={r["Calls"]:>7s} total_ms={float(r["TotalDurationNs"])/1e6:10.2f} avg_us={float(r["AverageNs"])/1e3:9.2f} pct={r["Percentage"]}')
PY
tail -3 $OUT/bench_c2.err | cut -c1-300
