#!/bin/bash
# Round 2, GPU session 34 (the round's reference run): the whole -m gpu suite, smoke(), then the driver's bench command under
# rocprofv3 --kernel-trace --stats -- one run gives the JSON line and the per-kernel table it has to agree with.
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/s34; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 5 60 leann_amd/lib/bin/kbench 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
echo "== pytest -m gpu"
timeout -k 10 600 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "rc=$? $(tail -1 $OUT/pytest_gpu.log)"; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log | head
echo "== smoke"
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "rc=$? $(tail -1 $OUT/smoke.log | cut -c1-120)"
echo "== rocprofv3 --kernel-trace --stats -- python bench.py"
( cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err ); echo "rc=$?"
python - <<'PY'
import json
r = json.loads(open("gpurun_out/s34/bench.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "recall_at_10", "roofline", "roofline_encoder", "parity_check", "cpu_baseline", "extras_errors"):
    print(k, json.dumps(r.get(k))[:1200])
PY
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $OUT/bench_kernel_stats.csv && head -8 $OUT/bench_kernel_stats.csv | cut -c1-220
