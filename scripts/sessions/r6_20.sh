#!/bin/bash
# Round 6, GPU session 20: k_pq_traverse with a code row's pieces requested together and four expansion passes in flight -- the PQ GPU tests, then scripts/bench_c3.py at
# 1M chunks (C3's graph / L / W / m) on the new library and on the library built from the commit before (leann_amd/lib_before_pq/, built on the CPU box), same box.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6s20; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 300 python -m pytest tests/test_gpu_pq.py tests/test_gpu_plugin_callers.py tests/test_abi.py -m gpu -q > $OUT/pytest_pq.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/pytest_pq.log)"; grep -E "^(FAILED|ERROR)" $OUT/pytest_pq.log | head | cut -c1-250
run() {
  timeout -k 10 500 python scripts/bench_c3.py --chunks 1000000 --steps 3 --warmup 1 --M 32 --efc 200 --rerank-expanded 0 --cpu-baseline-queries 2 > $OUT/bench_c3_1M_$1.json 2> $OUT/bench_c3_1M_$1.err; echo "c3 $1 rc=$?"
  python - "$1" <<'PY'
import json, sys
try:
    r = json.load(open(f"gpurun_out/r6s20/bench_c3_1M_{sys.argv[1]}.json"))
    print(sys.argv[1], "value", r["value"], "recall", r.get("recall_at_10"), "traversal", json.dumps(r.get("roofline_traversal"))[:700]); print(json.dumps(r.get("parity_check"))[:500])
except Exception as e:
    print("no json:", e)
PY
}
run new
cp leann_amd/lib/libleann_mi355x.so /tmp/new.so && cp leann_amd/lib_before_pq/libleann_mi355x.so leann_amd/lib/libleann_mi355x.so && touch leann_amd/lib/libleann_mi355x.so
run before
cp /tmp/new.so leann_amd/lib/libleann_mi355x.so
run new_again
