#!/bin/bash
# Round 3, GPU session 5: attention with a head pair per workgroup (A/B), the small-forward layer form in the default launch path
# (B = 1 latency), the PQ traversal at 256 / 512 / 1024 threads per query (C3 at 2M chunks), encoder sub-batch size.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/s5; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
timeout -k 5 60 $KB 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
timeout -k 10 600 python -m pytest tests/test_gpu_encoder_kernels.py tests/test_gpu_pq.py -m gpu -q -x -k "attention or one_call or pq_search or general_gemm_switches" > $OUT/pytest.log 2>&1; echo "tests rc=$? $(tail -1 $OUT/pytest.log)"; grep -E "^(FAILED|ERROR)|Error|assert " $OUT/pytest.log | head
timeout -k 5 200 $KB 262107 10 attn > $OUT/kbench_attn.jsonl 2> $OUT/kbench.err; echo "== attn rc=$?"; grep -v kbench $OUT/kbench_attn.jsonl | cut -c1-260
timeout -k 10 300 python scripts/encoder_switch_ab.py all-MiniLM-L6-v2 11264 524160 LEANN_MI355X_ATTN_HPW=1 LEANN_MI355X_ATTN_HPW=2 2> /dev/null | cut -c1-240
timeout -k 10 300 python scripts/encoder_switch_ab.py all-MiniLM-L6-v2 11264 1048576 - 2> /dev/null | cut -c1-240
timeout -k 10 300 python scripts/latency_bench.py 2> /dev/null | tail -1 | cut -c1-400
LEANN_MI355X_SMALL_TOKENS=0 timeout -k 10 300 python scripts/latency_bench.py 2> /dev/null | tail -1 | cut -c1-400
timeout -k 10 600 python scripts/bench_c3.py --chunks 2000000 --steps 3 --warmup 1 > $OUT/bench_c3_2M.json 2> $OUT/bench_c3.err; echo "c3 rc=$?"; python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/s5/bench_c3_2M.json"))
    print(json.dumps({k: d[k] for k in ("value", "recall_at_10", "roofline", "per_query", "complexity_sweep")}))
except Exception as ex:
    print("c3 json:", ex)
PY
