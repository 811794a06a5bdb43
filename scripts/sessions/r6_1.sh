#!/bin/bash
# Round 6, GPU session 1: (a) the wait-state A/B VERDICT r5 asked for first -- layer tail and QKV kernel with LM_DMA_NOP=3 (product) against 0,
# alternating processes; (b) the whole GPU suite on HEAD (incl. the new dynamic-batching parity tests); (c) bench.py with the box probe (TCC pass
# forced: this box's numbers become the fast-box reference if it is one); (d) FETCH_SIZE pass of the stored-embedding search, 8192 distinct queries;
# (e) SURVEY 8(d)'s two variants on the current kernels; (f) the B = 1 latency frontier on the 200k index.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6s1; rm -rf "$OUT"; mkdir -p "$OUT"
bash scripts/probe_box.sh > $OUT/box.txt 2>&1
KB=leann_amd/lib/bin/kbench
for rep in 1 2 3; do
  for v in "" _nop0; do
    KBENCH_TAIL4_ONLY=1 timeout -k 5 60 ${KB}$v 262107 10 tail4 2>/dev/null | grep '"round"' | sed "s/^/{\"build\": \"nop${v:-3}\", \"rep\": $rep, \"row\": /; s/$/}/" >> $OUT/kbench_dma_wait_states_ab.jsonl
    timeout -k 5 60 ${KB}$v 262107 10 qkv 2>/dev/null | grep -E '"round"|lm_qkv' | sed "s/^/{\"build\": \"nop${v:-3}\", \"rep\": $rep, \"row\": /; s/$/}/" >> $OUT/kbench_dma_wait_states_ab.jsonl
  done
done
python - <<'PY'
import json, collections
acc = collections.defaultdict(list)
for ln in open("gpurun_out/r6s1/kbench_dma_wait_states_ab.jsonl"):
    try:
        r = json.loads(ln)
    except Exception:
        continue
    row = r["row"]
    if "us" in row and row.get("round", 0) >= 1:
        acc[(row["kernel"], str(row.get("variant", "")), r["build"])].append(row["us"])
for k, v in sorted(acc.items()):
    print(k, "n", len(v), "min", min(v), "median", sorted(v)[len(v) // 2], "max", max(v))
PY
timeout -k 10 420 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$? $(tail -1 $OUT/pytest_gpu.log)"; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log | head -20 | cut -c1-250
BENCH_FORCE_TCC_PASS=1 timeout -k 10 560 python bench.py --gpus 1 --steps 6 --warmup 2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    r = json.load(open("gpurun_out/r6s1/bench_c2.json"))
    print("value", r["value"], "recall", r["recall_at_10"], "no-memo", r["without_call_memo"]["value"], "roofline", r["roofline"]["frac"], "encoder", r["roofline_encoder"]["frac"])
    print("box_probe", json.dumps(r["roofline"].get("box_probe"))[:2500])
    print(json.dumps(r.get("small_batch_latency"))[:700])
    print("frontier", json.dumps(r.get("small_batch_latency_frontier"))[:6000])
    print("table", json.dumps(r.get("roofline_table_mode"))[:1500])
    print(json.dumps(r.get("parity_check"))[:500]); print(r.get("extras_errors"))
except Exception as e:
    print("no bench json:", e)
PY
tail -3 $OUT/bench_c2.err | cut -c1-300
bash scripts/pmc_table_mode.sh r6s1 2>&1 | tail -4 | cut -c1-2500
timeout -k 10 240 python bench.py --fixed-len 256 --steps 2 --warmup 1 --no-latency-rows --no-min-ef-step --no-provider-ab --no-box-probe --cpu-baseline-seconds 5 > $OUT/bench_c2_fixed_len_256.json 2> $OUT/bench_c2_fixed_len_256.err; echo "fixed-len rc=$? $(cut -c1-260 $OUT/bench_c2_fixed_len_256.json)"
timeout -k 10 200 python scripts/bench_table_provider.py > $OUT/bench_c2_table_provider.json 2> $OUT/bench_c2_table_provider.err; echo "table provider rc=$? $(cut -c1-300 $OUT/bench_c2_table_provider.json)"
timeout -k 10 300 python scripts/latency_bench.py --frontier > $OUT/latency_frontier_200k.json 2> $OUT/latency_frontier_200k.err; echo "frontier rc=$?"
python - <<'PY'
import json
try:
    r = json.load(open("gpurun_out/r6s1/latency_frontier_200k.json"))
    f = r["small_batch_latency_frontier"]
    for c in f["cells"]:
        print(c)
    print("best", f["best_at_recall_0.9"]); print(f.get("batch_256")); print(r["latency"])
except Exception as e:
    print("no frontier json:", e)
PY
