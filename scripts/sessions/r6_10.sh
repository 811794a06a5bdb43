#!/bin/bash
# Round 6, GPU session 10: the layer tail's out-projection -- fragment ring 8 deep and continuous over the slabs (this build) vs 8 deep per slab (kbench_ocont0)
# vs round 5's form (4 deep per slab: kbench_ord4), alternating processes; the QKV kernel with its ring 8 deep; B = 1 latency with dynamic batching against
# the small-forward limit; a box survey line.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6s10; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 300 python bench.py --box-survey > $OUT/box_survey.json 2> $OUT/box_survey.err; echo "survey rc=$? $(cut -c1-1200 $OUT/box_survey.json)"
for rep in 1 2 3; do
  for v in "" _ocont0 _ord4; do
    KBENCH_TAIL4_ONLY=1 timeout -k 5 60 leann_amd/lib/bin/kbench$v 262107 10 tail4 2>/dev/null | grep -E '"round"|max_abs' | sed "s/^/{\"build\": \"tail${v:-_ocont1_ord8}\", \"rep\": $rep, \"row\": /; s/$/}/" >> $OUT/kbench_tail_outproj_ring.jsonl
  done
done
python - <<'PY'
import json, collections
acc = collections.defaultdict(list)
for ln in open("gpurun_out/r6s10/kbench_tail_outproj_ring.jsonl"):
    try: r = json.loads(ln)
    except Exception: continue
    row = r["row"]
    if "us" in row and row.get("round", 0) >= 1 and "layer_tail" in row["kernel"]: acc[r["build"]].append(row["us"])
    if "max_abs_err" in row and "layer_tail" in row["kernel"]: print(r["build"], row)
for k, v in sorted(acc.items()): print(k, "n", len(v), "min", min(v), "median", sorted(v)[len(v) // 2], "max", max(v))
PY
KBENCH_QKV_RD8=1 timeout -k 5 90 leann_amd/lib/bin/kbench 262107 10 qkv 2>&1 | grep -E "weight streaming|ring 8" | tee $OUT/kbench_qkv_ring8.jsonl | cut -c1-200
for lim in 16384 8192 32768 65536; do
  LAT_VARIANTS_ONLY=1 LAT_BATCHES=1 LAT_BATCH_SIZE=128 LAT_VARIANTS="LEANN_MI355X_SMALL_TOKENS=$lim" timeout -k 10 200 python scripts/latency_bench.py 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'small_tokens_limit': $lim, 'batch_size': 128, 'rows': r['small_forward_variants']}))" | tee -a $OUT/latency_b1_batch128_small_forward_limit.jsonl | cut -c1-400
done
