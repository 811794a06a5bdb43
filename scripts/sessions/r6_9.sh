#!/bin/bash
# Round 6, GPU session 9: a box survey line (layer-tail probe + clocks + L2 / fabric counter pass), then C5 (BASELINE.json configs[4]: 10M chunks, bge-base
# shape fp16, query batch 1024) again at its stated size -- round 5's evidence file carries a roofline.achieved its own erratum calls wrong (work booked
# twice in front of the > 4 GiB operand split; fixed after that run).
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6s9; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 300 python bench.py --box-survey > $OUT/box_survey.json 2> $OUT/box_survey.err; echo "survey rc=$? $(cut -c1-1500 $OUT/box_survey.json)"
timeout -k 10 1700 python bench.py --config c5 --steps 2 --warmup 1 --no-latency-rows --no-min-ef-step --no-table-roofline --no-provider-ab --cpu-baseline-seconds 10 --extra-batches 128 > $OUT/bench_c5_10M.json 2> $OUT/bench_c5_10M_log.txt; echo "c5 rc=$?"
python - <<'PY'
import json
try:
    r = json.load(open("gpurun_out/r6s9/bench_c5_10M.json"))
    print("value", r["value"], "ms_per_step", r["ms_per_step"], "recall", r["recall_at_10"], "no-memo", r["without_call_memo"]["value"], "roofline", json.dumps(r["roofline"])[:600], "encoder", r["roofline_encoder"]["frac"])
    print(json.dumps(r.get("encoder_kernels_profiled_step"))[:700]); print(json.dumps(r.get("extra_batch_rows"))[:400]); print(json.dumps(r.get("parity_check"))[:500]); print(json.dumps(r.get("cpu_baseline"))[:300], r.get("extras_errors"))
except Exception as e:
    print("no c5 json:", e)
PY
tail -3 $OUT/bench_c5_10M_log.txt | cut -c1-400
