#!/bin/bash
# Round 6, GPU session 28: the driver's command once more on the final tree (another call = possibly another box: one more line for the box survey).
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6s28; rm -rf "$OUT"; mkdir -p "$OUT"
T0=$(date +%s)
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_c2_driver_command.json 2> $OUT/bench_c2_driver_command.err; echo "bench rc=$? in $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
try:
    r = json.load(open("gpurun_out/r6s28/bench_c2_driver_command.json"))
    bp = r["roofline"]["box_probe"]
    print("value", r["value"], "ms_per_step", r["ms_per_step"], "recall", r["recall_at_10"], "no-memo", r["without_call_memo"]["value"], "roofline", r["roofline"]["frac"], r["roofline"]["avg_launch_us"], "encoder", r["roofline_encoder"]["frac"], "traffic", r["roofline"]["traffic"], r["roofline"]["traffic_source"][:160])
    print("probe", bp["layer_tail_262107_tokens_us"], bp["box_class"], bp["clocks_during_the_timed_steps"]["sclk_mhz"], bp["clocks_during_the_timed_steps"]["power_w"], bp["profiled_step_us_per_262144_tokens"], bp.get("card") or bp["clocks_during_the_probe_launches"].get("card"))
    print(json.dumps(r.get("parity_check"))[:300])
except Exception as e:
    print("no bench json:", e)
PY
