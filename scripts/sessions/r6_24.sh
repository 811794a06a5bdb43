#!/bin/bash
# Round 6, GPU session 24: does the fused QKV + attention kernel (a tie with the pair at the corpus' lengths in kbench's bursts of launches) pay under the SUSTAINED,
# power-limited load of the bench (it moves 1.2 GB less per 262 k tokens)?  bench.py with the kernel forced on / forbidden, alternating, one box.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6s24; rm -rf "$OUT"; mkdir -p "$OUT"
for rep in 1 2; do
  for f in 1 0; do
    LEANN_MI355X_FUSED_QKV_ATTN=$f timeout -k 10 400 python bench.py --gpus 1 --steps 6 --warmup 2 --no-latency-rows --no-min-ef-step --no-table-roofline --no-provider-ab --no-cpu-baseline --no-parity-check > $OUT/bench_fused_${f}_rep$rep.json 2> $OUT/bench_fused_${f}_rep$rep.err; echo "fused=$f rep=$rep rc=$?"
    python - $f $rep <<'PY'
import json, sys
try:
    r = json.load(open(f"gpurun_out/r6s24/bench_fused_{sys.argv[1]}_rep{sys.argv[2]}.json"))
    bp = r["roofline"].get("box_probe") or {}
    print("fused", sys.argv[1], "value", r["value"], "ms_per_step", r["ms_per_step"], "recall", r.get("recall_at_10"), "encoder", r.get("roofline_encoder", {}).get("frac"), "clocks", (bp.get("clocks_during_the_timed_steps") or {}).get("sclk_mhz"), "power", (bp.get("clocks_during_the_timed_steps") or {}).get("power_w"), bp.get("profiled_step_us_per_262144_tokens"))
except Exception as e:
    print("no json:", e)
PY
  done
done
