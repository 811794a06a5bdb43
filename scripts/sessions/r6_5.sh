#!/bin/bash
# Round 6, GPU session 5: generation 2 of the fused QKV + attention kernel (two wave groups half a head apart) against generation 1 and the pair, three length
# profiles; its GPU tests; bench.py with it on.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6s5; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
for fl in "" 256 128; do
  if [ -n "$fl" ]; then export KBENCH_FIXED_LEN=$fl; else unset KBENCH_FIXED_LEN; fi
  timeout -k 5 120 $KB 262107 10 fusedqa 2>&1 | grep -v '"kbench"' | sed "s/^/{\"lengths\": \"${fl:-N(180,50)}\", \"row\": /; s/$/}/" | tee -a $OUT/kbench_fusedqa_gen2.jsonl | cut -c1-300
done
unset KBENCH_FIXED_LEN
timeout -k 10 400 python -m pytest tests/test_gpu_encoder_kernels.py tests/test_gpu_native_provider.py -m gpu -q > $OUT/pytest_encoder.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/pytest_encoder.log)"; grep -E "^(FAILED|ERROR)" $OUT/pytest_encoder.log | head -20 | cut -c1-250
timeout -k 10 400 python bench.py --gpus 1 --steps 3 --warmup 1 --no-latency-rows --no-min-ef-step --no-provider-ab --no-table-roofline --no-cpu-baseline > $OUT/bench_c2_fused_gen2.json 2> $OUT/bench_c2_fused_gen2.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    r = json.load(open("gpurun_out/r6s5/bench_c2_fused_gen2.json"))
    print("value", r["value"], "recall", r["recall_at_10"], "no-memo", r["without_call_memo"]["value"], "roofline", r["roofline"]["frac"], "encoder", r["roofline_encoder"]["frac"])
    print("  probe", json.dumps(r["roofline"].get("box_probe"))[:1200])
    print("  kernels", json.dumps(r.get("encoder_kernels_profiled_step"))[:900])
    print("  parity", json.dumps(r.get("parity_check"))[:400], r.get("extras_errors"))
except Exception as e:
    print("no bench json:", e)
PY
tail -2 $OUT/bench_c2_fused_gen2.err | cut -c1-300
