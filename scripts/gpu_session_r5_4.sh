#!/bin/bash
# Round 5, GPU session 4 (~8 GPU-minutes): first hardware contact of the generation-3 attention kernel (csrc/lm_attn_v3.hip):
# GPU tests of every variant vs fp32 torch, kbench timing of generation 2 and the four variants, SQ counters of two variants, a short bench.py.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r5s4; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
timeout -k 10 300 python -m pytest tests/test_gpu_encoder_kernels.py -m gpu -q -x -k "attention" > $OUT/pytest_attention.log 2>&1; rc=$?; echo "pytest attention rc=$rc $(tail -1 $OUT/pytest_attention.log)"
if [ $rc -ne 0 ]; then tail -40 $OUT/pytest_attention.log | cut -c1-300; fi
timeout -k 5 120 $KB 262107 20 attn > $OUT/kbench_attn.jsonl 2>&1; cut -c1-260 $OUT/kbench_attn.jsonl
for v in 0 3; do
  export KBENCH_ATTN_ONLY=$v
  bash scripts/pmc_pass.sh r5s4 attn3_var${v}_sq_a attn 262107 -- SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE 2>&1 | grep -A12 k_attn_varlen | head -14
  bash scripts/pmc_pass.sh r5s4 attn3_var${v}_sq_b attn 262107 -- SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT 2>&1 | grep -A12 k_attn_varlen | head -14
done
unset KBENCH_ATTN_ONLY
if [ $rc -eq 0 ]; then
  timeout -k 10 400 python bench.py --gpus 1 --steps 3 --warmup 1 --no-latency-rows --no-min-ef-step --no-table-roofline --no-provider-ab --no-parity-check --no-cpu-baseline > $OUT/bench_c2_short.json 2> $OUT/bench_c2_short.err; echo "bench rc=$?"
  python - <<'PY'
import json
try:
    r = json.load(open("gpurun_out/r5s4/bench_c2_short.json"))
    print("value", r["value"], "recall", r["recall_at_10"], "no-memo", r["without_call_memo"]["value"], "roofline", r["roofline"]["frac"], "encoder", r["roofline_encoder"]["frac"])
    print(json.dumps(r.get("encoder_kernels_profiled_step"))[:900])
except Exception as e:
    print("no bench json:", e)
PY
  tail -3 $OUT/bench_c2_short.err | cut -c1-300
fi
