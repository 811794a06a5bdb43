#!/bin/bash
# Round 6, GPU session 2: first contact of the fused QKV + attention kernel (lm_qkv_attn_h384.hip): kbench fusedqa (fused vs the pair, interleaved; N(180,50)
# lengths, every sequence 256, every sequence 128), its GPU tests, FETCH_SIZE / WRITE_SIZE passes, SQ counters, then bench.py with the kernel on and off.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6s2; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
timeout -k 5 90 $KB 262107 10 fusedqa > $OUT/kbench_fusedqa.jsonl 2> $OUT/kbench_fusedqa.err; cut -c1-260 $OUT/kbench_fusedqa.jsonl; tail -2 $OUT/kbench_fusedqa.err
KBENCH_FIXED_LEN=256 timeout -k 5 90 $KB 262107 10 fusedqa 2>&1 | grep -v kbench | sed 's/^/{"fixed_len": 256, "row": /; s/$/}/' | tee $OUT/kbench_fusedqa_len256.jsonl | cut -c1-230
KBENCH_FIXED_LEN=128 timeout -k 5 90 $KB 262107 10 fusedqa 2>&1 | grep -v kbench | sed 's/^/{"fixed_len": 128, "row": /; s/$/}/' | tee $OUT/kbench_fusedqa_len128.jsonl | cut -c1-230
timeout -k 10 400 python -m pytest tests/test_gpu_encoder_kernels.py tests/test_gpu_native_provider.py -m gpu -q -x > $OUT/pytest_encoder.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/pytest_encoder.log)"; grep -E "^(FAILED|ERROR)|Error|assert" $OUT/pytest_encoder.log | head -20 | cut -c1-250
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE"; do
  set -- $pass; name=$1; shift
  bash scripts/pmc_pass.sh r6s2 fusedqa_$name fusedqa 262107 -- "$@" 2>&1 | grep -A14 -E "k_qkv_attn|k_qkv_h384|k_attn_varlen" | head -48 | cut -c1-200
done
for f in 1 0; do
  LEANN_MI355X_FUSED_QKV_ATTN=$f timeout -k 10 400 python bench.py --gpus 1 --steps 3 --warmup 1 --no-latency-rows --no-min-ef-step --no-provider-ab --no-table-roofline --no-cpu-baseline > $OUT/bench_c2_fused$f.json 2> $OUT/bench_c2_fused$f.err; echo "bench fused=$f rc=$?"
done
python - <<'PY'
import json
for f in (1, 0):
    try:
        r = json.load(open(f"gpurun_out/r6s2/bench_c2_fused{f}.json"))
        print("fused", f, "value", r["value"], "recall", r["recall_at_10"], "no-memo", r["without_call_memo"]["value"], "roofline", r["roofline"]["frac"], "encoder", r["roofline_encoder"]["frac"])
        print("  probe", json.dumps(r["roofline"].get("box_probe"))[:900])
        print("  kernels", json.dumps(r.get("encoder_kernels_profiled_step"))[:900])
        print("  parity", json.dumps(r.get("parity_check"))[:400], r.get("extras_errors"))
    except Exception as e:
        print("no bench json:", f, e)
PY
