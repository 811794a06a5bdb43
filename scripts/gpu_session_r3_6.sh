#!/bin/bash
# Round 3, GPU session 6: the persistent form of lm_gemm_f16 (tests, timing against the one-tile-per-workgroup grid, stamps), bge-base
# encoder A/B, SQ counters of the two MFMA kernels, the SURVEY 8(d) table-provider variant at 1M vectors, C5 at 500k chunks.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/s6; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
timeout -k 5 60 $KB 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
timeout -k 10 600 python -m pytest tests/test_gpu_encoder_kernels.py -m gpu -q -x -k "general_gemm or hidden_768 or one_call" > $OUT/pytest.log 2>&1; echo "tests rc=$? $(tail -1 $OUT/pytest.log)"; grep -E "^(FAILED|ERROR)|Error|assert " $OUT/pytest.log | head
for grid in cus tiles cus tiles; do
  LEANN_MI355X_GEMM_GRID=$grid timeout -k 5 150 $KB 262107 10 gemmf16 > $OUT/kbench_gemm_$grid.jsonl 2>> $OUT/kbench.err
  echo "== grid $grid rc=$?"; grep '"round": 1' $OUT/kbench_gemm_$grid.jsonl | grep lm_gemm_f16 | cut -c1-230
done
LEANN_MI355X_GEMM_VARIANT=7 timeout -k 5 120 $KB 262107 3 gemmstamp > $OUT/kbench_gemm_stamps.jsonl 2>> $OUT/kbench.err; echo "== stamps rc=$?"; grep stamps $OUT/kbench_gemm_stamps.jsonl | cut -c1-400
timeout -k 10 400 python scripts/encoder_switch_ab.py bge-base-en-v1.5 4096 1048576 - LEANN_MI355X_GEMM=0 2> /dev/null | cut -c1-240
bash scripts/pmc_sq.sh s6/pmc_tail tail 2>&1 | tail -40 | cut -c1-200
bash scripts/pmc_sq.sh s6/pmc_gemm gemmstamp 2>&1 | tail -30 | cut -c1-200
timeout -k 10 400 python scripts/bench_table_provider.py > $OUT/bench_table_provider_1M.json 2> $OUT/bench_table.err; echo "table rc=$?"; cut -c1-1800 $OUT/bench_table_provider_1M.json
timeout -k 10 900 python bench.py --config c5 --chunks 500000 --batch 256 --steps 2 --warmup 1 > $OUT/bench_c5_500k.json 2> $OUT/bench_c5.err; echo "c5 rc=$?"; python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/s6/bench_c5_500k.json"))
    print(json.dumps({k: d.get(k) for k in ("value", "recall_at_10", "ms_per_step", "roofline", "roofline_encoder", "parity_check", "cpu_baseline")})[:3000])
except Exception as ex:
    print("c5 json:", ex)
PY
tail -5 $OUT/bench_c5.err | cut -c1-300
