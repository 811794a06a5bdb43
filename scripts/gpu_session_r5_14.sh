#!/bin/bash
# Round 5, GPU session 14 (~10 GPU-minutes): the round's reference run -- the driver's bench command under rocprofv3 --kernel-trace --stats (per-kernel
# table of the same run), smoke(), SQ counters of the final attention kernel, which kind of box this is (layer tail 675-690 us = fast; clocks beside it).
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r5s14; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
bash scripts/probe_box.sh > $OUT/box.txt 2>&1
( sleep 4; /opt/rocm/bin/rocm-smi --showclocks --showpower > $OUT/rocm_smi_under_tail4.txt 2>&1 ) &
KBENCH_TAIL4_ONLY=1 timeout -k 5 120 $KB 262107 30 tail4 > $OUT/tail4.jsonl 2> $OUT/tail4.err; wait
echo "tail4: $(grep -o '"round": 2, "us": [0-9.]*' $OUT/tail4.jsonl | head -3 | tr '\n' ' ')"
slow=$(python - <<'PY'
import json
us = [json.loads(l)["us"] for l in open("gpurun_out/r5s14/tail4.jsonl") if '"lm_layer_tail_h384_f16"' in l and '"round": 2' in l]
print(1 if us and us[0] > 800 else 0)
PY
)
if [ "$slow" = "1" ]; then  # a SLOW box: the first experiment towards what they have in common -- L2 hit / miss / fabric requests of the same command
  KBENCH_TAIL4_ONLY=1 bash scripts/pmc_pass.sh r5s14 tail4_tcc_slow_box tail4 262107 -- TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE 2>&1 | tail -12
fi
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 10 560 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o c2 -- python bench.py --gpus 1 --steps 6 --warmup 2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "bench rc=$?"
find $OUT/prof -name "*kernel_trace.csv" -size +8M -delete; find $OUT/prof -name "*.db" -delete
python - <<'PY'
import csv, glob, json
try:
    r = json.load(open("gpurun_out/r5s14/bench_c2.json"))
    print("value", r["value"], "recall", r["recall_at_10"], "no-memo", r["without_call_memo"]["value"], "roofline", r["roofline"]["frac"], "encoder", r["roofline_encoder"]["frac"])
    print(json.dumps(r.get("encoder_kernels_profiled_step"))[:900])
    print(json.dumps(r.get("small_batch_latency"))[:900])
    print(json.dumps(r.get("parity_check"))[:600]); print(json.dumps(r.get("cpu_baseline"))[:400]); print(r.get("extras_errors"))
except Exception as e:
    print("no bench json:", e)
f = glob.glob("gpurun_out/r5s14/prof/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    for x in rows[:12]:
        print(f'{x["Name"][:90]:90s} calls={x["Calls"]:>7s} avg_us={float(x["AverageNs"])/1e3:9.2f} pct={x["Percentage"]}')
PY
tail -3 $OUT/bench_c2.err | cut -c1-300
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $OUT/smoke.log | cut -c1-200)"
export KBENCH_ATTN_ONLY=0
bash scripts/pmc_pass.sh r5s14 attn3_final_sq_a attn 262107 -- SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE 2>&1 | grep -A12 k_attn_varlen | head -14
bash scripts/pmc_pass.sh r5s14 attn3_final_fetch attn 262107 -- FETCH_SIZE 2>&1 | grep -A4 k_attn_varlen | head -6
bash scripts/pmc_pass.sh r5s14 attn3_final_write attn 262107 -- WRITE_SIZE 2>&1 | grep -A4 k_attn_varlen | head -6
