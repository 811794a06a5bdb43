#!/bin/bash
# Round 5, GPU session 1 (~10 GPU-minutes): what needs no new code.
#   1. which kind of box is this (layer tail 675-690 us on "fast" boxes, 950-1240 us on "slow" ones) + clocks beside it
#   2. SQ counters of the attention kernel (no PMC pass of k_attn_varlen_hd32_v2 exists; "VALU bound" is unverified)
#   3. the GPU tests of the three small-forward features whose xfail markers are gone
#   4. small-forward variants A/B'd in one process (scripts/latency_bench.py LAT_VARIANTS)
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r5s1; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
bash scripts/probe_box.sh > $OUT/box.txt 2>&1
/opt/rocm/bin/rocm-smi --showclocks --showpower > $OUT/rocm_smi_idle.txt 2>&1
( sleep 4; /opt/rocm/bin/rocm-smi --showclocks --showpower > $OUT/rocm_smi_under_tail4.txt 2>&1 ) &
KBENCH_TAIL4_ONLY=1 timeout -k 5 120 $KB 262107 30 tail4 > $OUT/tail4.jsonl 2> $OUT/tail4.err; wait
echo "tail4: $(grep -o "\"round\": 2, \"us\": [0-9.]*" $OUT/tail4.jsonl | head -3 | tr "\n" " ")"
timeout -k 5 60 $KB 262107 20 attn > $OUT/kbench_attn.jsonl 2>&1; cut -c1-220 $OUT/kbench_attn.jsonl
export KBENCH_ATTN_DEFAULT_ONLY=1
bash scripts/pmc_pass.sh r5s1 attn_sq_a attn 262107 -- SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE 2>&1 | tail -25
bash scripts/pmc_pass.sh r5s1 attn_sq_b attn 262107 -- SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT 2>&1 | tail -25
bash scripts/pmc_pass.sh r5s1 attn_sq_c attn 262107 -- SQ_INSTS_VALU_TRANS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_THREAD_CYCLES_VALU 2>&1 | tail -25
unset KBENCH_ATTN_DEFAULT_ONLY
cd /tmp; timeout 60 rocprofv3 -L 2>/dev/null | grep -o -E '\b(SQ|TCC|TCP|TA|GRBM)_[A-Z0-9_]+' | sort -u > "${GRAFT_REPO_ROOT:-/root/repo}/$OUT/counter_names.txt"; cd "${GRAFT_REPO_ROOT:-/root/repo}"
wc -l $OUT/counter_names.txt
timeout -k 10 400 python -m pytest tests/test_gpu_encoder_kernels.py tests/test_gpu_parity.py -m gpu -q -x > $OUT/pytest_two_files.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/pytest_two_files.log)"; grep -E "^(FAILED|ERROR)" $OUT/pytest_two_files.log | head
LAT_VARIANTS_ONLY=1 LAT_VARIANTS=default,rowln,slayer,direct,direct+rowln,direct+slayer LAT_BATCHES=1,4,16,64 timeout -k 10 400 python scripts/latency_bench.py > $OUT/latency_variants_200k.json 2> $OUT/latency_variants_200k.err; echo "latency variants rc=$?"; python - <<'PY'
import json
try:
    r = json.load(open("gpurun_out/r5s1/latency_variants_200k.json"))
    for v in r["small_forward_variants"]:
        print(v["variant"], {k: (x["p50_ms"], x["calls_with_the_first_variants_labels"]) for k, x in v.items() if k != "variant"})
except Exception as e:
    print("no latency json:", e)
PY
tail -3 $OUT/latency_variants_200k.err
