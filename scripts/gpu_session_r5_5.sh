#!/bin/bash
# Round 5, GPU session 5 (~5 GPU-minutes): generation 4 of the attention kernel (persistent waves, wave-private K / V rings) on hardware:
# tests of every variant, kbench timing, SQ counters, and the whole encoder with attention generations / the sliced QKV -> attention schedule A/B'd.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r5s5; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
timeout -k 10 300 python -m pytest tests/test_gpu_encoder_kernels.py -m gpu -q -k "attention" > $OUT/pytest_attention.log 2>&1; rc=$?; echo "pytest attention rc=$rc $(tail -1 $OUT/pytest_attention.log)"
if [ $rc -ne 0 ]; then grep -E "^E  |^FAILED" $OUT/pytest_attention.log | head -30 | cut -c1-300; fi
timeout -k 5 120 $KB 262107 20 attn > $OUT/kbench_attn.jsonl 2>&1; cut -c1-260 $OUT/kbench_attn.jsonl
timeout -k 5 120 $KB 1048000 10 attn > $OUT/kbench_attn_1M.jsonl 2>&1; grep '"round": 1' $OUT/kbench_attn_1M.jsonl | cut -c1-260
export KBENCH_ATTN_ONLY=4
bash scripts/pmc_pass.sh r5s5 attn4_sq_a attn 262107 -- SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE 2>&1 | grep -A12 k_attn_varlen | head -14
bash scripts/pmc_pass.sh r5s5 attn4_sq_b attn 262107 -- SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT 2>&1 | grep -A12 k_attn_varlen | head -14
unset KBENCH_ATTN_ONLY
timeout -k 10 300 python scripts/encoder_switch_ab.py sentence-transformers/all-MiniLM-L6-v2 22000 1048576 "-" "LEANN_MI355X_ATTN3=9" "LEANN_MI355X_ATTN3=2" "LEANN_MI355X_QKV_SLICE=65536" "LEANN_MI355X_QKV_SLICE=131072" "LEANN_MI355X_QKV_SLICE=32768" "LEANN_MI355X_QKV_SLICE=65536,LEANN_MI355X_ATTN3=9" > $OUT/encoder_switch_ab.jsonl 2> $OUT/encoder_switch_ab.err; echo "encoder ab rc=$?"; cut -c1-300 $OUT/encoder_switch_ab.jsonl; tail -2 $OUT/encoder_switch_ab.err | cut -c1-300
