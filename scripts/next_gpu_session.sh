#!/bin/bash
# First GPU call of the next session: validate the opt-in encoder kernels, time them, and (if they pass) run the bench
# with them switched on.  Everything lands under gpurun_out/next/.  Usage (from the repo root, on the GPU box):
#     bash scripts/next_gpu_session.sh            # ~6-8 GPU-minutes
# Each step has its own timeout so that a hang costs one step, not the call.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/next
mkdir -p $OUT
export TMPDIR=/tmp

echo "== 1. gpu_next tests (one pytest process per kernel family: a device fault in one does not hide the others)" | tee $OUT/summary.txt
for k in attention layernorm meanpool embed pack_tokens "fused_mlp" linear "every_opt_in"; do
    timeout 300 python -m pytest tests/test_gpu_next.py -m gpu_next -q -x -k "$k" > $OUT/test_$k.log 2>&1
    echo "   $k: rc=$? $(tail -1 $OUT/test_$k.log)" | tee -a $OUT/summary.txt
done

echo "== 2. kernel A/B" | tee -a $OUT/summary.txt
timeout 400 python -m leann_amd.autotune --device 0 > $OUT/autotune.jsonl 2> $OUT/autotune.err; echo "   autotune rc=$? $(tail -1 $OUT/autotune.jsonl | cut -c1-300)" | tee -a $OUT/summary.txt
timeout 300 python scripts/attn_bench.py > $OUT/attn_bench.json 2> $OUT/attn_bench.err; echo "   attn_bench rc=$?" | tee -a $OUT/summary.txt
timeout 600 python scripts/encoder_ops_bench.py > $OUT/encoder_ops_bench.json 2> $OUT/encoder_ops_bench.err; echo "   encoder_ops_bench rc=$?" | tee -a $OUT/summary.txt

echo "== 3. per-kernel times with every switch on (rocprofv3 --kernel-trace --stats)" | tee -a $OUT/summary.txt
( cd /tmp && LEANN_MI355X_ATTN=2 LEANN_MI355X_LN=2 LEANN_MI355X_POOL=1 LEANN_MI355X_EMBED=1 LEANN_MI355X_MLP=1 LEANN_MI355X_MLP_VARIANT=2 LEANN_MI355X_LINEAR=1 LEANN_MI355X_PACK=1 \
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_next -- python $OLDPWD/scripts/encoder_bench.py > $OLDPWD/$OUT/encoder_bench_all_on.json 2> $OLDPWD/$OUT/rocprof.err )
find /tmp/prof_next -name "*kernel_stats.csv" -exec cp {} $OUT/encoder_all_on_kernel_stats.csv \; 2>/dev/null
echo "   rocprof rc=$?" | tee -a $OUT/summary.txt

echo "== 4. bench with the switches that passed (edit the list below after reading summary.txt if something failed)" | tee -a $OUT/summary.txt
if ! grep -q "rc=[1-9]" $OUT/summary.txt; then
    LEANN_MI355X_ATTN=2 LEANN_MI355X_LN=2 LEANN_MI355X_POOL=1 LEANN_MI355X_EMBED=1 LEANN_MI355X_MLP=1 LEANN_MI355X_MLP_VARIANT=2 LEANN_MI355X_LINEAR=1 LEANN_MI355X_PACK=1 \
      timeout 1500 python bench.py --no-cpu-baseline > $OUT/bench_all_on.json 2> $OUT/bench_all_on.err
    echo "   bench rc=$? $(python -c "import json;d=json.load(open('$OUT/bench_all_on.json'));print(d['value'],d['recall_at_10'],d['roofline_encoder'])" 2>/dev/null)" | tee -a $OUT/summary.txt
else
    echo "   skipped: a step above failed" | tee -a $OUT/summary.txt
fi
cat $OUT/summary.txt
