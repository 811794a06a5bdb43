#!/bin/bash
# What to run FIRST in the next round's first GPU session (~14 GPU-minutes), in this order.
#   1. `pytest -m gpu` on the tree as it is (the tests marked NOT_YET_ON_HARDWARE / xfail(strict=False) are the ones of 3b / 3c: XPASS
#      = remove the marker), then the driver's bench command under rocprofv3 (reference line + per-kernel table).
#   2. Is this box in the layer tail's slow mode?  `kbench tail4` on the product form next to the six-stage / non-temporal
#      builds; on a SLOW box additionally the TCC hit / miss counters of the same command (one --pmc pass, --kernel-trace only): the
#      first experiment towards knowing what the slow boxes have in common.
#   3. B = 1 latency with the speculative prefetch swept (S = 0, 2, 4, 8, 16, 32) on the 200k-chunk index of scripts/latency_bench.py.
#   3b. lm_rowgemm_ln_h384_f16 (written after round 4's GPU budget was spent: emulation-validated, never run on hardware; off by default):
#      its GPU tests are part of step 1; here `kbench <tokens> 50 rowln` at the token counts of small rounds and the latency rows with
#      LEANN_MI355X_SMALL_ROWLN=1 against the default.  If it wins, make it the default of the small-forward form.
#   3b'. lm_small_layer_h384_f16 (same status): the REST of a small-forward layer + the next layer's QKV projection in one launch (2 launches per
#      layer instead of 7): `kbench <tokens> 50 slayer`, latency rows with LEANN_MI355X_SMALL_LAYER=1.  The expected winner of the three.
#   3c. option single_query_direct (also written after the budget was spent; off by default): LAT_DIRECT=1 rows next to the default.
#   4. C5 at its stated size WITH the extras (memo-off steps, parity block, CPU baseline): needs ~25 GPU-minutes (set-up alone 13),
#      so only if the round's budget allows: `python bench.py --config c5 --steps 2 --warmup 1 --no-latency-rows --no-min-ef-step
#      --no-table-roofline --no-provider-ab --cpu-baseline-seconds 10` under `timeout 1700`.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/next1; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
[ -x $KB ] || bash scripts/build_kbench.sh
timeout -k 10 300 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$? $(tail -1 $OUT/pytest_gpu.log)"; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log | head
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 10 540 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o c2 -- python bench.py --gpus 1 --steps 6 --warmup 2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "bench rc=$?"
find $OUT/prof -name "*kernel_trace.csv" -size +8M -delete; find $OUT/prof -name "*.db" -delete
for v in diag diag_W6 diag_NT; do
  [ -f leann_amd/lib/$v/libleann_mi355x.so ] || continue
  LD_LIBRARY_PATH=$PWD/leann_amd/lib/$v KBENCH_TAIL4_ONLY=1 KBENCH_TAIL4_STAMP=1 timeout -k 5 120 $KB 262107 10 tail4 > $OUT/tail_$v.jsonl 2> $OUT/tail_$v.err
  echo "tail $v: gen4 $(grep '"variant": "0", "round": 2' $OUT/tail_$v.jsonl | grep -o '"us": [0-9.]*') gen3 $(grep 'generation 3)", "round": 2' $OUT/tail_$v.jsonl | grep -o '"us": [0-9.]*')"
done
for tk in 1500 6000 12000; do timeout -k 5 60 $KB $tk 50 rowlnslayer > $OUT/kbench_small_$tk.jsonl 2>&1; grep '"round": 2' $OUT/kbench_small_$tk.jsonl | cut -c1-300; done
LAT_BATCHES=1,4,16 timeout -k 10 120 python scripts/latency_bench.py > $OUT/latency_default.json 2> $OUT/latency_default.err; echo "latency default rc=$?"; cut -c1-900 $OUT/latency_default.json
LAT_BATCHES=1 LAT_DIRECT=1 timeout -k 10 120 python scripts/latency_bench.py > $OUT/latency_direct.json 2> $OUT/latency_direct.err; echo "latency with single_query_direct rc=$?"; cut -c1-600 $OUT/latency_direct.json
LAT_BATCHES=1,4,16 LAT_DIRECT=1 LEANN_MI355X_SMALL_ROWLN=1 timeout -k 10 120 python scripts/latency_bench.py > $OUT/latency_direct_rowln.json 2> $OUT/latency_direct_rowln.err; echo "latency with both rc=$?"; cut -c1-900 $OUT/latency_direct_rowln.json
LAT_BATCHES=1,4,16 LAT_DIRECT=1 LEANN_MI355X_SMALL_LAYER=1 timeout -k 10 120 python scripts/latency_bench.py > $OUT/latency_direct_slayer.json 2> $OUT/latency_direct_slayer.err; echo "latency with single_query_direct + the small-layer kernel rc=$?"; cut -c1-900 $OUT/latency_direct_slayer.json
LAT_BATCHES=1,4,16 LEANN_MI355X_SMALL_ROWLN=1 timeout -k 10 120 python scripts/latency_bench.py > $OUT/latency_rowln.json 2> $OUT/latency_rowln.err; echo "latency with the row-complete GEMM + LayerNorm rc=$?"; cut -c1-900 $OUT/latency_rowln.json
timeout -k 10 240 python scripts/latency_bench.py --speculate 0,2,4,8,16,32 > $OUT/latency_speculate.json 2> $OUT/latency_speculate.err; echo "latency rc=$?"; cut -c1-1500 $OUT/latency_speculate.json
