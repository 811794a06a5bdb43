#!/bin/bash
# Round 6, GPU session 12: B = 1 latency (200k index) against the two size limits of the hidden-384 forward -- LEANN_MI355X_SMALL_TOKENS (general kernels below it)
# x LEANN_MI355X_QKV_GEMM_TOKENS (QKV from the general GEMM below it; 0 = round 5's large form) -- at batch_size 0, 64, 128; same queries in every variant.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6s12; rm -rf "$OUT"; mkdir -p "$OUT"
V=""
for sm in 4096 8192 16384; do for qg in 0 45056; do V="$V,LEANN_MI355X_SMALL_TOKENS=$sm+LEANN_MI355X_QKV_GEMM_TOKENS=$qg"; done; done
V=${V#,}
for bs in 0 64 128; do
  LAT_VARIANTS_ONLY=1 LAT_BATCHES=1,16 LAT_BATCH_SIZE=$bs LAT_VARIANTS="$V" timeout -k 10 300 python scripts/latency_bench.py 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
for v in r['small_forward_variants']:
    print(json.dumps({'batch_size': $bs, 'variant': v['variant'], 'B1_p50_ms': v['B1']['p50_ms'], 'B1_rounds': v['B1']['rounds_last_call'], 'B16_p50_ms': v['B16']['p50_ms'], 'same_labels': [v['B1']['calls_with_the_first_variants_labels'], v['B16']['calls_with_the_first_variants_labels']]}))" | tee -a $OUT/latency_b1_forward_size_limits.jsonl
done
