#!/bin/bash
# Round 5, GPU session 3 (~17 GPU-minutes): instruction costs (vopbench), then the evidence runs VERDICT r4 asked for:
#   C4: ONE shard at its stated per-rank size (7.5M chunks, B = 256) with exchange + merge in the timed region and the shard-level parity block;
#   C3: 10M chunks with `roofline` = the encoder's dominant kernel, the traversal as roofline_traversal, parity on the run's own index.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r5s3; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 5 120 leann_amd/lib/bin/vopbench > $OUT/vopbench.jsonl 2> $OUT/vopbench.err; echo "vopbench rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/r5s3/vopbench.jsonl"):
    if not l.startswith("{"): continue
    r = json.loads(l)
    if "op" in r:
        print(f'{r["op"]:44s} w/simd {r["waves_per_simd"]}  per wave {r["ticks_per_instr_per_wave"]:7.2f}  per SIMD {r["simd_ticks_per_instr"]:7.2f}')
    else:
        print(json.dumps(r)[:1200])
PY
timeout -k 10 400 python -m pytest tests/test_gpu_bench_scripts.py tests/test_gpu_pq.py -m gpu -q -x > $OUT/pytest_bench_scripts.log 2>&1; rc=$?; echo "pytest bench scripts + pq rc=$rc $(tail -1 $OUT/pytest_bench_scripts.log)"
if [ $rc -ne 0 ]; then tail -40 $OUT/pytest_bench_scripts.log; exit 0; fi
timeout -k 10 560 python scripts/bench_c4.py --chunks 7500000 --batch 256 --steps 3 --warmup 1 --cpu-baseline-seconds 15 > $OUT/bench_c4_one_shard_7p5M.json 2> $OUT/bench_c4_one_shard_7p5M.log; echo "c4 rc=$?"; tail -3 $OUT/bench_c4_one_shard_7p5M.log | cut -c1-300; cut -c1-1500 $OUT/bench_c4_one_shard_7p5M.json
timeout -k 10 900 python scripts/bench_c3.py --chunks 10000000 --steps 3 --warmup 1 --M 32 --efc 200 --rerank-expanded 0 --cpu-baseline-queries 3 > $OUT/bench_c3_10M.json 2> $OUT/bench_c3_10M.log; echo "c3 rc=$?"; tail -6 $OUT/bench_c3_10M.log | cut -c1-400; cut -c1-2500 $OUT/bench_c3_10M.json
