#!/bin/bash
# Round 4, GPU session 2: generation 4 of the layer tail as the product default -- pytest -m gpu, the driver's bench command under
# rocprofv3 --kernel-trace --stats (library-side provider in the timed region, library-side event pairs), SQ counters + clock of the
# new kernel next to generation 3, HBM-side traffic (FETCH_SIZE / WRITE_SIZE passes), a fresh PMC pass of k_update.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r4s2; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
timeout -k 5 60 $KB 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
timeout -k 10 420 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$? $(tail -1 $OUT/pytest_gpu.log)"; grep -E "^(FAILED|ERROR)|Error|assert " $OUT/pytest_gpu.log | head -20
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 10 560 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o c2 -- python bench.py --gpus 1 --steps 4 --warmup 1 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "bench rc=$?"
find $OUT/prof -name "*kernel_trace.csv" -size +8M -delete; find $OUT/prof -name "*.db" -delete
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r4s2/bench_c2.json").read().strip().splitlines()[-1])
    keep = ("value", "recall_at_10", "ms_per_step", "roofline", "roofline_encoder", "without_call_memo", "encoder_kernels_profiled_step", "value_by_batch", "small_batch_latency", "full_step_over_the_python_provider", "extras_errors", "parity_check", "cpu_baseline")
    print(json.dumps({k: d.get(k) for k in keep})[:6000])
except Exception as ex:
    print("bench json:", ex)
PY
tail -4 $OUT/bench_c2.err | cut -c1-300
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-200
KBENCH_TAIL4_ONLY=1 bash scripts/pmc_sq.sh r4s2/pmc_sq tail4 2>&1 | tail -45 | cut -c1-220
bash scripts/pmc_tail.sh r4s2/pmc_tail 2>&1 | tail -40 | cut -c1-200
bash scripts/pmc_kernels.sh r4_k_update --provider --batch 2048 --beam 1 --deg 10 --reps 1 2>&1 | tail -12 | cut -c1-200
