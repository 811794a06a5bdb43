#!/bin/bash
# Round 4, GPU session 9 (final validation): the GPU suite on the final tree, smoke(), and the driver's bench command under
# rocprofv3 --kernel-trace --stats (JSON line + per-kernel table of ONE run).
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r4s9; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
timeout -k 5 60 $KB 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
KBENCH_TAIL4_ONLY=1 timeout -k 5 120 $KB 262107 10 tail4 > $OUT/box_probe_tail4.jsonl 2>/dev/null; echo "box probe: gen4 $(grep '"variant": "0", "round": 2' $OUT/box_probe_tail4.jsonl | grep -o '"us": [0-9.]*') gen3 $(grep 'generation 3)", "round": 2' $OUT/box_probe_tail4.jsonl | grep -o '"us": [0-9.]*')"
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout -k 10 540 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o c2 -- python bench.py --gpus 1 --steps 6 --warmup 2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "bench rc=$?"
find $OUT/prof -name "*kernel_trace.csv" -size +8M -delete; find $OUT/prof -name "*.db" -delete
tail -4 $OUT/bench_c2.err | cut -c1-300
cut -c1-1800 $OUT/bench_c2.json
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -r head -12 | cut -c1-200
timeout -k 10 280 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$? $(tail -1 $OUT/pytest_gpu.log)"; grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log | head -20
timeout -k 10 60 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $OUT/smoke.log | cut -c1-200)"
