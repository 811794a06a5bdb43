#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/s11; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 5 60 leann_amd/lib/bin/kbench 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
echo "== pytest -m gpu"
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "rc=$? $(tail -1 $OUT/pytest_gpu.log)"; tail -15 $OUT/pytest_gpu.log | head -14
echo "== bench (2 steps)"
timeout -k 10 900 python bench.py --steps 2 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?"; tail -c 7000 $OUT/bench.json; tail -8 $OUT/bench.err
