#!/bin/bash
# Round 2, GPU session 20: the whole -m gpu suite, then the driver's bench command under rocprofv3 --kernel-trace --stats
# (one run gives the JSON line and the per-kernel table it has to agree with).
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/s20; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 5 60 leann_amd/lib/bin/kbench 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
echo "== pytest -m gpu"
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "rc=$? $(tail -1 $OUT/pytest_gpu.log)"; tail -15 $OUT/pytest_gpu.log | head -14
echo "== rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1"
( cd /tmp && timeout -k 10 1000 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python $ROOT/bench.py --steps 2 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err ); echo "rc=$?"
tail -c 9000 $OUT/bench.json; tail -5 $OUT/bench.err
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $OUT/bench_kernel_stats.csv && head -25 $OUT/bench_kernel_stats.csv | cut -c1-200
