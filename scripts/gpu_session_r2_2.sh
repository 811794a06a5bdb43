#!/bin/bash
# Round 2, GPU session 2: health probe (bail out at once on a faulting box), C++ kernel micro-benchmarks (encoder kernels incl.
# the second-generation linear kernel, against fp32 references), then the -m gpu suite.  Output: gpurun_out/s2/.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/s2
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
KB=leann_amd/lib/bin/kbench
echo "== 0. health probe" | tee $OUT/summary.txt
timeout -k 5 120 $KB 4096 2 ln > $OUT/probe.log 2>&1
rc=$?; echo "   rc=$rc $(tail -1 $OUT/probe.log)" | tee -a $OUT/summary.txt
if [ $rc -ne 0 ]; then echo "   BOX UNHEALTHY -- stopping" | tee -a $OUT/summary.txt; cat $OUT/probe.log; exit 0; fi
echo "== 1. kbench" | tee -a $OUT/summary.txt
for w in linear mlp attn ln; do
  timeout -k 5 240 $KB 262107 20 $w > $OUT/kbench_$w.jsonl 2> $OUT/kbench_$w.err
  echo "   $w rc=$?" | tee -a $OUT/summary.txt; cat $OUT/kbench_$w.jsonl | tee -a $OUT/summary.txt; tail -2 $OUT/kbench_$w.err
done
echo "== 2. pytest -m gpu" | tee -a $OUT/summary.txt
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1
echo "   rc=$? $(tail -1 $OUT/pytest_gpu.log)" | tee -a $OUT/summary.txt
tail -25 $OUT/pytest_gpu.log
