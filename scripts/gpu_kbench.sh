#!/bin/bash
# quick kernel session: health probe + kbench of the named groups.  usage: gpu_kbench.sh <tag> <what...>
set -u
cd "$(dirname "$0")/.."
TAG=$1; shift
OUT=gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
timeout -k 5 60 $KB 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
for w in "$@"; do
  timeout -k 5 200 $KB 262107 20 $w > $OUT/kbench_$w.jsonl 2> $OUT/kbench_$w.err; echo "== $w rc=$?"; cat $OUT/kbench_$w.jsonl; tail -2 $OUT/kbench_$w.err
done
