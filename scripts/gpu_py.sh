#!/bin/bash
# generic python session: health probe, then the given command with a timeout.  usage: gpu_py.sh <tag> <timeout_s> <cmd...>
set -u
cd "$(dirname "$0")/.."
TAG=$1; TO=$2; shift 2
OUT=gpurun_out/$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 5 60 leann_amd/lib/bin/kbench 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
timeout -k 10 $TO "$@" > $OUT/out.log 2> $OUT/err.log; echo "rc=$?"; tail -c 6000 $OUT/out.log; echo "--- stderr tail"; tail -5 $OUT/err.log
