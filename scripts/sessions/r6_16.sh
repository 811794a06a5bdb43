#!/bin/bash
# Round 6, GPU session 16: which kernels does the vendor library run on the encoder's GEMM shapes (names carry the macro tile, the MFMA shape, the LDS scheme)?
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6s16; rm -rf "$OUT"; mkdir -p "$OUT"
KB=$PWD/leann_amd/lib/bin/kbench
export TMPDIR=/tmp
( cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof16 -o g -- $KB 65536 5 gemmf16 > /tmp/kb16.log 2>&1 )
grep -v '"kbench"' /tmp/kb16.log | cut -c1-220 | tail -30
find /tmp/prof16 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/gemm_kernel_stats.csv
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r6s16/gemm_kernel_stats.csv')
for r in csv.DictReader(open(f[0])):
    print(r['Name'][:330], r['Calls'], r['AverageNs'])
PY
