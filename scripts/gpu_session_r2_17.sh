#!/bin/bash
# Round 2, GPU session 17 (fused MLP v3 only): PMC counters of the MFMA kernels (own passes, --pmc only + kernel trace), on the C++ harness.
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/s17
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
KB=$ROOT/leann_amd/lib/bin/kbench
timeout -k 5 60 $KB 4096 2 ln > $OUT/probe.log 2>&1 || { echo "BOX UNHEALTHY"; cat $OUT/probe.log; exit 0; }
pass() {  # tag what counters...
  local tag=$1 what=$2; shift 2
  ( cd /tmp && timeout -k 5 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$tag -o $tag -- $KB 131072 2 $what > $OUT/pmc_$tag.log 2>&1 )
  f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" $tag >> $OUT/pmc_summary.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:70]
    if "ref_" in k or "Cijk" in k or "rocblas" in k.lower(): continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    print(sys.argv[2], k, {c: round(v / cnt[(k, c)]) for c, v in d.items()}, flush=True)
PY
}
for what in mlp; do
  pass ${what}_a $what SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
  pass ${what}_b $what SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM
  pass ${what}_c $what GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum
done
cat $OUT/pmc_summary.txt
