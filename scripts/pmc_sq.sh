#!/bin/bash
# SQ counters of the MFMA-bound kernels ("there is no SQ PMC pass of k_attn_out_mlp_h384"): one --pmc pass
# (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES, SQ_WAIT_INST_ANY, SQ_WAIT_ANY, SQ_ACTIVE_INST_ANY, SQ_INSTS_VALU_MFMA_MOPS...)
# over `kbench 262107 3 <mode>`; --kernel-trace only (no other trace domain).  MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles
# (32 per 32x32x16 MFMA), the SQ_WAIT_* / SQ_WAVE_CYCLES counters count quad-cycles.
#   scripts/pmc_sq.sh <tag> <kbench mode> [kernel-name substring ...]
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1; MODE=$2
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
rm -rf /tmp/pmc_sq_$TAG
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
  -d /tmp/pmc_sq_$TAG -o pmc -- $ROOT/leann_amd/lib/bin/kbench 262107 3 $MODE > $OUT/pmc_sq.log 2>&1
echo "pmc rc=$?"; tail -2 $OUT/pmc_sq.log | cut -c1-200
python - "$OUT" "/tmp/pmc_sq_$TAG" <<'PY'
import csv, glob, json, sys, collections
out_dir, src = sys.argv[1], sys.argv[2]
f = glob.glob(src + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
if f:
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0][-64:]
        a = agg[k][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
res = {}
for k, cs in agg.items():
    if not any(s in k for s in ("k_attn_out_mlp", "k_layer_tail", "k_gemm_f16", "k_gemm_ws", "k_attn_varlen")):
        continue
    d = {c: round(v / n) for c, (n, v) in cs.items()}
    d["dispatches"] = max(n for n, _ in cs.values())
    if d.get("SQ_BUSY_CYCLES") and d.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        # SQ_BUSY_CYCLES is summed over the chip's shader engines / XCDs as the counter defines it; the per-kernel ratios below are the readable part
        d["mfma_busy_per_wave_cycle"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / max(4.0 * d.get("SQ_WAVE_CYCLES", 0), 1), 4)
    if d.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in d:
                d[c + "_share_of_wave_cycles"] = round(d[c] / d["SQ_WAVE_CYCLES"], 4)
    res[k] = d
json.dump(res, open(out_dir + "/pmc_sq.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:4000])
PY
