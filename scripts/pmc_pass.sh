#!/bin/bash
# ONE rocprofv3 --pmc pass (counters of the caller's choice, --kernel-trace only: no other trace domain) over a kbench mode, per-kernel means
# written to gpurun_out/<tag>/pmc_<pass>.json.   scripts/pmc_pass.sh <tag> <pass-name> <kbench mode> <tokens> -- COUNTER [COUNTER ...]
# MI355X_MICROARCH.md: SQ_WAIT_* / SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES cycles.
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1; PASS=$2; MODE=$3; TOK=$4; shift 5
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
rm -rf /tmp/pmc_${TAG}_$PASS
timeout -k 5 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$PASS -o pmc -- $ROOT/leann_amd/lib/bin/kbench $TOK 3 $MODE > $OUT/pmc_$PASS.log 2>&1
echo "pmc pass $PASS rc=$?"; tail -2 $OUT/pmc_$PASS.log | cut -c1-200
python - "$OUT" "/tmp/pmc_${TAG}_$PASS" "$PASS" <<'PY'
import csv, glob, json, sys, collections
out_dir, src, name = sys.argv[1:4]
f = glob.glob(src + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
if f:
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0][-72:]
        a = agg[k][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
res = {}
for k, cs in agg.items():
    if k.startswith("ref_") or "rand" in k:
        continue
    d = {c: round(v / n) for c, (n, v) in cs.items()}
    d["dispatches"] = max(n for n, _ in cs.values())
    res[k] = d
json.dump(res, open(f"{out_dir}/pmc_{name}.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
