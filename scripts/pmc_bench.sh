#!/bin/bash
# HBM traffic (FETCH_SIZE) of the timed-region kernels of bench.py, own PMC pass (kernel-trace only).
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/pmc_bench; rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT" -o pmc -- python bench.py "$@" > "$OUT/run.log" 2>&1
tail -1 "$OUT/run.log" | cut -c1-400
python - "$OUT" <<'PY'
import csv, glob, sys, collections, os
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f[0])):
    if r.get("Counter_Name") == "FETCH_SIZE" and ("lm::" in r["Kernel_Name"]):
        k = r["Kernel_Name"][:70]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
with open(sys.argv[1] + "/fetch_summary_lm_kernels.csv", "w") as o:
    o.write("kernel,dispatches,FETCH_SIZE_total_KB,FETCH_SIZE_per_dispatch_KB\n")
    for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        o.write(f'"{k}",{n},{v:.1f},{v/n:.2f}\n')
        print(f"{k:70s} n={n:7d} FETCH/dispatch={v/n:12.2f} KB")
for g in glob.glob(sys.argv[1] + "/**/*.csv", recursive=True):
    if os.path.getsize(g) > 4 << 20:
        os.remove(g)
PY
