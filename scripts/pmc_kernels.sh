#!/bin/bash
# HBM traffic of the hand-written kernels from PMC counters (own pass, kernel-trace only -- see the
# gfx950 notes in /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE reads 1/2 of a wide coalesced stream).
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/pmc_$1; shift
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT" -o pmc -- python scripts/kernel_bench.py "$@" > "$OUT/run.log" 2>&1
tail -2 "$OUT/run.log"
python - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
print(f)
if f:
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f[0])):
        if r.get("Counter_Name") == "FETCH_SIZE":
            k = r["Kernel_Name"][:60]
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]
    with open(sys.argv[1] + "/fetch_summary.csv", "w") as o:
        o.write("kernel,dispatches,FETCH_SIZE_total_KB,FETCH_SIZE_per_dispatch_KB\n")
        for k, (n, v) in rows:
            o.write(f'"{k}",{n},{v:.1f},{v/n:.1f}\n')
            print(f"{k:60s} n={n:6d} FETCH_SIZE/dispatch={v/n:12.1f} KB")
    import os
    for g in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True) + glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
        if os.path.getsize(g) > 4 << 20:
            os.remove(g)
PY
