#!/bin/bash
# rocprofv3 kernel-trace + stats of a bench.py run; keeps only the (small) stats CSVs.
#   scripts/prof_bench.sh <tag> <bench args...>
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o "$TAG" -- python bench.py "$@" > "$OUT/bench.log" 2>&1
tail -3 "$OUT/bench.log"
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
find "$OUT" -name "*.db" -delete
python - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
print(f)
if f:
    rows = list(csv.DictReader(open(f[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("total kernel ms", tot / 1e6)
    for r in rows[:45]:
        print(f'{r["Name"][:120]:120s} calls={r["Calls"]:>7s} total_ms={float(r["TotalDurationNs"])/1e6:10.2f} avg_us={float(r["AverageNs"])/1e3:9.2f} pct={r["Percentage"]}')
PY
