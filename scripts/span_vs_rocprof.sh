#!/bin/bash
# checks that the in-kernel span timing of k_update agrees with rocprofv3's kernel duration
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/span_check; rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o s -- python scripts/kernel_bench.py "$@" > "$OUT/run.log" 2>&1
tail -1 "$OUT/run.log" | python -c "import sys,json; d=json.loads(sys.stdin.read())['search']; print('bench: span us/launch', 1e3*d['update_span_ms']/d['launches'], 'event-pair us/launch', 1e3*d['update_ms']/d['launches'], 'launches', d['launches'])"
grep "k_update" "$OUT"/*kernel_stats.csv | cut -d, -f1-4
find "$OUT" -name "*kernel_trace.csv" -delete
