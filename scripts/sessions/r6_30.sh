#!/bin/bash
# Round 6, GPU session 30: session 29's question with the chip kept full: two / three forwards in flight on their own streams (one provider handle, stream and host
# thread each), so that a small forward's partly filled rounds of workgroups run beside another forward's kernels.  If small (Infinity-Cache-resident) forwards buy
# clocks at the power cap, this is where it would show as time.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6s30; rm -rf "$OUT"; mkdir -p "$OUT"
T0=$(date +%s)
timeout -k 10 700 python scripts/forward_size_ab.py --budgets 1048512,131072,65536,49152 --streams 1,2,3 > $OUT/forward_size_streams_ab.jsonl 2> $OUT/forward_size_streams_ab.err; echo "ab rc=$? in $(( $(date +%s) - T0 )) s"
tail -3 $OUT/forward_size_streams_ab.err
python - <<'PY'
import json
for l in open("gpurun_out/r6s30/forward_size_streams_ab.jsonl"):
    r = json.loads(l)
    if "round" in r:
        print(r["round"], r["forward_tokens_budget"], r["streams"], r["ms_per_call"], r["TFLOPs"], (r.get("sclk_mhz") or {}).get("median"), (r.get("power_w") or {}).get("median"))
    else:
        print(l.strip())
PY
