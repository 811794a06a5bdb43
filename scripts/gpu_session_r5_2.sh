#!/bin/bash
# Round 5, GPU session 2 (~1 GPU-minute): instruction costs behind the attention kernel's softmax (scripts/vopbench.cpp).
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r5s2; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 5 120 leann_amd/lib/bin/vopbench > $OUT/vopbench.jsonl 2> $OUT/vopbench.err; echo "rc=$?"
python - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/r5s2/vopbench.jsonl") if l.startswith("{")]
for r in rows:
    if "op" in r:
        print(f'{r["op"]:48s} w/simd {r["waves_per_simd"]}  per wave {r["ticks_per_instr_per_wave"]:7.2f}  per SIMD {r["simd_ticks_per_instr"]:7.2f}')
    else:
        print(json.dumps(r)[:1500])
PY
