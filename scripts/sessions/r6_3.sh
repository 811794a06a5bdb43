#!/bin/bash
# Round 6, GPU session 3: (a) what each ingredient of the fused QKV + attention kernel costs (diagnosis build: tile loop / slab barriers / DMA / fragment
# reads ablated, timing only) at three length profiles; (b) the stored-embedding search: queries in flight x register allocation (4 vs 5 waves per SIMD).
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6s3; rm -rf "$OUT"; mkdir -p "$OUT"
KB=leann_amd/lib/bin/kbench
for fl in "" 256 128; do
  if [ -n "$fl" ]; then export KBENCH_FIXED_LEN=$fl; else unset KBENCH_FIXED_LEN; fi
  KBENCH_QA_ABLATIONS=1 timeout -k 5 120 $KB 262107 10 fusedqa 2>&1 | grep -E "ablation|round\": 2" | sed "s/^/{\"lengths\": \"${fl:-N(180,50)}\", \"row\": /; s/$/}/" | tee -a $OUT/kbench_fusedqa_ablations.jsonl | cut -c1-260
done
unset KBENCH_FIXED_LEN
timeout -k 10 400 python scripts/table_mode_bench.py --occupancy-sweep > $OUT/table_mode_occupancy_sweep.json 2> $OUT/table_mode_occupancy_sweep.err; echo "table sweep rc=$?"
python - <<'PY'
import json, collections
try:
    r = json.load(open("gpurun_out/r6s3/table_mode_occupancy_sweep.json"))
    acc = collections.defaultdict(list)
    for c in r["occupancy_sweep"]:
        acc[(c["queries"], c["beam"], c["form"], str(c["waves_per_simd_target"]))].append((c["GBps"], c["identical_labels"]))
    for k, v in sorted(acc.items()):
        print(k, v)
except Exception as e:
    print("no sweep:", e)
PY
tail -2 $OUT/table_mode_occupancy_sweep.err | cut -c1-300
