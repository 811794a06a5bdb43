#!/bin/bash
# Round 6, GPU session 21: a hop's new keys placed by counting instead of a bitonic sort in k_search_table / k_update (rank_merge_unsorted): the parity suites, then the
# stored-embedding bench (1M index, 8192 / 16384 / 32768 distinct queries in flight) on the new library and on the library of the commit before, same box.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6s21; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_c2.py tests/test_gpu_pipeline.py -m gpu -q > $OUT/pytest_parity.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/pytest_parity.log)"; grep -E "^(FAILED|ERROR)" $OUT/pytest_parity.log | head | cut -c1-250
run() {
  timeout -k 10 400 python scripts/table_mode_bench.py --queries-in-flight > $OUT/table_mode_$1.json 2> $OUT/table_mode_$1.err; echo "table $1 rc=$?"
  python - "$1" <<'PY'
import json, sys
try:
    r = json.load(open(f"gpurun_out/r6s21/table_mode_{sys.argv[1]}.json"))
    for k in ("persistent_wave_per_query_beam1", "persistent_wave_per_query_beam4", "persistent_workgroup_per_query_beam1", "persistent_workgroup_per_query_beam4", "lockstep_beam1", "lockstep_beam4"):
        print(sys.argv[1], k, [(x["GBps"], x["ms"], x["identical_results"]) for x in r[k]])
    for x in r.get("occupancy_sweep", []):
        if x["round"] == 1: print(sys.argv[1], x["queries"], x["form"], "beam", x["beam"], x["GBps"], x["ms"], x["identical_labels"])
except Exception as e:
    print("no json:", e)
PY
}
run new
cp leann_amd/lib/libleann_mi355x.so /tmp/new.so && cp leann_amd/lib_before/libleann_mi355x.so leann_amd/lib/libleann_mi355x.so && touch leann_amd/lib/libleann_mi355x.so
run before
cp /tmp/new.so leann_amd/lib/libleann_mi355x.so
