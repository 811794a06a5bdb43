#!/bin/bash
# FETCH_SIZE pass (own pass, kernel trace only) over scripts/pmc_table_mode.py; joins the counter rows of k_dist_pairs (calibration) and
# k_search_table (the four searches, in call order) with the script's own line -> gpurun_out/<tag>/pmc_table_mode.json
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r6}
NQ=${2:-8192}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
rm -rf /tmp/pmc_tm_$TAG
( cd $ROOT && timeout -k 5 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_tm_$TAG -o pmc -- python scripts/pmc_table_mode.py 1000000 $NQ > $OUT/pmc_table_mode_run.json 2> $OUT/pmc_table_mode_run.err )
echo "pmc table mode rc=$?"; tail -2 $OUT/pmc_table_mode_run.err | cut -c1-300
python - "$OUT" /tmp/pmc_tm_$TAG <<'PY'
import csv, glob, json, sys
out_dir, src = sys.argv[1:3]
run = None
for ln in open(out_dir + "/pmc_table_mode_run.json"):
    if ln.startswith("{"):
        run = json.loads(ln)
f = glob.glob(src + "/**/*counter_collection.csv", recursive=True)
cal, srch = [], []
if f:
    for r in csv.DictReader(open(f[0])):
        if r.get("Counter_Name") != "FETCH_SIZE":
            continue
        k = r["Kernel_Name"]
        if "k_dist_pairs" in k:
            cal.append(float(r["Counter_Value"]))
        elif "k_search_table" in k:
            srch.append((k.split("(")[0][-60:], float(r["Counter_Value"])))
res = {"run": run, "FETCH_SIZE_KB_calibration_launch": cal, "FETCH_SIZE_KB_search_launches": srch}
if run and cal and len(srch) >= len(run["calls"]):
    srch = srch[-len(run["calls"]):]  # (the graph builder searches with the same kernel: the script's own four calls are the LAST launches of the pass)
    res["FETCH_SIZE_KB_search_launches"] = srch
    factor = run["calibration"]["known_bytes"] / (cal[0] * 1024.0)
    res["factor_known_bytes_over_FETCH_SIZE_in_this_access_pattern"] = round(factor, 4)
    for c, (kn, kb) in zip(run["calls"], srch):
        c["kernel"] = kn
        c["FETCH_SIZE_KB"] = round(kb, 1)
        c["l2_miss_side_bytes"] = round(kb * 1024 * factor)
        c["l2_miss_side_over_algorithmic"] = round(c["l2_miss_side_bytes"] / c["algorithmic_bytes"], 4)
        c["l2_miss_side_GBps"] = round(c["l2_miss_side_bytes"] / (c["ms"] * 1e-3) / 1e9, 1)
        c["frac_of_8TBps_algorithmic"] = round(c["algorithmic_GBps"] / 8000.0, 4)
        c["frac_of_8TBps_l2_miss_side"] = round(c["l2_miss_side_GBps"] / 8000.0, 4)
json.dump(res, open(out_dir + "/pmc_table_mode.json", "w"), indent=1)
print(json.dumps(res)[:2500])
PY
