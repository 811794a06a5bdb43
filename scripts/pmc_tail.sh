#!/bin/bash
# HBM-side traffic of the fused layer-tail kernel (or, with a second argument such as `bwqkv`, of another kbench mode's kernels) from the
# TCC counters, in two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over `kbench 262107 3 <bw + mode>`, calibrated in the same passes on kbench's streaming read / write probes of a known byte count
# (MI355X_MICROARCH.md, HBM: FETCH_SIZE reports 1/2 of a wide coalesced read on gfx950; WRITE_SIZE uncalibrated).  --kernel-trace only.
set -u
export KBENCH_TAIL4_ONLY=1
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/$1; mkdir -p "$OUT"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o pmc -- $ROOT/leann_amd/lib/bin/kbench 262107 3 ${2:-bwtail4} > $OUT/pmc_$c.log 2>&1
  echo "$c rc=$?"
done
python - "$OUT" "${2:-bwtail4}" <<'PY'
import csv, glob, json, sys, collections
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/pmc_{c}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: [0, 0.0])
    if f:
        for r in csv.DictReader(open(f[0])):
            if r.get("Counter_Name") == c:
                k = r["Kernel_Name"].split("(")[0][-48:]
                agg[k][0] += 1
                agg[k][1] += float(r["Counter_Value"])
    out[c] = {k: {"dispatches": n, "per_dispatch_KB": round(v / n, 1)} for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]}
json.dump(out, open(sys.argv[1] + "/pmc_" + (sys.argv[2] if len(sys.argv) > 2 else "bwtail4") + ".json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3500])
PY
