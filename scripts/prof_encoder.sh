#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/prof_enc; rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o enc -- python - <<'PY' > "$OUT/run.log" 2>&1
import sys, torch
sys.path.insert(0, '.')
from leann_amd.encoder import BertEncoder, config_for
from leann_amd.synth import CorpusSpec, SyntheticCorpus, pad_batch
dev = torch.device("cuda")
enc = BertEncoder.random_init(config_for("all-MiniLM-L6-v2"), 0).to(dev, dtype=torch.float16)
c = SyntheticCorpus(CorpusSpec(n_chunks=8192))
ids, lens = pad_batch(*c.chunks(), 256)
ti, tl = torch.from_numpy(ids).to(dev), torch.from_numpy(lens).to(dev)
for _ in range(4):
    enc.encode_tokens_packed(ti, tl, 524288)
torch.cuda.synchronize()
PY
find "$OUT" -name "*kernel_trace.csv" -delete
python - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6)
for r in rows[:22]:
    print(f'{r["Name"][:100]:100s} calls={r["Calls"]:>6s} total_ms={float(r["TotalDurationNs"])/1e6:9.2f} avg_us={float(r["AverageNs"])/1e3:9.2f} pct={r["Percentage"]}')
PY
