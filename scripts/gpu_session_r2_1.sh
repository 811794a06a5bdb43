#!/bin/bash
# Round 2, GPU session 1: (1) the whole -m gpu suite (now includes the encoder-kernel tests), (2) per-kernel times of the
# DEFAULT encoder path and of the path with the hidden-384 linear kernel on (rocprofv3 --kernel-trace --stats),
# (3) stored-embedding search micro-benchmark at beam 1 / beam 4.  Everything lands under gpurun_out/s1/.
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/s1
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== 1. pytest -m gpu" | tee $OUT/summary.txt
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1
echo "   rc=$? $(tail -1 $OUT/pytest_gpu.log)" | tee -a $OUT/summary.txt

prof_enc() {  # tag, env...
    local tag=$1; shift
    ( cd /tmp && env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- python - <<PY > $OUT/enc_$tag.log 2>&1
import sys, time, torch
sys.path.insert(0, "$ROOT")
from leann_amd.encoder import BertEncoder, config_for
from leann_amd.synth import CorpusSpec, SyntheticCorpus, pad_batch
dev = torch.device("cuda")
cfg = config_for("all-MiniLM-L6-v2")
enc = BertEncoder.random_init(cfg, 0).to(dev, dtype=torch.float16)
ids, lens = pad_batch(*SyntheticCorpus(CorpusSpec(n_chunks=14600)).chunks(), 256)
ti, tl = torch.from_numpy(ids).to(dev), torch.from_numpy(lens).to(dev)
flops = sum(cfg.flops_per_chunk(int(t)) for t in lens)
enc.encode_tokens_packed(ti, tl, 262144)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(4):
    enc.encode_tokens_packed(ti, tl, 262144)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
print({"tokens": int(tl.sum()), "chunks": len(lens), "ms": round(dt * 1e3, 2), "TFLOPs": round(flops / dt / 1e12, 1)})
PY
    )
    find /tmp/prof_$tag -name "*kernel_stats.csv" -exec cp {} $OUT/enc_${tag}_kernel_stats.csv \;
    echo "   enc $tag: $(tail -1 $OUT/enc_$tag.log)" | tee -a $OUT/summary.txt
}
echo "== 2. encoder per-kernel profile" | tee -a $OUT/summary.txt
prof_enc default LEANN_X=0
prof_enc linear LEANN_MI355X_LINEAR=1
python - $OUT <<'PY' | tee -a $OUT/summary.txt
import csv, sys
for tag in ("default", "linear"):
    try:
        rows = list(csv.DictReader(open(f"{sys.argv[1]}/enc_{tag}_kernel_stats.csv")))
    except OSError as e:
        print(tag, "no stats", e); continue
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"-- {tag}: total kernel ms {tot/1e6:.2f}")
    for r in rows[:16]:
        print(f'   {r["Name"][:90]:90s} calls={r["Calls"]:>5s} total_ms={float(r["TotalDurationNs"])/1e6:8.2f} avg_us={float(r["AverageNs"])/1e3:9.2f} pct={r["Percentage"]}')
PY
echo "== 3. encoder ops A/B (event timing, no profiler)" | tee -a $OUT/summary.txt
timeout 600 python scripts/encoder_ops_bench.py > $OUT/encoder_ops_bench.json 2> $OUT/encoder_ops_bench.err; echo "   rc=$?" | tee -a $OUT/summary.txt
echo "== 4. stored-embedding search micro-benchmark" | tee -a $OUT/summary.txt
for beam in 1 4; do
  timeout 300 python scripts/kernel_bench.py --beam $beam --deg 10 > $OUT/kernel_bench_beam${beam}_deg10.json 2>&1; echo "   beam $beam deg 10 rc=$? $(tail -c 400 $OUT/kernel_bench_beam${beam}_deg10.json | tr '\n' ' ')" | tee -a $OUT/summary.txt
done
cat $OUT/summary.txt
