#!/bin/bash
# Round 5, GPU session 13 (~4.5 GPU-minutes): head-major QKV layout (the projection writes [36 heads][tokens][32], attention generation 3 reads a
# (sequence, head)'s K / V / Q rows as contiguous blocks) against the row-major layout: whole encoder A/B, per-kernel times from the library's event
# pairs, then the encoder / provider / pipeline GPU tests on the new default.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r5s13; rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 300 python scripts/encoder_switch_ab.py sentence-transformers/all-MiniLM-L6-v2 22000 1048576 "-" "LEANN_MI355X_QKV_LAYOUT=0" "LEANN_MI355X_ATTN3=9" > $OUT/encoder_switch_ab.jsonl 2> $OUT/encoder_switch_ab.err; echo "encoder ab rc=$?"; cut -c1-300 $OUT/encoder_switch_ab.jsonl; tail -2 $OUT/encoder_switch_ab.err | cut -c1-300
python - > $OUT/kernel_times.json 2> $OUT/kernel_times.err <<'PY'
import json, os, sys, time
sys.path.insert(0, ".")
import torch
from leann_amd import _lib
from leann_amd.encoder import BertEncoder, config_for
from leann_amd.synth import CorpusSpec, SyntheticCorpus, pad_batch
dev = torch.device("cuda")
enc = BertEncoder.random_init(config_for("sentence-transformers/all-MiniLM-L6-v2", strict=True), 0).to(dev, dtype=torch.float16)
ids, lens = pad_batch(*SyntheticCorpus(CorpusSpec(n_chunks=11000)).chunks(), 256)
ti, tl = torch.from_numpy(ids).to(dev), torch.from_numpy(lens).to(dev)
out = {}
for name, env in (("head_major", {}), ("row_major", {"LEANN_MI355X_QKV_LAYOUT": "0"})):
    os.environ.pop("LEANN_MI355X_QKV_LAYOUT", None); os.environ.update(env)
    enc.encode_tokens_packed(ti, tl, 1048576); torch.cuda.synchronize()
    _lib.kernel_timing_enable((1 << _lib.KT_COUNT) - 1); _lib.kernel_timing_read(reset=True)
    for _ in range(3): enc.encode_tokens_packed(ti, tl, 1048576)
    torch.cuda.synchronize()
    kt = _lib.kernel_timing_read(reset=True); _lib.kernel_timing_enable(0)
    out[name] = {k: {"launches": v["launches"], "avg_us": round(1e3 * v["ms"] / max(v["launches"], 1), 1), "TFLOPs": round(v["work"] / max(v["ms"], 1e-9) / 1e9, 1)} for k, v in kt.items() if v["launches"]}
print(json.dumps(out))
PY
cut -c1-1200 $OUT/kernel_times.json; tail -2 $OUT/kernel_times.err | cut -c1-300
timeout -k 10 400 python -m pytest tests/test_gpu_encoder_kernels.py tests/test_gpu_native_provider.py tests/test_gpu_pipeline.py tests/test_config1_golden.py -m gpu -q > $OUT/pytest_encoder_provider_pipeline.log 2>&1; echo "pytest rc=$? $(tail -1 $OUT/pytest_encoder_provider_pipeline.log)"; grep -E "^(FAILED|ERROR)|^E  " $OUT/pytest_encoder_provider_pipeline.log | head -12 | cut -c1-250
