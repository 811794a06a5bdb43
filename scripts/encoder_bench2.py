#!/usr/bin/env python
"""Encoder knobs A/B (run with different env vars): packed path at a FIXED token count."""
import json, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from leann_amd.encoder import BertEncoder, config_for
from leann_amd.synth import CorpusSpec, SyntheticCorpus, pad_batch
dev = torch.device("cuda")
cfg = config_for("all-MiniLM-L6-v2")
enc = BertEncoder.random_init(cfg, 0).to(dev, dtype=torch.float16)
c = SyntheticCorpus(CorpusSpec(n_chunks=8192))
ids, lens = pad_batch(*c.chunks(), 256)
ti, tl = torch.from_numpy(ids).to(dev), torch.from_numpy(lens).to(dev)
flops = sum(cfg.flops_per_chunk(int(t)) for t in lens)
for _ in range(3):
    enc.encode_tokens_packed(ti, tl, 1 << 21)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    enc.encode_tokens_packed(ti, tl, 1 << 21)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(json.dumps({"env": {k: v for k, v in os.environ.items() if "TUNABLE" in k or "FA_PREFER" in k or "BLAS" in k}, "ms": round(dt * 1e3, 2), "TFLOPs": round(flops / dt / 1e12, 1)}))
