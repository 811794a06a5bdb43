#!/usr/bin/env python
"""Encoder throughput A/B on the GPU: packed varlen path vs length-bucketed padded path."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from leann_amd.encoder import BertEncoder, config_for
from leann_amd.synth import CorpusSpec, SyntheticCorpus, pad_batch

dev = torch.device("cuda")
cfg = config_for("all-MiniLM-L6-v2")
enc = BertEncoder.random_init(cfg, 0).to(dev, dtype=torch.float16)
c = SyntheticCorpus(CorpusSpec(n_chunks=16384))
tok, off = c.chunks()
ids, lens = pad_batch(tok, off, 256)
ti, tl = torch.from_numpy(ids).to(dev), torch.from_numpy(lens).to(dev)
flops = sum(cfg.flops_per_chunk(int(t)) for t in lens)
out = {}
ref = None
for name, fn in (("padded_b2048", lambda: enc.encode_tokens_padded(ti, tl, 2048, 32)),
                 ("padded_b1024_bucket16", lambda: enc.encode_tokens_padded(ti, tl, 1024, 16)),
                 ("packed_256k", lambda: enc.encode_tokens_packed(ti, tl, 262144)),
                 ("packed_128k", lambda: enc.encode_tokens_packed(ti, tl, 131072)),
                 ("packed_512k", lambda: enc.encode_tokens_packed(ti, tl, 524288))):
    try:
        e = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            e = fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        if ref is None:
            ref = e
        out[name] = {"ms": round(dt * 1e3, 2), "chunks_per_s": round(len(lens) / dt), "TFLOPs": round(flops / dt / 1e12, 1),
                     "max_abs_diff_vs_first": float((e - ref).abs().max())}
    except Exception as ex:  # noqa: BLE001
        out[name] = {"error": repr(ex)[:300]}
# small-round regime (what a search round looks like): 2000 chunks
ti2, tl2 = ti[:2000], tl[:2000]
for name, fn in (("small_padded", lambda: enc.encode_tokens_padded(ti2, tl2, 2048, 32)), ("small_packed", lambda: enc.encode_tokens_packed(ti2, tl2, 393216))):
    try:
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        out[name] = {"ms": round((time.perf_counter() - t0) / 5 * 1e3, 2)}
    except Exception as ex:  # noqa: BLE001
        out[name] = {"error": repr(ex)[:300]}
print(json.dumps(out, indent=1))
