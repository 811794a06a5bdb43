#!/usr/bin/env python
"""Small-batch latency of the recompute search (LEANN's real call is one query at a time, leann/api.py:644-796) on a 200k-chunk
index (set-up ~20 s), for A/B runs of host-side switches:   LEANN_MI355X_ONECALL=1 python scripts/latency_bench.py
Prints one JSON line: p50 / mean latency at B = 1, 4, 16."""
import json, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from leann_amd import _lib
from leann_amd.encoder import BertEncoder, config_for
from leann_amd.gpu_graph_build import build_graph_gpu
from leann_amd.index import Mi355xIndex
from leann_amd.recompute import RecomputeProvider
from leann_amd.synth import CorpusSpec, SyntheticCorpus
from leann_amd.token_store import TokenStore

_lib.require_gpu()
dev = torch.device("cuda", 0)
n = int(os.environ.get("LAT_CHUNKS", "200000"))
corpus = SyntheticCorpus(CorpusSpec(n_chunks=n, seed=1234))
tok, off = corpus.chunks()
enc = BertEncoder.load("sentence-transformers/all-MiniLM-L6-v2", allow_random=True).to(dev, dtype=torch.float16).eval()
provider = RecomputeProvider(enc, TokenStore(tok, off, device=0), 384, dev)
X = torch.empty((n, 384), dtype=torch.float32, device=dev)
for b0 in range(0, n, 32768):
    ids = torch.arange(b0, min(n, b0 + 32768), dtype=torch.int32, device=dev)
    X[b0 : b0 + ids.shape[0]] = provider.embed_ids(ids)
g = build_graph_gpu(X, "mips", M=32, ef_construction=200)
idx = Mi355xIndex.from_csr(g, device=0)
idx.set_stream(torch.cuda.current_stream().cuda_stream)
idx.set_provider(provider)
qt, qo, _ = corpus.queries(1024, seed=4321)
Q = RecomputeProvider(enc, TokenStore(qt, qo, device=0), 384, dev).embed_ids(torch.arange(1024, dtype=torch.int32, device=dev)).contiguous()
rows, lo = [], 0
for b in (1, 4, 16):
    prm = idx.make_params(ef=64, beam=1, recompute=True, max_batch=b)
    idx.search_device(Q[lo : lo + b].contiguous(), 10, prm)
    lo += b
    lat = []
    for _ in range(24 if b == 1 else 12):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        idx.search_device(Q[lo : lo + b].contiguous(), 10, prm)
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t0) * 1e3)
        lo += b
    st = idx.stats()
    rows.append({"batch": b, "p50_ms": round(float(np.median(lat)), 2), "mean_ms": round(float(np.mean(lat)), 2), "rounds_last_call": st["nrounds"]})
print(json.dumps({"chunks": n, "switches": {k: v for k, v in os.environ.items() if k.startswith("LEANN_MI355X_")}, "latency": rows}))
