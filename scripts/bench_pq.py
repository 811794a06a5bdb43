#!/usr/bin/env python
"""DiskANN-style path at 1M chunks (config C3 in miniature): PQ-ADC persistent traversal + ONE deferred rerank
through the real recompute provider (HBM token store -> BERT forward), on the bench's synthetic corpus."""
import argparse
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from leann_amd.gpu_graph_build import build_graph_gpu
from leann_amd.index import Mi355xIndex
from leann_amd.pq import encode_pq, flat_graph, train_pq

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--d", type=int, default=384)
ap.add_argument("--m", type=int, default=96)
ap.add_argument("--batch", type=int, default=1024)
args = ap.parse_args()
dev = torch.device("cuda")
from leann_amd.encoder import BertEncoder
from leann_amd.recompute import RecomputeProvider
from leann_amd.synth import CorpusSpec, SyntheticCorpus
from leann_amd.token_store import TokenStore

corpus = SyntheticCorpus(CorpusSpec(n_chunks=args.n, seed=1234))
tok, off = corpus.chunks()
tokens = TokenStore(tok, off)
enc = BertEncoder.load("sentence-transformers/all-MiniLM-L6-v2", allow_random=True).to(dev, dtype=torch.float16).eval()
provider = RecomputeProvider(enc, tokens, 384, dev)
X = torch.empty((args.n, 384), dtype=torch.float32, device=dev)
for b0 in range(0, args.n, 32768):
    ids = torch.arange(b0, min(args.n, b0 + 32768), dtype=torch.int32, device=dev)
    X[b0 : b0 + ids.shape[0]] = provider.embed_ids(ids)
qt, qo, _ = corpus.queries(args.batch, seed=4321)
Q = RecomputeProvider(enc, TokenStore(qt, qo), 384, dev).embed_ids(torch.arange(args.batch, dtype=torch.int32, device=dev)).contiguous()
t0 = time.time()
graph = build_graph_gpu(X, "mips", M=32, ef_construction=128)
t_graph = time.time() - t0
fg = flat_graph(graph, X.cpu().numpy())
t0 = time.time()
cb = train_pq(X, args.m, iters=10)
codes = encode_pq(X, cb)
torch.cuda.synchronize()
t_pq = time.time() - t0
idx = Mi355xIndex.from_csr(fg)
idx.set_stream(torch.cuda.current_stream().cuda_stream)
idx.attach_pq(cb.cpu().numpy(), codes.cpu().numpy())
idx.set_provider(provider)
idx.set_profiling(True)
gt = torch.topk(Q @ X.T, 10, dim=1).indices.cpu().numpy()
out = {"n": args.n, "d": args.d, "pq_bytes": args.m, "graph_build_s": round(t_graph, 1), "pq_train_encode_s": round(t_pq, 1), "runs": []}
for L, W in ((64, 4), (64, 64), (128, 16), (256, 16)):
    prm = idx.make_pq_params(L, W, use_deferred_fetch=True)
    idx.pq_search_device(Q, 10, prm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        labels, dist = idx.pq_search_device(Q, 10, prm)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    st = idx.stats()
    l = labels.cpu().numpy()
    rec = float(np.mean([len(set(l[i]) & set(gt[i])) / 10 for i in range(args.batch)]))
    code_bytes = st["ndis"] * (args.m + 4)
    out["runs"].append({"complexity": L, "beam_width": W, "queries_per_s": round(args.batch / dt, 1), "recall_at_10": round(rec, 4),
                        "adc_evals_per_query": round(st["ndis"] / args.batch, 1), "reranked_unique_per_query": round(st["nunique"] / args.batch, 1),
                        "traverse_kernel_ms": round(st["update_ms"], 3),
                        "traverse_code_GBps": round(code_bytes / max(st["update_ms"], 1e-9) / 1e6, 1), "iterations_max": st["nrounds"]})
print(json.dumps(out))
