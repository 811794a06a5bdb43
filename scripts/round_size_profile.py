#!/usr/bin/env python
"""Where does a 2048-query benchmark step spend its encoder time, BY SIZE OF THE ROUND'S FORWARD?  The lock-step search runs ~150 rounds per
step; the early rounds recompute hundreds of thousands of chunks, the last ones belong to a thin tail of slow queries -- forwards of a few
thousand tokens that fill a fraction of the chip.  This script runs bench.py's C2 step over the PYTHON form of the provider with a
synchronising wrapper around every provider call and prints the rounds bucketed by tokens per call: rounds, tokens, time, time per token
against the big rounds' rate -- i.e. how much of the step a perfectly size-independent encoder would save.

    python scripts/round_size_profile.py [--chunks 1000000] [--batch 2048] [--steps 2]
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ["LEANN_MI355X_NATIVE_PROVIDER"] = "0"  # the wrapper below needs the per-round call to come back to the interpreter
import numpy as np
import torch

from leann_amd.encoder import BertEncoder, config_for
from leann_amd.gpu_graph_build import build_graph_gpu
from leann_amd.index import Mi355xIndex
from leann_amd.recompute import RecomputeProvider
from leann_amd.synth import CorpusSpec, SyntheticCorpus
from leann_amd.token_store import TokenStore

ap = argparse.ArgumentParser()
ap.add_argument("--chunks", type=int, default=1_000_000)
ap.add_argument("--batch", type=int, default=2048)
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--ef", type=int, default=64)
ap.add_argument("--model", default="sentence-transformers/all-MiniLM-L6-v2")
args = ap.parse_args()

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
corpus = SyntheticCorpus(CorpusSpec(n_chunks=args.chunks, seed=1234))
tok, off = corpus.chunks()
tokens = TokenStore(tok, off, device=0)
cfg = config_for(args.model)
enc = BertEncoder.load(args.model, allow_random=True).to(dev, dtype=torch.float16).eval()
D = cfg.hidden
provider = RecomputeProvider(enc, tokens, (D + 63) // 64 * 64, dev)
X = torch.empty((args.chunks, D), dtype=torch.float32, device=dev)
os.environ["LEANN_MI355X_NATIVE_PROVIDER"] = "1"  # (corpus embedding: the fast path)
for b0 in range(0, args.chunks, 32768):
    ids = torch.arange(b0, min(args.chunks, b0 + 32768), dtype=torch.int32, device=dev)
    X[b0: b0 + ids.shape[0]] = provider.embed_ids(ids)
os.environ["LEANN_MI355X_NATIVE_PROVIDER"] = "0"
g = build_graph_gpu(X, "mips", M=32, ef_construction=200)
idx = Mi355xIndex.from_csr(g, device=0)
idx.set_stream(torch.cuda.current_stream().cuda_stream)
qt, qo, _ = corpus.queries(args.batch * (args.steps + 1), seed=4321)
Q = RecomputeProvider(enc, TokenStore(qt, qo, device=0), provider.dp, dev).embed_ids(torch.arange(args.batch * (args.steps + 1), dtype=torch.int32, device=dev))
del X
calls = []


def logging_provider(d_ids_ptr, n, stream_ptr):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    p = provider(d_ids_ptr, n, stream_ptr)
    ntok = int(provider._lens[:n].sum())
    torch.cuda.synchronize()
    calls.append((n, ntok, time.perf_counter() - t0))
    return p


idx.set_provider(logging_provider)
assert not idx.native_provider
prm = idx.make_params(ef=args.ef, beam=1, recompute=True, max_batch=args.batch)
idx.search_device(Q[: args.batch], 10, prm)  # warm-up
torch.cuda.synchronize()
calls.clear()
t0 = time.perf_counter()
for s in range(1, args.steps + 1):
    idx.search_device(Q[s * args.batch: (s + 1) * args.batch], 10, prm)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
edges = [0, 4096, 8192, 16384, 45056, 100_000, 200_000, 400_000, 800_000, 1 << 62]
big = [c for c in calls if c[1] >= 400_000]
rate = sum(c[2] for c in big) / max(sum(c[1] for c in big), 1)  # seconds per token of the big rounds
rows = []
for lo, hi in zip(edges[:-1], edges[1:]):
    b = [c for c in calls if lo <= c[1] < hi]
    if not b:
        continue
    t, k = sum(c[2] for c in b), sum(c[1] for c in b)
    rows.append({"tokens_per_round": f"{lo}..{hi if hi < 1 << 62 else 'inf'}", "rounds_per_step": round(len(b) / args.steps, 1), "tokens_per_step": k // args.steps,
                 "ms_per_step": round(1e3 * t / args.steps, 1), "share_of_provider_time": round(t / sum(c[2] for c in calls), 4),
                 "ns_per_token": round(1e9 * t / max(k, 1), 2), "ms_per_step_at_the_big_rounds_rate": round(1e3 * k * rate / args.steps, 1)})
    print(json.dumps(rows[-1]), flush=True)
tot = sum(c[2] for c in calls)
print(json.dumps({"steps": args.steps, "queries_per_step": args.batch, "rounds_per_step": len(calls) / args.steps, "wall_ms_per_step": round(1e3 * wall / args.steps, 1),
                  "provider_ms_per_step": round(1e3 * tot / args.steps, 1), "big_round_ns_per_token": round(1e9 * rate, 3),
                  "provider_ms_per_step_if_every_round_ran_at_the_big_rounds_rate": round(1e3 * rate * sum(c[1] for c in calls) / args.steps, 1),
                  "note": "Python form of the provider with a device synchronisation on both sides of every call: the sizes are the product path's, the per-call times carry the "
                          "interpreter's launch path (a few hundred microseconds per small call more than the library-side provider)"}), flush=True)
