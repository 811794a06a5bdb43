#!/usr/bin/env python
"""Small-batch latency of the recompute search (LEANN's real call is one query at a time, leann/api.py:644-796) on a 200k-chunk
index (set-up ~20 s), for A/B runs of host-side switches:   LEANN_MI355X_ONECALL=1 python scripts/latency_bench.py
Prints one JSON line: p50 / mean latency at B = 1, 4, 16, 64, 256, for the library-side provider and the Python provider (same queries)."""
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
if os.environ.get("LAT_DIRECT") == "1":  # option single_query_direct: a one-query pass hands its new-list to the provider as it is
    idx.set_option("single_query_direct", 1)
qt, qo, _ = corpus.queries(2048, seed=4321)
Q = RecomputeProvider(enc, TokenStore(qt, qo, device=0), 384, dev).embed_ids(torch.arange(2048, dtype=torch.int32, device=dev)).contiguous()
# A/B in one process on the SAME queries: the library-side provider (csrc/lm_recompute.hip, the default) against the Python provider
# (LEANN_MI355X_NATIVE_PROVIDER=0), interleaved per batch size; results must be identical
def attach(native: bool):
    if native:
        os.environ.pop("LEANN_MI355X_NATIVE_PROVIDER", None)
    else:
        os.environ["LEANN_MI355X_NATIVE_PROVIDER"] = "0"
    idx.set_provider(provider)
    os.environ.pop("LEANN_MI355X_NATIVE_PROVIDER", None)
    return idx.native_provider


forms = [("library_side_provider", True), ("python_provider", False)] if attach(True) else [("python_provider", False)]
rows, lo, same = [], 0, True
for b in (() if os.environ.get("LAT_VARIANTS_ONLY") == "1" else (1,) if "--speculate" in sys.argv else tuple(int(v) for v in os.environ.get("LAT_BATCHES", "1,4,16,64,256").split(","))):  # (the prefetch sweep below: B = 1 only)
    prm = idx.make_params(ef=64, beam=1, recompute=True, max_batch=b)
    reps = 24 if b == 1 else (12 if b <= 16 else 4)
    if lo + b * (reps + 1) > Q.shape[0]:
        lo = 0
    row, ref = {"batch": b}, None
    for name, native in forms:
        assert attach(native) == native
        idx.search_device(Q[lo : lo + b].contiguous(), 10, prm)
        lat, outs, p = [], [], lo + b
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            d, l = idx.search_device(Q[p : p + b].contiguous(), 10, prm)
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - t0) * 1e3)
            outs.append((d, l))
            p += b
        st = idx.stats()
        row[name] = {"p50_ms": round(float(np.median(lat)), 2), "mean_ms": round(float(np.mean(lat)), 2), "queries_per_s": round(b / float(np.mean(lat)) * 1e3, 1),
                     "rounds_last_call": st["nrounds"]}
        if ref is None:
            ref = outs
        else:
            same &= all(torch.equal(a[0], c[0]) and torch.equal(a[1], c[1]) for a, c in zip(ref, outs))
    lo += b * (reps + 1)
    rows.append(row)
attach(True)
# --speculate 0,4,16: B = 1 latency with the speculative prefetch (option "speculate": a round also embeds the neighbours of its S best
# unexpanded candidates), same queries for every S; labels must not change
spec_rows = []
if "--speculate" in sys.argv:
    svals = [int(s) for s in sys.argv[sys.argv.index("--speculate") + 1].split(",")]
    prm = idx.make_params(ef=64, beam=1, recompute=True, max_batch=1)
    ref_labels = None
    for S in svals:
        idx.set_option("speculate", S)
        idx.search_device(Q[0:1].contiguous(), 10, prm)
        f0 = provider.native_stats()["forwards"]
        lat, labels, chunks = [], [], 0
        for i in range(1, 33):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            d, l = idx.search_device(Q[i : i + 1].contiguous(), 10, prm)
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - t0) * 1e3)
            labels.append(l.cpu())
            chunks += int(idx.stats()["nunique"])
        if ref_labels is None:
            ref_labels = labels
        spec_rows.append({"speculate": S, "p50_ms": round(float(np.median(lat)), 2), "mean_ms": round(float(np.mean(lat)), 2),
                          "recomputed_chunks_per_query": round(chunks / 32, 1), "forwards_per_query": round((provider.native_stats()["forwards"] - f0) / 32, 1),
                          "queries_with_the_labels_of_S0": int(sum(torch.equal(a, b) for a, b in zip(ref_labels, labels)))})
    idx.set_option("speculate", 0)
# LAT_VARIANTS=default,direct[,KEY=VALUE+KEY=VALUE ...]: switches A/B'd IN ONE PROCESS on the same queries (library-side provider only):
# "direct" = index option single_query_direct, KEY=VALUE = an environment switch the library reads per call (e.g. LEANN_MI355X_SMALL_TOKENS=32768);
# the labels of every variant must equal the first variant's.  (Round 5 timed the two small-forward kernels of round 4 this way -- row-complete
# GEMM + LayerNorm: B = 1 p50 47.5 ms, whole-layer kernel: 60.5 ms, against 38.2 ms for the default -- and deleted them:
# profiles/r5_latency_small_forward_variants_200k.json.)
variant_rows = []
if os.environ.get("LAT_VARIANTS"):
    ref_by_b = {}
    for vname in os.environ["LAT_VARIANTS"].split(","):
        parts = set(vname.split("+"))
        set_here = {kv.split("=", 1)[0]: kv.split("=", 1)[1] for kv in parts if "=" in kv}
        os.environ.update(set_here)
        idx.set_option("single_query_direct", 1 if "direct" in parts else 0)
        vrow = {"variant": vname}
        for b in tuple(int(v) for v in os.environ.get("LAT_BATCHES", "1,4,16,64,256").split(",")):
            prm = idx.make_params(ef=64, beam=1, recompute=True, max_batch=b, batch_size=int(os.environ.get("LAT_BATCH_SIZE", "0")))  # LAT_BATCH_SIZE: dynamic batching target
            reps = 24 if b == 1 else (12 if b <= 16 else 4)
            idx.search_device(Q[0:b].contiguous(), 10, prm)
            lat, labels, p = [], [], b
            for _ in range(reps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                d, l = idx.search_device(Q[p : p + b].contiguous(), 10, prm)
                torch.cuda.synchronize()
                lat.append((time.perf_counter() - t0) * 1e3)
                labels.append(l.cpu())
                p += b
            if b not in ref_by_b:
                ref_by_b[b] = labels
            vrow[f"B{b}"] = {"p50_ms": round(float(np.median(lat)), 2), "mean_ms": round(float(np.mean(lat)), 2), "queries_per_s": round(b / float(np.mean(lat)) * 1e3, 1),
                             "rounds_last_call": idx.stats()["nrounds"], "calls_with_the_first_variants_labels": int(sum(torch.equal(a, c) for a, c in zip(ref_by_b[b], labels))), "calls": reps}
        variant_rows.append(vrow)
        for k_ in set_here:
            os.environ.pop(k_, None)
    idx.set_option("single_query_direct", 0)
# --frontier: B = 1 latency vs recall over efSearch x beam_width x batch_size (bench.py: latency_frontier), exact ground truth on this index
frontier = None
if "--frontier" in sys.argv:
    from bench import latency_frontier
    from leann_amd.exact import exact_topk_ip

    gt = exact_topk_ip(Q, X, 10)[1].cpu().numpy()

    def recall(labels_np, rws):
        return sum(len(set(labels_np[i].tolist()) & set(gt[r].tolist())) for i, r in enumerate(rws)) / (10 * len(rws))

    frontier = latency_frontier(idx, Q, recall, 64, efs=(16, 32, 64), beams=(1, 2, 4, 8), batch_sizes=(0, 32, 64, 128), reps=16, budget_s=240.0)
print(json.dumps({"chunks": n, "small_batch_latency_frontier": frontier, "small_forward_variants": variant_rows, "b1_speculative_prefetch": spec_rows, "switches": {k: v for k, v in os.environ.items() if k.startswith("LEANN_MI355X_")}, "latency": rows,
                  "identical_results_between_the_providers": same, "library_side_provider_stats": provider.native_stats()}))
