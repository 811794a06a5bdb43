#!/usr/bin/env python
"""Stored-embedding search on the bench's own 1M-chunk index (MiniLM-shaped embeddings, GPU-built HNSW M=32): persistent kernel
with a wave / a workgroup per query vs lock-step rounds, beam 1 and 4, 8192 queries in flight.  One JSON line."""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from leann_amd.encoder import BertEncoder, config_for
from leann_amd.gpu_graph_build import build_graph_gpu
from leann_amd.index import Mi355xIndex
from leann_amd.recompute import RecomputeProvider
from leann_amd.synth import CorpusSpec, SyntheticCorpus
from leann_amd.token_store import TokenStore

n, dev = 1_000_000, torch.device("cuda")
corpus = SyntheticCorpus(CorpusSpec(n_chunks=n, seed=1234))
tok, off = corpus.chunks()
enc = BertEncoder.random_init(config_for("all-MiniLM-L6-v2"), 0).to(dev, dtype=torch.float16).eval()
prov = RecomputeProvider(enc, TokenStore(tok, off), 384, dev)
X = torch.empty((n, 384), dtype=torch.float32, device=dev)
for b0 in range(0, n, 32768):
    ids = torch.arange(b0, min(n, b0 + 32768), dtype=torch.int32, device=dev)
    X[b0:b0 + ids.shape[0]] = prov.embed_ids(ids)
g = build_graph_gpu(X, "mips", M=32, ef_construction=200)
qt, qo, _ = corpus.queries(8192, seed=4321)
Q = RecomputeProvider(enc, TokenStore(qt, qo), 384, dev).embed_ids(torch.arange(8192, dtype=torch.int32, device=dev)).contiguous()
idx = Mi355xIndex.from_csr(g)
idx.set_stream(torch.cuda.current_stream().cuda_stream)
idx.attach_table(X)
idx.set_profiling(True)
out = {"mean_degree0": float(g.level0_degrees().mean())}
ref = {}
for rnd in range(2):
    for name, persistent, wave in (("persistent_wave_per_query", 1, 1), ("persistent_workgroup_per_query", 1, 0), ("lockstep", 0, 0)):
        idx.set_option("persistent_table", persistent)
        idx.set_option("persistent_wave", wave)
        for beam in (1, 4):
            prm = idx.make_params(ef=64, beam=beam, recompute=False, max_batch=16384)
            d, l = idx.search_device(Q, 10, prm)
            st = idx.stats()
            ms = max(st["update_span_ms"], 1e-9)
            key = f"{name}_beam{beam}"
            if beam not in ref:
                ref[beam] = (d.clone(), l.clone())
            same = bool(torch.equal(ref[beam][0], d) and torch.equal(ref[beam][1], l))
            out.setdefault(key, []).append({"GBps": round(st["ndis"] * 1540 / ms / 1e6, 1), "ms": round(ms, 3), "ndis_per_query": round(st["ndis"] / 8192, 1),
                                            "identical_results": same})
# round 6: queries in flight -- beam 1 is a latency chain per query: throughput = queries resident per CU / chain latency.  (The same sweep also carried a
# five-waves-per-SIMD register allocation of the kernel, option "persistent_occupancy": slower everywhere, deleted;
# profiles/r6_table_mode_queries_in_flight_and_occupancy_sweep.json.)
if "--occupancy-sweep" in sys.argv or "--queries-in-flight" in sys.argv:
    qt2, qo2, _ = corpus.queries(32768, seed=97531)
    Q2 = RecomputeProvider(enc, TokenStore(qt2, qo2), 384, dev).embed_ids(torch.arange(32768, dtype=torch.int32, device=dev)).contiguous()
    sweep = []
    refs = {}
    for rnd in range(2):
        for nq in (8192, 16384, 32768):
            for occ in (0,):
                for wave in (1, 0):
                    idx.set_option("persistent_table", 1)
                    idx.set_option("persistent_wave", wave)
                    for beam in (1, 4):
                        prm = idx.make_params(ef=64, beam=beam, recompute=False, max_batch=32768)
                        d, l = idx.search_device(Q2[:nq], 10, prm)
                        st = idx.stats()
                        ms = max(st["update_span_ms"], 1e-9)
                        key = (nq, beam)
                        if key not in refs:
                            refs[key] = l.clone()
                        sweep.append({"round": rnd, "queries": nq, "waves_per_simd_target": occ or "compiler (4)", "form": "wave" if wave else "workgroup", "beam": beam,
                                      "GBps": round(st["ndis"] * 1540 / ms / 1e6, 1), "ms": round(ms, 3), "identical_labels": bool(torch.equal(refs[key], l))})
    out["occupancy_sweep"] = sweep
print(json.dumps(out))
