#!/usr/bin/env python
"""SURVEY 8(d), the clustered-Gaussian TABLE-PROVIDER variant of config C2: 1M vectors, 1000 centres, sigma 0.15, D = 384, seed 7
(structured data, unlike the near-isotropic embeddings of a random-weight encoder; sigma is the NORM of the noise vector relative to the
unit-norm centre, i.e. 0.15 / sqrt(D) per coordinate -- 0.15 per coordinate would be noise of norm 2.9 around centres of norm 1: no
structure left, recall@10 0.46 at ef 64 in GPU session r3-6); the recompute provider is a gather from an
HBM-resident table instead of the BERT forward, so the timed region is the search machinery alone -- expand / visited / per-round
dedup / provider gather / fused distance + beam update -- in RECOMPUTE mode (lock-step rounds, sorted unique ids to the provider).

    python scripts/bench_table_provider.py [--chunks 1000000] [--batch 2048] [--steps 5] [--warmup 2]

Prints ONE JSON line (bench.py's keys; roofline = the fused distance / beam-update kernel k_update against the HBM peak)."""
import argparse
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=1_000_000)
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--ef", type=int, default=64)
    ap.add_argument("--beam", type=int, default=1)
    ap.add_argument("--centres", type=int, default=1000)
    ap.add_argument("--sigma", type=float, default=0.15)
    ap.add_argument("--M", type=int, default=32)
    ap.add_argument("--efc", type=int, default=200)
    args = ap.parse_args()
    from leann_amd import _lib
    from leann_amd.devmem import as_tensor
    from leann_amd.gpu_graph_build import build_graph_gpu
    from leann_amd.index import Mi355xIndex

    _lib.require_gpu()
    dev = torch.device("cuda", 0)
    n, D, B, K, W = args.chunks, 384, args.batch, args.steps, args.warmup
    t_all = time.time()
    g_ = torch.Generator(device=dev).manual_seed(7)
    cen = torch.nn.functional.normalize(torch.randn((args.centres, D), generator=g_, device=dev), dim=1)

    def draw(m):
        c = torch.randint(0, args.centres, (m,), generator=g_, device=dev)
        return torch.nn.functional.normalize(cen[c] + (args.sigma / D**0.5) * torch.randn((m, D), generator=g_, device=dev), dim=1)

    X = draw(n).contiguous()
    nq = B * (K + W + 1)
    Q = draw(nq).contiguous()
    t0 = time.time()
    g = build_graph_gpu(X, "mips", M=args.M, ef_construction=args.efc)
    t_graph = time.time() - t0
    idx = Mi355xIndex.from_csr(g, device=0)
    idx.set_stream(torch.cuda.current_stream().cuda_stream)
    keep = {}

    def provider(d_ids, cnt, stream):  # the table gather standing in for token gather + BERT forward
        ids = as_tensor(d_ids, (cnt,), "int32")
        keep["e"] = X.index_select(0, ids.long())
        return keep["e"].data_ptr()

    idx.set_provider(provider)
    from leann_amd.exact import exact_topk_ip

    gt = exact_topk_ip(Q, X, 10)[1].cpu().numpy()
    prm = idx.make_params(ef=args.ef, beam=args.beam, recompute=True, max_batch=B)
    for w in range(W):
        idx.search_device(Q[w * B : (w + 1) * B], 10, prm)
    agg = {"ndis": 0, "nunique": 0, "nrounds": 0}
    labels = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(K):
        lo = (W + s) * B
        _, l = idx.search_device(Q[lo : lo + B], 10, prm)
        labels.append(l)
        st = idx.stats()
        for k_ in agg:
            agg[k_] += st[k_]
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    lab = torch.cat(labels).cpu().numpy()
    rec = float(np.mean([len(set(lab[i]) & set(gt[W * B + i])) / 10 for i in range(K * B)]))
    idx.set_profiling(True)
    lo = (W + K) * B
    idx.search_device(Q[lo : lo + B], 10, prm)
    torch.cuda.synchronize()
    pst = idx.stats()
    idx.set_profiling(False)
    bytes_eval = D * 4 + 4
    upd_s = max(pst["update_span_ms"], 1e-9) * 1e-3
    ach = pst["ndis"] * bytes_eval / upd_s / 1e9
    deg0 = g.level0_degrees()
    print(json.dumps({
        "metric": "queries/sec at recall@10>=0.9, 1M-vector HNSW, recompute-mode search with a TABLE provider (SURVEY 8(d) clustered-Gaussian variant)",
        "value": round(K * B / elapsed, 3), "unit": "queries/s", "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": round(1e3 * elapsed / K, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{n} vectors = {args.centres} unit-norm centres + noise of norm ~{args.sigma} (N(0, {args.sigma}^2 / D I)), D = 384, seed 7, normalised; HNSW M={args.M} GPU-built "
                               f"(mean level-0 degree {deg0.mean():.1f}); ef_search={args.ef}, beam={args.beam}, top-10, {B} queries/step; provider = gather "
                               "from the HBM table (no encoder)", "baseline_config": "c2-table-provider-variant", "n_chunks": n, "queries_per_step": B},
        "recall_at_10": round(rec, 4),
        "roofline": {"bound": "hbm", "kernel": "lm::k_update<6,false,false,1,256> (fused gather + distance + beam update, rows by rank in the round's unique list)",
                     "achieved": round(ach, 2), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 5), "traffic": None,
                     "bytes_per_eval": bytes_eval, "evals_per_launch": round(pst["ndis"] / max(pst["update_launches"], 1), 1),
                     "us_per_launch": round(1e3 * pst["update_span_ms"] / max(pst["update_span_launches"], 1), 2),
                     "timing": "device wall-clock span per launch, one profiled step after the timed ones"},
        "per_query": {"distance_evals": round(agg["ndis"] / (K * B), 1), "provider_rows": round(agg["nunique"] / (K * B), 1),
                      "rounds_per_step": round(agg["nrounds"] / K, 1)},
        "time_split_profiled_step_ms": {"provider": round(pst["provider_ms"], 2), "expand": round(pst["expand_ms"], 2), "update": round(pst["update_ms"], 2)},
        "setup_s": {"total": round(time.time() - t_all), "build_graph": round(t_graph)}}), flush=True)


if __name__ == "__main__":
    main()
