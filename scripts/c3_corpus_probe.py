#!/usr/bin/env python3
"""What the exact top-10 of the synthetic corpus are made of.  (Written to find out why recall@10 of BASELINE configs[2] saturated near 0.86
at 10M chunks for the graph search AND for a brute-force ADC ranking alike -- profiles/r4_bench_c3_10M_chunks_*_with_diagnosis.json.  The
corpus was not the reason -- at 4M chunks 99.8 % of the true top-10 are chunks of the query's own document, PQ's ADC top-64 holds 94-99 % of
them -- the ground truth was: one GEMM call with 2.56e9 output elements, see leann_amd/exact.py.)  For a few corpus specs this script
embeds N chunks (the C3 encoder shape, fp16 kernels), takes the exact top-10 of 256 held-out queries and reports
  * what the true top-10 are made of: chunks of the query's own document / of its topic / strangers,
  * the score margins (own-document chunks vs the best stranger),
  * the share of the true top-10 inside a brute-force ADC top-L (PQ trained on the same embeddings).
One JSON line per spec.  Measurement script (GPU); nothing in the product imports it."""
import argparse
import json
import sys
import time
from dataclasses import replace
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

SPECS = {
    "default": {},
    "doc_signal_0.50": {"p_topic": 0.35, "p_doc": 0.50},
    "doc_signal_0.50_vocab_32": {"p_topic": 0.35, "p_doc": 0.50, "doc_vocab": 32},
    "docs_of_32_chunks": {"chunks_per_doc": 32},
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=4_000_000)
    ap.add_argument("--model", default="BAAI/bge-small-en-v1.5")
    ap.add_argument("--specs", default="default,doc_signal_0.50,docs_of_32_chunks")
    ap.add_argument("--pq-bytes", type=int, default=96)
    ap.add_argument("--no-pq", default="default", help="comma list of specs whose ADC part is skipped")
    args = ap.parse_args()
    from leann_amd import _lib
    from leann_amd.encoder import BertEncoder
    from leann_amd.pq import encode_pq, train_pq
    from leann_amd.recompute import RecomputeProvider
    from leann_amd.synth import CorpusSpec, SyntheticCorpus
    from leann_amd.token_store import TokenStore

    _lib.require_gpu()
    dev = torch.device("cuda", 0)
    n, nq = args.chunks, 256
    enc = BertEncoder.load(args.model, allow_random=True).to(dev, dtype=torch.float16).eval()
    if enc.weights_source == "random":
        enc.cfg = replace(enc.cfg, pooling="mean")
    D = enc.cfg.hidden
    for name in args.specs.split(","):
        t0 = time.time()
        spec = CorpusSpec(n_chunks=n, seed=1234, n_topics=max(1000, n // 1000), **SPECS[name])
        corpus = SyntheticCorpus(spec)
        tok, off = corpus.chunks_torch(dev)
        provider = RecomputeProvider(enc, TokenStore(tok, off), (D + 63) // 64 * 64, dev)
        X = torch.empty((n, D), dtype=torch.float32, device=dev)
        for b0 in range(0, n, 32768):
            ids = torch.arange(b0, min(n, b0 + 32768), dtype=torch.int32, device=dev)
            X[b0 : b0 + ids.shape[0]] = provider.embed_ids(ids)
        qt, qo, qdocs = corpus.queries(nq, seed=4321)
        Q = RecomputeProvider(enc, TokenStore(qt, qo), provider.dp, dev).embed_ids(torch.arange(nq, dtype=torch.int32, device=dev)).contiguous()
        S = Q @ X.T  # (nq, n)
        top = torch.topk(S, 10, dim=1)
        gt = top.indices.cpu().numpy()
        doc_of = np.arange(n, dtype=np.int64) // spec.chunks_per_doc
        same_doc = doc_of[gt] == qdocs[:, None]
        same_topic = corpus.doc_topic[doc_of[gt]] == corpus.doc_topic[qdocs][:, None]
        # margins: mean score of the own-document chunks, the 10th best score, the best score outside the document
        qd = torch.from_numpy(qdocs).to(dev)
        sib = (torch.arange(spec.chunks_per_doc, device=dev)[None, :] + (qd * spec.chunks_per_doc)[:, None]).clamp(max=n - 1)
        s_sib = torch.gather(S, 1, sib)
        S2 = S.clone()
        S2.scatter_(1, sib, float("-inf"))
        best_stranger = S2.max(1).values
        out = {"spec": name, "overrides": SPECS[name], "n_chunks": n, "n_topics": spec.n_topics, "chunks_per_doc": spec.chunks_per_doc,
               "true_top10_own_document": round(float(same_doc.mean()), 4), "true_top10_same_topic_other_document": round(float((same_topic & ~same_doc).mean()), 4),
               "true_top10_other_topic": round(float((~same_topic).mean()), 4),
               "score_mean_own_document": round(float(s_sib.mean()), 5), "score_std_within_own_document": round(float(s_sib.std(1).mean()), 5),
               "score_10th_best": round(float(top.values[:, 9].mean()), 5), "score_best_stranger": round(float(best_stranger.mean()), 5),
               "score_mean_all": round(float(S.mean()), 5), "score_std_all": round(float(S.std()), 5)}
        del S, S2
        if name not in args.no_pq.split(","):
            cb = train_pq(X, args.pq_bytes, iters=10)
            codes = encode_pq(X, cb)
            m_, dsub = args.pq_bytes, D // args.pq_bytes
            Ls = (64, 256, 1024)
            hit = {L_: 0 for L_ in Ls}
            nd = 64
            for qi in range(nd):
                lut = -(cb.to(dev) * Q[qi].view(m_, 1, dsub)).sum(-1)
                adc = torch.zeros((n,), dtype=torch.float32, device=dev)
                for c0 in range(0, n, 2_000_000):
                    cc = codes[c0 : c0 + 2_000_000].long()
                    adc[c0 : c0 + cc.shape[0]] = lut[torch.arange(m_, device=dev)[None, :], cc].sum(1)
                t_ = torch.topk(adc, max(Ls), largest=False).indices.cpu().numpy()
                truth = set(gt[qi].tolist())
                for L_ in Ls:
                    hit[L_] += len(truth & set(t_[:L_].tolist()))
            for L_ in Ls:
                out[f"true_top10_inside_bruteforce_adc_top{L_}"] = round(hit[L_] / (10 * nd), 4)
            del codes, cb
        out["seconds"] = round(time.time() - t0, 1)
        print(json.dumps(out), flush=True)
        del X, provider, corpus
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
