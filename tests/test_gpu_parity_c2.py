"""Parity at BASELINE.json configs[1] (the headline configuration): 1 M synthetic chunks, MiniLM-L6 shaped encoder, HNSW M = 32,
ef_search = 64, beam 1 -- on the very index bench.py measures.  GPU ids, distances AND distance-evaluation counts must be
identical to the set-semantics oracle for 256 queries in stored-embedding mode (one-launch persistent kernel and lock-step
rounds); the independent heap-based faiss transcription must agree as well; in recompute mode the oracle replays the GPU
encoder's own per-round outputs for 16 queries (it must be asked for exactly the same sorted unique ids every round).
(the largest bit-exact comparison used to be N = 4 000.)"""
import pytest


def _has_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        return False


pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not _has_gpu(), reason="needs an MI355X")]


def test_headline_configuration_is_bit_exact_with_both_oracles(built_libs):
    import torch

    from bench import parity_check
    from leann_amd.encoder import BertEncoder, config_for
    from leann_amd.gpu_graph_build import build_graph_gpu
    from leann_amd.index import Mi355xIndex
    from leann_amd.recompute import RecomputeProvider
    from leann_amd.synth import CorpusSpec, SyntheticCorpus
    from leann_amd.token_store import TokenStore

    n, dev = 1_000_000, torch.device("cuda")
    corpus = SyntheticCorpus(CorpusSpec(n_chunks=n, seed=1234))
    tok, off = corpus.chunks()
    tokens = TokenStore(tok, off)
    cfg = config_for("sentence-transformers/all-MiniLM-L6-v2")
    enc = BertEncoder.random_init(cfg, 0).to(dev, dtype=torch.float16).eval()
    provider = RecomputeProvider(enc, tokens, 384, dev)
    X = torch.empty((n, 384), dtype=torch.float32, device=dev)
    for b0 in range(0, n, 32768):
        ids = torch.arange(b0, min(n, b0 + 32768), dtype=torch.int32, device=dev)
        X[b0 : b0 + ids.shape[0]] = provider.embed_ids(ids)
    g = build_graph_gpu(X, "mips", M=32, ef_construction=200)
    idx = Mi355xIndex.from_csr(g)
    idx.set_stream(torch.cuda.current_stream().cuda_stream)
    idx.attach_table(X)
    idx.set_provider(provider)
    qt, qo, _ = corpus.queries(272, seed=4321)
    Q = RecomputeProvider(enc, TokenStore(qt, qo), 384, dev).embed_ids(torch.arange(272, dtype=torch.int32, device=dev)).contiguous()
    r = parity_check(idx, g, X, Q, provider, ef=64, beam=1, dim=384, n_table=256, n_recompute=16)
    assert r["n"] == 256 and r["ids_exact"] and r["max_abs_dist"] == 0.0 and r["ndis_equal"], r
    assert r["faiss_transcription_agrees"], r
    rc = r["recompute"]
    assert rc["n"] == 16 and rc["same_ids_requested_every_round"] and rc["ids_exact"] and rc["max_abs_dist"] == 0.0, r
    # north_star tolerance, stated explicitly: ids exact, distances within 1e-4 (here: exactly equal)
    assert r["max_abs_dist"] <= 1e-4 and rc["max_abs_dist"] <= 1e-4
