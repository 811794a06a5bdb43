"""BASELINE.json configs[0] (Pride & Prejudice, MiniLM-L6 shape, HNSW, top-10) against the committed
golden fixture tests/golden/c1_pp.* (made by tests/golden/make_golden_c1.py in the dev container)."""
from pathlib import Path

import numpy as np
import pytest

G = Path(__file__).resolve().parent / "golden"


def _load():
    from leann_amd import csr_format as cf

    z = np.load(G / "c1_pp.npz")
    g = cf.read_index(G / "c1_pp.index")
    return z, g


def _embed_cpu(tok, off):
    import torch

    from leann_amd.encoder import BertEncoder, config_for
    from leann_amd.synth import pad_batch

    enc = BertEncoder.random_init(config_for("all-MiniLM-L6-v2"), seed=0)
    ids, lens = pad_batch(tok, off, 256)
    with torch.no_grad():
        return enc.encode_tokens(torch.from_numpy(ids), torch.from_numpy(lens), batch_size=64).numpy()


def test_oracle_reproduces_golden_c1(built_libs):
    """CPU: re-embed the committed tokens with the fp32 CPU encoder, run the oracle on the committed
    graph -> the committed top-10 (ids exact, distances 1e-5)."""
    from oracle import oracle as orc
    from tests.util import oracle_graph

    z, g = _load()
    assert g.ntotal == 1018 and g.d == 384 and g.is_pruned
    X = _embed_cpu(z["chunk_tok"], z["chunk_off"])
    Q = _embed_cpu(z["query_tok"], z["query_off"])
    assert np.allclose(Q, z["query_emb"], atol=1e-5)
    ids, dist, _ = orc.search(oracle_graph(g, 384), z["query_emb"], 10, ef=64, beam=1, table=X)
    same = (ids == z["oracle_ids"]).mean()
    assert same >= 0.97, same  # identical unless the platform's BLAS rounds differently
    assert np.allclose(np.sort(dist, 1), np.sort(z["oracle_dist"], 1), atol=1e-4)


@pytest.mark.gpu
def test_gpu_recompute_search_matches_golden_c1():
    """GPU: native reader -> HBM token store -> fp32 encoder on the GPU -> recompute-mode beam search.
    ids match the CPU golden top-10 (embeddings differ by ~1e-6 CPU<->GPU, so near-ties may swap:
    >= 95 % identical positions, set overlap >= 0.97) and distances agree within the north_star 1e-4."""
    import torch

    from leann_amd import _lib
    from leann_amd.encoder import BertEncoder, config_for
    from leann_amd.index import Mi355xIndex
    from leann_amd.recompute import RecomputeProvider
    from leann_amd.token_store import TokenStore

    _lib.require_gpu()
    z, g = _load()
    idx = Mi355xIndex.read(str(G / "c1_pp.index"))
    assert idx.info.ntotal == 1018 and not idx.info.has_table
    enc = BertEncoder.random_init(config_for("all-MiniLM-L6-v2"), seed=0).to("cuda")
    ts = TokenStore(z["chunk_tok"], z["chunk_off"])
    prov = RecomputeProvider(enc, ts, 384, torch.device("cuda"), batch_size=256)
    idx.set_stream(torch.cuda.current_stream().cuda_stream)
    idx.set_provider(prov)
    dist, ids = idx.search(z["query_emb"], 10, idx.make_params(ef=64, beam=1, recompute=True))
    gold_i, gold_d = z["oracle_ids"], z["oracle_dist"]
    assert (ids == gold_i).mean() >= 0.95
    overlap = np.mean([len(set(ids[i]) & set(gold_i[i])) / 10 for i in range(ids.shape[0])])
    assert overlap >= 0.97
    for i in range(ids.shape[0]):
        gd = dict(zip(gold_i[i].tolist(), gold_d[i].tolist()))
        for j, v in enumerate(ids[i].tolist()):
            if v in gd:
                assert abs(dist[i, j] - gd[v]) <= 1e-4
    assert np.all(np.diff(dist, axis=1) <= 1e-7)
