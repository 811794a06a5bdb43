"""The set-semantics oracle of the DiskANN-style path (oracle/lm_oracle_pq.c, the form the HIP kernels match bit for bit) against the
second, independent one: a literal transcription of upstream DiskANN's PQFlashIndex::cached_beam_search / NeighborPriorityQueue
(oracle/lm_oracle_diskann.c).  Two differently shaped programs -- sorted list + "evaluate the round, then insert" + lock-step over the
batch vs a sorted array with expanded flags and a cursor, neighbours inserted one by one, one query at a time -- must agree on ids,
distances, the number of expansions and the number of PQ evaluations on tie-free inputs when both rank the FINAL candidate list (the
product's deferred fetch: packages/leann-backend-diskann/leann_backend_diskann/diskann_backend.py:444-449, 453-467).  Ranking
upstream's full_retset (every expanded node) instead can only improve the exact distances; the last test quantifies how often it
changes the answer at the benchmark's operating points ("oracle unpinned")."""
import numpy as np
import pytest

from leann_amd.csr_format import METRIC_INNER_PRODUCT, METRIC_L2
from leann_amd.hnsw_builder import build_hnsw
from oracle import oracle as orc
from tests.util import clustered, oracle_graph, queries_near, recall_at_k


def _case(n, d, metric, seed, m=8, M=10, nq=24):
    import torch

    from leann_amd.pq import encode_pq, train_pq

    x = clustered(n, d, seed)
    if metric == "mips":
        x /= np.linalg.norm(x, axis=1, keepdims=True)
    g = build_hnsw(x, metric, M=M, ef_construction=40)
    cb = train_pq(torch.from_numpy(x), m, iters=5, seed=seed).numpy()
    codes = encode_pq(torch.from_numpy(x), torch.from_numpy(cb)).numpy()
    return x, g, cb, codes, queries_near(x, nq, seed + 1)


@pytest.mark.parametrize("metric", ["l2", "mips"])
@pytest.mark.parametrize("k,L,W", [(1, 1, 1), (5, 16, 1), (10, 64, 4), (10, 64, 64), (10, 10, 2), (3, 200, 8), (20, 5, 3)])
def test_set_oracle_equals_diskann_transcription(built_libs, metric, k, L, W):
    x, g, cb, codes, q = _case(3000, 48, metric, seed=7 * k + L + W)
    og = oracle_graph(g, 48)
    ia, da, sa = orc.pq_search(og, cb, codes, q, k, L=L, W=W, table=x)
    ib, db, sb = orc.diskann_search(og, cb, codes, q, k, L=L, W=W, table=x, rerank_final_list_only=True)
    assert np.array_equal(ia, ib)
    assert np.array_equal(da, db)  # same canonical LUT / ADC / exact-distance routines => identical bits, identical order
    assert sa["n_expand"] == sb["n_expanded"]
    assert sa["n_adc"] == sb["n_cmps"] + q.shape[0]  # upstream does not count the medoid's evaluation
    assert sa["n_rounds"] == sb["max_hops"] + 1  # (the set oracle counts the terminating round)


def test_transcription_on_unequal_pq_chunks_and_edge_shapes(built_libs):
    """The public pq_pivots chunking (unequal chunk lengths, a zero-length chunk) and degenerate shapes: one node, k > N."""
    rng = np.random.default_rng(3)
    x, g, _, _, q = _case(800, 24, "l2", seed=5, m=4)
    og = oracle_graph(g, 24)
    co = np.array([0, 5, 5, 13, 24], np.int32)  # lengths 5, 0, 8, 11
    cb = rng.standard_normal(256 * 24).astype(np.float32)
    codes = rng.integers(0, 256, (800, 4)).astype(np.uint8)
    ia, da, sa = orc.pq_search(og, cb, codes, q, 5, L=32, W=4, table=x, chunk_off=co)
    ib, db, sb = orc.diskann_search(og, cb, codes, q, 5, L=32, W=4, table=x, chunk_off=co)
    assert np.array_equal(ia, ib) and np.array_equal(da, db) and sa["n_expand"] == sb["n_expanded"]
    x1 = rng.standard_normal((1, 16)).astype(np.float32)
    g1 = build_hnsw(x1, "l2", M=4, ef_construction=8)
    cb1 = rng.standard_normal((2, 256, 8)).astype(np.float32)
    c1 = np.zeros((1, 2), np.uint8)
    i1, d1, _ = orc.diskann_search(oracle_graph(g1, 16), cb1, c1, x1, 3, L=4, W=2, table=x1)
    j1, e1, _ = orc.pq_search(oracle_graph(g1, 16), cb1, c1, x1, 3, L=4, W=2, table=x1)
    assert i1.tolist() == [[0, -1, -1]] and np.array_equal(i1, j1) and np.array_equal(d1, e1)


@pytest.mark.parametrize("metric", ["l2", "mips"])
def test_full_retset_ranking_is_never_worse_and_rarely_different(built_libs, metric):
    """Upstream ranks every expanded node (full_retset); the product ranks the final candidate list (a subset of it).  Per query and
    rank the exact distance of upstream's answer is <= the product's; at L >= 64 the two answers coincide for almost every query
    (the nodes a traversal pushes out of its list are the far ones), and recall against brute force differs by well under a point --
    which is why the cheaper set (<= L embeddings to recompute per query instead of one per expansion) is the product's choice."""
    x, g, cb, codes, q = _case(6000, 64, metric, seed=21, m=16, M=12, nq=80)
    og = oracle_graph(g, 64)
    gt, _ = orc.bruteforce_topk(x, q, 10, METRIC_L2 if metric == "l2" else METRIC_INNER_PRODUCT)
    for L, W in ((16, 1), (64, 4), (128, 8)):
        il, dl, sl = orc.diskann_search(og, cb, codes, q, 10, L=L, W=W, table=x, rerank_final_list_only=True)
        iu, du, su = orc.diskann_search(og, cb, codes, q, 10, L=L, W=W, table=x, rerank_final_list_only=False)
        assert sl["n_expanded"] == su["n_expanded"] and sl["n_cmps"] == su["n_cmps"]  # the traversal is the same program
        better = du <= dl if metric == "l2" else du >= dl  # +IP goes out for inner product
        assert better.all()
        same = float(np.mean(np.all(il == iu, axis=1)))
        r_list, r_full = recall_at_k(il, gt), recall_at_k(iu, gt)
        assert r_full >= r_list - 1e-9
        if L >= 64:
            assert same >= 0.9 and r_full - r_list <= 0.01, (L, same, r_list, r_full)
        # the extra cost of upstream's set, in embeddings to recompute per query (what a deferred fetch would have to encode)
        assert su["n_expanded"] >= q.shape[0] * min(L, 10)
