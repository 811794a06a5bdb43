"""The set-semantics oracle (oracle/lm_oracle.c, the form the HIP kernels match bit for bit) against the second,
independent oracle: a literal heap-based transcription of upstream faiss' HNSW::search / search_from_candidates /
MinimaxHeap / result heap (oracle/lm_oracle_faiss.c).  Two differently shaped programs -- sorted pool + lock-step
rounds over the batch vs binary heaps with lazily deleted slots, one query at a time -- must agree on ids,
distances, the number of distance evaluations and the number of expansions, for beam_size 1 on tie-free inputs
(VERDICT r1 "next round" 2c)."""
import numpy as np
import pytest

from leann_amd.csr_format import METRIC_INNER_PRODUCT, METRIC_L2
from leann_amd.hnsw_builder import build_hnsw
from oracle import oracle as orc
from tests.util import clustered, oracle_graph, queries_near


def _case(n, d, metric, seed, M=8):
    x = clustered(n, d, seed)
    if metric == "mips":
        x /= np.linalg.norm(x, axis=1, keepdims=True)
    g = build_hnsw(x, metric, M=M, ef_construction=40)
    return x, g, queries_near(x, 24, seed + 1)


def _both(og, x, q, k, ef, check=True):
    a = orc.search(og, q, k, ef=ef, beam=1, check_relative_distance=check, table=x)
    b = orc.faiss_search(og, q, k, ef=ef, check_relative_distance=check, table=x)
    return a, b


def _upper_evals(st_set, st_faiss, B):
    """The set oracle counts every (query, node) evaluation in ndis; the transcription separates level 0."""
    return st_set["ndis"] - st_faiss["ndis"]


@pytest.mark.parametrize("metric", ["l2", "mips"])
@pytest.mark.parametrize("k,ef", [(1, 1), (1, 8), (5, 16), (10, 64), (10, 10), (3, 200)])
def test_set_oracle_equals_faiss_transcription(metric, k, ef):
    x, g, q = _case(3000, 48, metric, seed=k * 100 + ef)
    og = oracle_graph(g, 48)
    (ia, da, sa), (ib, db, sb) = _both(og, x, q, k, ef)
    assert np.array_equal(ia, ib)
    assert np.array_equal(da, db)  # same canonical distance routine => identical bits, identical order
    assert sa["nexpand"] == sb["nstep"]
    # level-0 evaluations: the set oracle's total minus the descent's (seed + upper-level neighbours + entry point)
    assert sa["ndis"] == sb["ndis"] + sb["ndis_upper"]


@pytest.mark.parametrize("metric", ["l2", "mips"])
@pytest.mark.parametrize("k,ef", [(10, 4), (20, 5), (50, 1), (7, 6)])
def test_k_larger_than_efsearch_uses_the_count_below_stop(metric, k, ef):
    """capacity = max(efSearch, k) but the relative-distance stop still counts against efSearch
    (search_from_candidates: count_below(d0) >= efSearch -> break): only reachable when k > efSearch."""
    x, g, q = _case(2500, 32, metric, seed=k + ef)
    og = oracle_graph(g, 32)
    (ia, da, sa), (ib, db, sb) = _both(og, x, q, k, ef)
    assert np.array_equal(ia, ib) and np.array_equal(da, db)
    assert sa["nexpand"] == sb["nstep"] and sa["ndis"] == sb["ndis"] + sb["ndis_upper"]


@pytest.mark.parametrize("k,ef", [(5, 8), (10, 32), (10, 4)])
def test_without_relative_distance_check_the_step_cap_is_efsearch_plus_one(k, ef):
    """hnsw_backend.py:209-217 switches check_relative_distance off for OpenAI-cosine models: faiss then stops after
    nstep > efSearch."""
    x, g, q = _case(4000, 32, "l2", seed=7 * k + ef, M=4)
    og = oracle_graph(g, 32)
    (ia, da, sa), (ib, db, sb) = _both(og, x, q, k, ef, check=False)
    assert np.array_equal(ia, ib) and np.array_equal(da, db)
    assert sa["nexpand"] == sb["nstep"] <= q.shape[0] * (ef + 1)
    assert sa["ndis"] == sb["ndis"] + sb["ndis_upper"]


def test_degenerate_graphs():
    """single node, unreachable nodes, empty neighbour lists, k > N."""
    from leann_amd.csr_format import csr_from_adjacency

    d = 64
    rng = np.random.default_rng(3)
    x = rng.standard_normal((6, d)).astype(np.float32)
    adj = [[np.array([1, 2], np.int32)], [np.array([0], np.int32)], [np.array([], np.int32)], [np.array([4], np.int32)],
           [np.array([3], np.int32)], [np.array([], np.int32)]]
    g = csr_from_adjacency(adj, d, METRIC_L2, entry_point=0, M=2)
    og = oracle_graph(g, d)
    q = rng.standard_normal((5, d)).astype(np.float32)
    for k, ef in ((4, 4), (8, 2), (2, 16)):
        (ia, da, _), (ib, db, _) = _both(og, x, q, k, ef)
        assert np.array_equal(ia, ib) and np.array_equal(da, db)
        assert set(ia[0][ia[0] >= 0].tolist()) <= {0, 1, 2}  # 3, 4, 5 are unreachable
    g1 = csr_from_adjacency([[np.array([], np.int32)]], d, METRIC_INNER_PRODUCT, entry_point=0, M=2)
    (ia, da, _), (ib, db, _) = _both(oracle_graph(g1, d), x[:1], q, 3, 8)
    assert np.array_equal(ia, ib) and np.array_equal(da, db) and ia[0].tolist() == [0, -1, -1]
    assert np.isneginf(da[0, 1:]).all()
