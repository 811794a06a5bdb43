"""CPU tests of the oracle itself (the oracle is test infrastructure; these pin ITS behaviour).

The faiss boundary is "parity unpinned" (no golden ids/distances in the reference), so
the oracle is pinned by (i) a hand-traced graph, (ii) exact brute force, (iii) the distance
formulas of hnsw_embedding_server.py:195-200, (iv) tie ordering by id.
"""
import numpy as np
import pytest

from leann_amd.csr_format import METRIC_INNER_PRODUCT, METRIC_L2, csr_from_adjacency
from oracle import oracle as orc
from tests.util import clustered, oracle_graph, queries_near, recall_at_k


def _line_graph():
    """8 nodes on a line (1-d coordinates 0..7, embedded in 64-d), level-0 links i<->i+1,
    node 3 is the entry point with 2 levels; level 1 holds {3, 6} linked to each other."""
    n, d = 8, 64
    x = np.zeros((n, d), np.float32)
    x[:, 0] = np.arange(n)
    adj = []
    for i in range(n):
        l0 = [j for j in (i - 1, i + 1) if 0 <= j < n]
        lv = [np.array(l0, np.int32)]
        if i == 3:
            lv.append(np.array([6], np.int32))
        if i == 6:
            lv.append(np.array([3], np.int32))
        adj.append(lv)
    g = csr_from_adjacency(adj, d, METRIC_L2, entry_point=3, M=2)
    return x, g


def test_hand_traced_line_graph():
    x, g = _line_graph()
    g.validate()
    og = oracle_graph(g, 64)
    q = np.zeros((1, 64), np.float32)
    q[0, 0] = 6.4
    # hand trace, ef=2,k=2,W=1: seed d(3)=11.56; level1: nbr 6 d=0.16 -> move; level1 again: nbr 3 worse -> level 0.
    # level0: pool={6}; pop 6 -> new {5(1.96),7(0.36)}; pool={6*,7}; pop 7 -> new {} (6 visited); pool {6*,7*} -> done.
    ids, dist, st = orc.search(og, q, 2, ef=2, beam=1, table=x)
    assert ids.tolist() == [[6, 7]]
    np.testing.assert_allclose(dist[0], [0.16, 0.36], rtol=1e-5)
    assert st["nrounds"] == 5 and st["ndis"] == 1 + 1 + 1 + 2 + 0 and st["nexpand"] == 2
    # ef=1 keeps only the best: pops 6 only, 5 and 7 evaluated, none enters the 1-pool
    ids, dist, st = orc.search(og, q, 1, ef=1, beam=1, table=x)
    assert ids.tolist() == [[6]] and st["nexpand"] == 1
    # ef=4: walks further left
    ids, _, _ = orc.search(og, q, 4, ef=4, beam=1, table=x)
    assert ids.tolist() == [[6, 7, 5, 4]]
    # beam 2 gives the same answer here
    ids2, _, _ = orc.search(og, q, 4, ef=4, beam=2, table=x)
    assert ids2.tolist() == [[6, 7, 5, 4]]


def test_hand_traced_dynamic_batching():
    """batch_size (hnsw_backend.py:163,181,234; paper section 4.2) on the line graph, traced by hand.  q = 6.4, ef = k = 4, W = 1, batch_size = 2:
    rounds 1-3 as above (seed 3, move to 6, level down); pool = {6}, pop 6.
    round 4: list of 6 -> new [5, 7]: 2 >= batch, no extra pop; pool {6*, 7 (0.36), 5 (1.96)}; pop 7.
    round 5: list of 7 = {6}: visited -> new []: 0 < 2 -> extra pop 5 (the pool as the round found it) -> list {4, 6} -> new [4]: 1 < 2 -> nothing
             unexpanded left; pool {6*, 7*, 5*, 4 (5.76)}; pop 4.
    round 6: list of 4 = {3, 5}: 3 was only evaluated on the way down (faiss marks level-0 visits only) -> new [3]; no candidate left;
             d(3) = 11.56 does not enter the full pool; nothing unexpanded -> done.
    6 rounds instead of 7, the same 7 evaluations and 4 expansions, the same answer."""
    x, g = _line_graph()
    og = oracle_graph(g, 64)
    q = np.zeros((1, 64), np.float32)
    q[0, 0] = 6.4
    ids0, d0, st0 = orc.search(og, q, 4, ef=4, beam=1, table=x)
    ids2, d2, st2 = orc.search(og, q, 4, ef=4, beam=1, table=x, batch_size=2)
    assert ids0.tolist() == ids2.tolist() == [[6, 7, 5, 4]] and np.array_equal(d0, d2)
    assert (st0["nrounds"], st0["ndis"], st0["nexpand"]) == (7, 7, 4)
    assert (st2["nrounds"], st2["ndis"], st2["nexpand"]) == (6, 7, 4)
    asked = []
    orc.search(og, q, 4, ef=4, beam=1, provider=lambda idv: (asked.append(idv.tolist()), x[idv])[1], batch_size=2)
    assert asked == [[3], [6], [3], [5, 7], [4], [3]]
    # a batch larger than anything reachable: everything the pool offers is popped in one round per frontier
    _, _, st9 = orc.search(og, q, 4, ef=4, beam=1, table=x, batch_size=99)
    assert st9["nrounds"] <= 6 and st9["ndis"] == 7
    # batch_size = 0 is the plain search, bit for bit, on a real graph; a batching search evaluates a few more nodes in far fewer rounds
    from leann_amd.hnsw_builder import build_hnsw

    xs = clustered(3000, 64, 6)
    qs = queries_near(xs, 1, 7)
    gs = build_hnsw(xs, "mips", M=12, ef_construction=60)
    ogs = oracle_graph(gs, 64)
    a = orc.search(ogs, qs, 10, ef=48, table=xs)
    b = orc.search(ogs, qs, 10, ef=48, table=xs, batch_size=0)
    c = orc.search(ogs, qs, 10, ef=48, table=xs, batch_size=48)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    assert c[2]["nrounds"] * 2 <= a[2]["nrounds"] and c[2]["ndis"] <= 1.25 * a[2]["ndis"]


def test_distance_formulas_match_reference_server():
    rng = np.random.default_rng(0)
    for d in (64, 100, 384, 768):
        e, q = rng.standard_normal(d).astype(np.float32), rng.standard_normal(d).astype(np.float32)
        # hnsw_embedding_server.py:195-200
        l2 = float(np.sum(np.square(e.astype(np.float64) - q)))
        ip = float(-np.dot(e.astype(np.float64), q))
        assert abs(orc.dist(e, q, METRIC_L2) - l2) <= 1e-4 * max(1.0, abs(l2))
        assert abs(orc.dist(e, q, METRIC_INNER_PRODUCT) - ip) <= 1e-4 * max(1.0, abs(ip))


@pytest.mark.parametrize("metric,mt", [("mips", METRIC_INNER_PRODUCT), ("l2", METRIC_L2)])
def test_recall_against_bruteforce(built_libs, metric, mt):
    from leann_amd.hnsw_builder import build_hnsw

    x = clustered(8000, 64, 1)
    q = queries_near(x, 100, 2)
    g = build_hnsw(x, metric, M=16, ef_construction=100)
    g.validate()
    og = oracle_graph(g, 64)
    gt, gd = orc.bruteforce_topk(x, q, 10, mt)
    # brute force agrees with numpy
    ref = np.argsort(-(q @ x.T) if mt == 0 else ((q[:, None, :] - x[None, :, :]) ** 2).sum(-1), axis=1, kind="stable")[:, :10]
    assert recall_at_k(gt, ref) > 0.999
    prev = 0.0
    for ef in (10, 40, 160):
        ids, dist, _ = orc.search(og, q, 10, ef=ef, beam=1, table=x)
        r = recall_at_k(ids, gt)
        assert r >= prev - 1e-9
        prev = r
        # best-first ordering of the contract (tests/test_diskann_partition.py:216-220 in the reference)
        assert np.all(np.diff(dist, axis=1) <= 0) if mt == 0 else np.all(np.diff(dist, axis=1) >= 0)
    assert prev >= 0.97


def test_provider_mode_equals_table_mode(built_libs):
    from leann_amd.hnsw_builder import build_hnsw

    x = clustered(3000, 100, 3)  # D not a multiple of 64 -> padding path
    q = queries_near(x, 30, 4)
    g = build_hnsw(x, "mips", M=12, ef_construction=60)
    og = oracle_graph(g, 100)
    calls = []

    def provider(ids):
        assert np.all(np.diff(ids) > 0), "provider must receive sorted unique ids"
        calls.append(len(ids))
        return x[ids]

    a = orc.search(og, q, 10, ef=48, beam=3, table=x)
    b = orc.search(og, q, 10, ef=48, beam=3, provider=provider)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert b[2]["nunique"] == sum(calls) <= b[2]["ndis"]
    # per-call memo (the product's default for a call of more than one query): every node requested at most once, same results
    per_round = sum(calls)
    calls.clear()
    fetched = []

    def logging_provider(ids):
        fetched.append(ids.copy())
        return provider(ids)

    c = orc.search(og, q, 10, ef=48, beam=3, provider=logging_provider, memo=True)
    allids = np.concatenate(fetched)
    assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1]) and c[2]["ndis"] == b[2]["ndis"]
    assert np.unique(allids).shape[0] == allids.shape[0] == c[2]["nunique"] < per_round


def test_ties_order_by_id(built_libs):
    from leann_amd.hnsw_builder import build_hnsw

    x = clustered(600, 64, 5, normalize=False)
    x[300:] = x[:300]
    g = build_hnsw(x, "l2", M=8, ef_construction=40)
    og = oracle_graph(g, 64)
    ids, dist, _ = orc.search(og, x[:20].copy(), 2, ef=32, beam=1, table=x)
    for i in range(20):
        assert ids[i].tolist() == [i, i + 300] and dist[i, 0] == 0.0 and dist[i, 1] == 0.0


def test_empty_small_and_unfilled():
    x, g = _line_graph()
    og = oracle_graph(g, 64)
    ids, dist, _ = orc.search(og, x[:2], 12, ef=16, beam=1, table=x)
    assert (ids[:, 8:] == -1).all() and np.isinf(dist[:, 8:]).all() and (ids[:, :8] >= 0).all()
    e = csr_from_adjacency([], 64, METRIC_INNER_PRODUCT, entry_point=-1)
    oe = orc.OracleGraph(e.node_offsets, e.level_ptr, e.neighbors, e.levels, -1, -1, 0, 64)
    ids, dist, _ = orc.search(oe, x[:2], 3, ef=4, table=np.zeros((0, 64), np.float32))
    assert (ids == -1).all() and (dist == -np.inf).all()


def test_check_relative_distance_off_caps_expansions(built_libs):
    from leann_amd.hnsw_builder import build_hnsw

    x = clustered(3000, 64, 6)
    q = queries_near(x, 10, 7)
    g = build_hnsw(x, "mips", M=12, ef_construction=60)
    og = oracle_graph(g, 64)
    _, _, st = orc.search(og, q, 5, ef=8, beam=1, check_relative_distance=False, table=x)
    assert st["nexpand"] <= 10 * (8 + 1)


def test_merge_topk():
    rng = np.random.default_rng(8)
    S, B, k = 4, 9, 5
    ids = rng.integers(0, 1000, (S, B, k)).astype(np.int64)
    d = np.sort(rng.standard_normal((S, B, k)).astype(np.float32), axis=2)
    oi, od = orc.merge_topk(ids, d, METRIC_L2)
    for b in range(B):
        allp = sorted(zip(d[:, b].ravel().tolist(), ids[:, b].ravel().tolist()))[:k]
        assert [p[1] for p in allp] == oi[b].tolist()


def test_two_level_search_oracle_semantics(built_libs):
    """prune_ratio (paper Alg. 2): fewer exact evaluations, recall stays high for the global strategy; prune_ratio 0
    or no PQ == plain search; the three strategies order as documented (local prunes hardest per hop)."""
    import torch

    from leann_amd.hnsw_builder import build_hnsw
    from leann_amd.pq import encode_pq, train_pq

    x = clustered(6000, 64, 11, n_centers=64, sigma=0.5)
    q = queries_near(x, 60, 12)
    g = build_hnsw(x, "mips", M=12, ef_construction=60)
    og = oracle_graph(g, 64)
    cb = train_pq(torch.from_numpy(x), 16, iters=6, seed=0).numpy()
    codes = encode_pq(torch.from_numpy(x), torch.from_numpy(cb)).numpy()
    gt, _ = orc.bruteforce_topk(x, q, 10, METRIC_INNER_PRODUCT)
    base_i, base_d, base = orc.search(og, q, 10, ef=64, table=x)
    same_i, same_d, st0 = orc.search(og, q, 10, ef=64, table=x, prune_ratio=0.0, pq=(cb, codes))
    assert np.array_equal(base_i, same_i) and np.array_equal(base_d, same_d) and st0["nadc"] == 0
    res = {}
    for strat in ("global", "local", "proportional"):
        ids, _, st = orc.search(og, q, 10, ef=64, table=x, prune_ratio=0.5, pruning_strategy=strat, pq=(cb, codes))
        res[strat] = (recall_at_k(ids, gt), st["ndis"], st["nadc"])
        assert st["ndis"] < 0.75 * base["ndis"] and st["nadc"] > 0
    assert res["global"][0] >= 0.97 and res["proportional"][0] >= 0.97
    assert res["local"][0] <= res["global"][0] + 1e-9
    # provider mode sees only the surviving nodes
    seen = []
    orc.search(og, q[:5], 10, ef=32, provider=lambda idv: (seen.append(len(idv)), x[idv])[1], prune_ratio=0.6, pq=(cb, codes))
    _, _, stp = orc.search(og, q[:5], 10, ef=32, table=x, prune_ratio=0.6, pq=(cb, codes))
    assert sum(seen) <= stp["ndis"]
