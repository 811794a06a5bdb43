"""Property tests of the oracle on random small graphs (hypothesis): with a pool as large as the graph, best-first /
beam search must return the EXACT top-k of the nodes reachable from the entry point, independent of beam width,
batching and provider/table mode -- a size-independent invariant of the algorithm both implementations follow."""
import numpy as np
from hypothesis import given, settings
from hypothesis import strategies as st

from leann_amd.csr_format import METRIC_INNER_PRODUCT, METRIC_L2, csr_from_adjacency
from oracle import oracle as orc
from tests.util import oracle_graph


def _random_graph(rng, n, d, metric):
    x = rng.standard_normal((n, d)).astype(np.float32)
    adj = []
    for i in range(n):
        deg = int(rng.integers(0, min(5, n)))
        nb = rng.choice([j for j in range(n) if j != i], size=min(deg, n - 1), replace=False) if n > 1 else np.zeros(0, int)
        adj.append([np.asarray(nb, np.int32)])
    # one extra level over a random subset containing the entry point
    ep = int(rng.integers(0, n))
    upper = sorted({ep, *rng.choice(n, size=min(n, 3), replace=False).tolist()})
    for u in upper:
        others = [v for v in upper if v != u]
        adj[u].append(np.asarray(others[: int(rng.integers(0, len(others) + 1))], np.int32))
    g = csr_from_adjacency(adj, d, metric, entry_point=ep, M=4)
    g.max_level = 1
    return x, g, ep


def _reachable(g, start_nodes):
    seen, stack = set(start_nodes), list(start_nodes)
    while stack:
        u = stack.pop()
        for v in g.neighbors_of(u, 0).tolist():
            if v not in seen:
                seen.add(v)
                stack.append(v)
    return seen


@settings(max_examples=60, deadline=None)
@given(seed=st.integers(0, 10**6), n=st.integers(1, 24), metric=st.sampled_from([METRIC_INNER_PRODUCT, METRIC_L2]),
       beam=st.integers(1, 5), k=st.integers(1, 6), batch=st.sampled_from([0, 0, 1, 3, 8, 40]))
def test_exhaustive_pool_gives_exact_topk_of_reachable_set(seed, n, metric, beam, k, batch):
    rng = np.random.default_rng(seed)
    x, g, ep = _random_graph(rng, n, 64, metric)
    g.validate()
    og = oracle_graph(g, 64)
    q = rng.standard_normal((3, 64)).astype(np.float32)
    # (any dynamic-batching target: it changes the ORDER nodes are expanded in, never the set a pool >= N ends with)
    ids, dist, stats = orc.search(og, q, k, ef=n + 4, beam=beam, table=x, batch_size=batch)
    ids_p, dist_p, _ = orc.search(og, q, k, ef=n + 4, beam=beam, provider=lambda idv: x[idv], batch_size=batch)
    assert np.array_equal(ids, ids_p) and np.array_equal(dist, dist_p)
    for qi in range(3):
        # where the greedy descent lands on level 0: emulate it with exact distances
        def dd(v):
            return orc.dist(x[v], q[qi], metric)

        cur = ep
        while True:
            nb = g.neighbors_of(cur, 1).tolist()
            best = min(nb, key=lambda v: (dd(v), v), default=None)
            if best is not None and (dd(best), best) < (dd(cur), cur):
                cur = best
            else:
                break
        reach = sorted(_reachable(g, [cur]), key=lambda v: (dd(v), v))[:k]
        got = [v for v in ids[qi].tolist() if v >= 0]
        assert got == reach, (got, reach)
        exp = np.array([dd(v) if metric == METRIC_L2 else -dd(v) for v in reach], np.float32)
        assert np.array_equal(dist[qi, : len(reach)], exp)
        assert (ids[qi, len(reach):] == -1).all()


@settings(max_examples=40, deadline=None)
@given(seed=st.integers(0, 10**6), n=st.integers(2, 20), beam=st.integers(1, 4))
def test_pq_path_with_exhaustive_list_and_rerank_is_exact(seed, n, beam):
    """DiskANN-style oracle: list size >= N => every reachable node is ranked; the exact rerank then returns the
    exact top-k of the reachable set, whatever the (arbitrarily bad) PQ codes are."""
    rng = np.random.default_rng(seed)
    x, g, ep = _random_graph(rng, n, 64, METRIC_L2)
    og = oracle_graph(g, 64)
    m = 16
    cb = rng.standard_normal((m, 256, 4)).astype(np.float32)
    codes = rng.integers(0, 256, (n, m)).astype(np.uint8)
    q = rng.standard_normal((2, 64)).astype(np.float32)
    ids, dist, _ = orc.pq_search(og, cb, codes, q, min(3, n), L=n + 2, W=beam, table=x)
    reach_all = _reachable(g, [ep])
    for qi in range(2):
        exp = sorted(reach_all, key=lambda v: (orc.dist(x[v], q[qi], METRIC_L2), v))[: min(3, n)]
        assert [v for v in ids[qi].tolist() if v >= 0] == exp
