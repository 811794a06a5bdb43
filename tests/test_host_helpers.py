"""CPU tests of host-side helpers: synthetic corpus, hub selection, PQ training/encoding, flat graph,
and the batched graph builder driven by the oracle search (the GPU default is injected away)."""
import numpy as np
import torch

from tests.util import clustered, oracle_graph, queries_near, recall_at_k


def test_synthetic_corpus_is_deterministic_and_structured():
    from leann_amd.synth import CLS_ID, SEP_ID, CorpusSpec, SyntheticCorpus, pad_batch

    spec = CorpusSpec(n_chunks=3000, n_topics=12)
    a_tok, a_off = SyntheticCorpus(spec).chunks(block=1024)
    b_tok, b_off = SyntheticCorpus(spec).chunks(block=1024)
    assert np.array_equal(a_tok, b_tok) and np.array_equal(a_off, b_off)
    lens = np.diff(a_off.astype(np.int64))
    assert lens.min() >= 16 and lens.max() <= 256 and 150 < lens.mean() < 210
    assert (a_tok[a_off[:-1].astype(np.int64)] == CLS_ID).all() and (a_tok[a_off[1:].astype(np.int64) - 1] == SEP_ID).all()
    ids, ln = pad_batch(a_tok, a_off, 256)
    assert ids.shape == (3000, 256) and (ids[np.arange(3000), ln - 1] == SEP_ID).all()
    # chunks of one document share vocabulary: Jaccard overlap inside a doc >> across docs
    sets = [set(ids[i, : ln[i]].tolist()) for i in range(64)]
    same = np.mean([len(sets[i] & sets[i + 1]) / len(sets[i] | sets[i + 1]) for i in range(0, 15)])  # doc 0
    diff = np.mean([len(sets[i] & sets[i + 32]) / len(sets[i] | sets[i + 32]) for i in range(0, 15)])  # doc 0 vs doc 2
    assert same > 2 * diff
    q_tok, q_off, docs = SyntheticCorpus(spec).queries(10, seed=7)
    q2 = SyntheticCorpus(spec).queries(10, seed=7)
    assert np.array_equal(q_tok, q2[0]) and docs.shape == (10,)


def test_hub_nodes_by_in_degree(built_libs):
    from leann_amd.backend import hub_nodes
    from leann_amd.hnsw_builder import build_hnsw

    x = clustered(2000, 32, 3)
    g = build_hnsw(x, "mips", M=8, ef_construction=40)
    indeg = np.zeros(2000, np.int64)
    for i in range(2000):
        np.add.at(indeg, g.neighbors_of(i, 0), 1)
    hubs = hub_nodes(g, 0.05)
    assert hubs.shape == (100,) and np.all(np.diff(hubs) > 0)
    assert indeg[hubs].min() >= np.sort(indeg)[-100]  # exactly the top-100 by in-degree (ties by id)
    assert hub_nodes(g, 0.0).shape == (0,) and hub_nodes(g, 1.0).shape == (2000,)


def test_pq_train_encode_and_flat_graph(built_libs):
    from leann_amd.hnsw_builder import build_hnsw
    from leann_amd.pq import encode_pq, flat_graph, train_pq
    from oracle import oracle as orc

    x = clustered(4000, 64, 5, n_centers=32, sigma=0.5)
    cb = train_pq(torch.from_numpy(x), 16, iters=6, seed=0)
    codes = encode_pq(torch.from_numpy(x), cb)
    assert cb.shape == (16, 256, 4) and codes.shape == (4000, 16) and codes.dtype == torch.uint8
    rec = cb[torch.arange(16)[None, :], codes.long()].reshape(4000, 64).numpy()
    assert np.mean((rec - x) ** 2) < 0.35 * np.mean(x**2)  # the quantiser explains most of the variance
    cb2 = train_pq(torch.from_numpy(x), 16, iters=6, seed=0)
    assert torch.equal(cb, cb2)  # deterministic
    g = build_hnsw(x, "l2", M=8, ef_construction=40)
    fg = flat_graph(g, x)
    fg.validate()
    assert fg.max_level == 0 and (fg.levels == 1).all()
    assert all(np.array_equal(fg.neighbors_of(i, 0), g.neighbors_of(i, 0)) for i in (0, 17, 3999))
    mean = x.mean(0)
    assert fg.entry_point == int(np.argmin(((x - mean) ** 2).sum(1)))
    # DiskANN-style oracle search on it reaches high recall after the exact rerank
    q = queries_near(x, 40, 6)
    ids, _, _ = orc.pq_search(oracle_graph(fg, 64), cb.numpy(), codes.numpy(), q, 10, L=96, W=4, table=x)
    gt, _ = orc.bruteforce_topk(x, q, 10, 1)
    assert recall_at_k(ids, gt) > 0.9


def test_batched_graph_builder_with_injected_search(built_libs):
    """gpu_graph_build.build_graph_gpu on CPU tensors with the oracle as candidate search: quality comparable to
    the sequential HNSW builder, all structural invariants hold."""
    from leann_amd.gpu_graph_build import build_graph_gpu
    from oracle import oracle as orc

    def oracle_search_fn(g, table, queries, ef, k):
        ids, dd, _ = orc.search(oracle_graph(g, g.d), queries.numpy(), k, ef=ef, beam=2, table=table.numpy())
        return torch.from_numpy(ids), torch.from_numpy(dd if g.metric_type == 0 else -dd)

    x = clustered(6000, 48, 9, n_centers=60, sigma=0.5)
    g = build_graph_gpu(torch.from_numpy(x), "mips", M=12, ef_construction=60, search_fn=oracle_search_fn, seed_nodes=512)
    g.validate()
    assert g.level0_degrees().max() <= 24 and g.level0_degrees().min() >= 1
    q = queries_near(x, 100, 10)
    gt, _ = orc.bruteforce_topk(x, q, 10, 0)
    ids, _, _ = orc.search(oracle_graph(g, 48), q, 10, ef=64, table=x)
    assert recall_at_k(ids, gt) > 0.97


def test_select_heuristic_selection_form_equals_the_candidate_scan():
    """gpu_graph_build._select_heuristic: the loop over selections (first alive candidate, strike what it dominates) returns the keep
    mask of the candidate-by-candidate scan bit for bit -- inner product and L2, exact ties (duplicate vectors), empty slots in the
    middle and at the tail of a row, m larger / smaller than the number of survivors."""
    from leann_amd import gpu_graph_build as gb
    from leann_amd.csr_format import METRIC_INNER_PRODUCT, METRIC_L2

    gen = torch.Generator().manual_seed(0)
    for metric in (METRIC_INNER_PRODUCT, METRIC_L2):
        for N, D, n, K, m in ((500, 16, 300, 24, 8), (200, 8, 150, 16, 16), (50, 4, 40, 12, 3), (300, 32, 500, 64, 32), (300, 8, 300, 64, 64), (64, 2, 200, 32, 5)):
            xs = torch.randn(N, D, generator=gen)
            nd = N // 7
            xs[:nd] = xs[nd : 2 * nd]  # duplicates: exact ties
            base = torch.randint(0, N, (n,), generator=gen)
            cand = torch.stack([torch.randperm(N, generator=gen)[:K] for _ in range(n)])
            s = (xs[base][:, None, :] * xs[cand]).sum(-1) if metric == METRIC_INNER_PRODUCT else -((xs[base][:, None, :] - xs[cand]) ** 2).sum(-1)
            s, o = torch.sort(s, dim=1, descending=True)
            cand = torch.gather(cand, 1, o)
            empty = (torch.arange(K)[None, :] >= torch.randint(0, K + 1, (n,), generator=gen)[:, None]) | (torch.rand(n, K, generator=gen) < 0.05)
            cand[empty] = -1
            s[empty] = -float("inf")
            a = gb._select_heuristic_scan(xs, cand, s, m, metric, block=128)
            b = gb._select_heuristic_selection(xs, cand, s, m, metric, block=128)
            assert torch.equal(a, b) and int(a.sum(1).max()) <= m, (metric, N, D, n, K, m)
            assert torch.equal(gb._select_heuristic(xs, cand, s, m, metric), a)
            # Vamana-style relaxation (alpha > 1: the strict pass first, then the relaxed rule over what is left while slots remain): both
            # forms agree, the strict pass's neighbours all stay, the cap holds
            xn = torch.nn.functional.normalize(xs, dim=1)
            sn = (xn[base][:, None, :] * xn[cand.clamp(min=0)]).sum(-1) if metric == METRIC_INNER_PRODUCT else s
            if metric == METRIC_INNER_PRODUCT:
                sn[empty] = -float("inf")
                sn, o2 = torch.sort(sn, dim=1, descending=True)
                cn = torch.gather(cand, 1, o2)
            else:
                cn = cand
            base1 = gb._select_heuristic_scan(xn, cn, sn, m, metric, block=128)
            for alpha in (1.2, 1.5):
                ra = gb._select_heuristic_scan(xn, cn, sn, m, metric, block=128, alpha=alpha)
                rb = gb._select_heuristic_selection(xn, cn, sn, m, metric, block=128, alpha=alpha)
                assert torch.equal(ra, rb) and int(ra.sum(1).max()) <= m and bool((ra | base1).eq(ra).all()) and int(ra.sum()) >= int(base1.sum())


def test_packed_weight_caches_follow_the_source_weights():
    """leann_amd/encoder.py: _packed keys every packed-weight copy on (device, storage address, in-place version, dtype) of its
    sources: an in-place update (load_state_dict, optimiser step) or a dtype change must rebuild the pack, an untouched module
    must not."""
    import torch

    from leann_amd.encoder import _packed

    lin = torch.nn.Linear(8, 4)
    calls = []

    def make():
        calls.append(1)
        return lin.weight.detach().clone()

    a = _packed(lin, "_t", (lin.weight, lin.bias), make)
    assert _packed(lin, "_t", (lin.weight, lin.bias), make) is a and len(calls) == 1
    with torch.no_grad():
        lin.weight.add_(1.0)  # in place: same storage, new version
    b = _packed(lin, "_t", (lin.weight, lin.bias), make)
    assert len(calls) == 2 and torch.equal(b, lin.weight)
    lin.load_state_dict({"weight": torch.zeros(4, 8), "bias": torch.zeros(4)})
    assert torch.equal(_packed(lin, "_t", (lin.weight, lin.bias), make), torch.zeros(4, 8)) and len(calls) == 3
    lin.half()  # new storage, new dtype
    assert _packed(lin, "_t", (lin.weight, lin.bias), make).dtype == torch.float16 and len(calls) == 4

def test_high_degree_preserving_pruning_alg3(built_libs):
    """LEANN paper Algorithm 3 (gpu_graph_build.prune_preserving_hubs): fewer links, hubs keep theirs, the graph stays valid and
    searchable (recall within 1 % of the unpruned graph at the same ef), fewer distance evaluations per query."""
    import torch

    from leann_amd.gpu_graph_build import prune_preserving_hubs
    from leann_amd.hnsw_builder import build_hnsw
    from oracle import oracle as orc
    from tests.util import clustered, oracle_graph, queries_near, recall_at_k

    x = clustered(5000, 48, 3)
    g = build_hnsw(x, "mips", M=16, ef_construction=100)
    g2 = prune_preserving_hubs(g, torch.from_numpy(x), M=16, m_low=6, hub_fraction=0.02)
    g2.validate()
    d1, d2 = g.level0_degrees(), g2.level0_degrees()
    assert g2.neighbors.shape[0] < g.neighbors.shape[0] and d2.mean() < d1.mean() and d2.max() <= 32
    indeg = np.bincount(g.neighbors[: int(d1.sum())] if g.max_level == 0 else np.concatenate([g.neighbors_of(i, 0) for i in range(5000)]), minlength=5000)
    hubs = np.argsort(-indeg)[:100]
    assert d2[hubs].mean() > 1.3 * d2.mean()  # the hubs are the well-connected nodes of the pruned graph
    # upper levels untouched
    for i in np.nonzero(g.levels > 1)[0][:50]:
        for l in range(1, int(g.levels[i])):
            assert np.array_equal(g.neighbors_of(int(i), l), g2.neighbors_of(int(i), l))
    q = queries_near(x, 150, 5)
    gt, _ = orc.bruteforce_topk(x, q, 10, 0)
    i1, _, s1 = orc.search(oracle_graph(g, 48), q, 10, ef=64, table=x)
    i2, _, s2 = orc.search(oracle_graph(g2, 48), q, 10, ef=64, table=x)
    assert recall_at_k(i2, gt) >= recall_at_k(i1, gt) - 0.01 and s2["ndis"] < s1["ndis"]
    # the builder applies it on request (host builder below the GPU threshold)
    from leann_amd import csr_format as cf
    from leann_amd.backend import Mi355xBuilder
    import tempfile, pathlib

    with tempfile.TemporaryDirectory() as td:
        Mi355xBuilder(M=16, efConstruction=100, hub_preserving_m=6).build(x, [str(i) for i in range(5000)], str(pathlib.Path(td) / "p.leann"))
        Mi355xBuilder(M=16, efConstruction=100).build(x, [str(i) for i in range(5000)], str(pathlib.Path(td) / "f.leann"))
        assert cf.read_index(pathlib.Path(td) / "p.index").neighbors.shape[0] < cf.read_index(pathlib.Path(td) / "f.index").neighbors.shape[0]


def test_blocked_exact_topk_equals_one_big_matmul():
    """leann_amd/exact.py (the benchmarks' ground truth): the blocked form -- query blocks x corpus blocks, top-k merge -- against torch.topk
    of the whole score matrix, with block sizes forced small enough that both loops and the merge run."""
    import torch

    import leann_amd.exact as ex

    torch.manual_seed(3)
    Q, X = torch.randn(37, 16), torch.randn(1003, 16)
    rv, ri = torch.topk(Q @ X.T, 10, dim=1)
    old = ex.MAX_SCORES
    try:
        for cap, qb in ((37 * 130, 256), (16 * 64, 16), (old, 5)):
            ex.MAX_SCORES = cap
            v, i = ex.exact_topk_ip(Q, X, 10, q_block=qb)
            assert torch.equal(i, ri) and torch.allclose(v, rv)
    finally:
        ex.MAX_SCORES = old
    v, i = ex.exact_topk_ip(Q, X[:4], 10)  # k > n
    assert i.shape == (37, 4) and torch.equal(i, torch.topk(Q @ X[:4].T, 4, dim=1).indices)
