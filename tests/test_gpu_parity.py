"""GPU parity tests proper: HIP path (through the C ABI) vs the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): top-k ids bit-exact under the (distance, id) tie-break; distances
bit-exact too, because the kernel implements the oracle's canonical fp32 reduction order (the
north_star tolerance of 1e-4 is asserted as well, as the contractual bound).
"""
import numpy as np
import pytest

from tests.util import clustered, oracle_graph, queries_near, recall_at_k

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch

    from leann_amd import _lib

    _lib.require_gpu()
    assert torch.cuda.is_available()
    return torch


_CACHE = {}


def _build(n, d, metric, seed, M=16, efc=60, normalize=True):
    from leann_amd.hnsw_builder import build_hnsw

    key = (n, d, metric, seed, M, efc, normalize)
    if key not in _CACHE:
        x = clustered(n, d, seed, normalize=normalize)
        _CACHE[key] = (x, build_hnsw(x, metric, M=M, ef_construction=efc, seed=seed))
    x, g = _CACHE[key]
    return x.copy(), g


def _check(torch, x, g, q, k, ef, beam, mode, check_rel=True, table_dtype=np.float32, variant=0, wave=None, batch_size=0):
    from leann_amd.devmem import as_tensor
    from leann_amd.index import Mi355xIndex
    from oracle import oracle as orc

    d = x.shape[1]
    og = oracle_graph(g, d)
    xt = x.astype(table_dtype)
    oi, od, ost = orc.search(og, q, k, ef=ef, beam=beam, check_relative_distance=check_rel, table=xt.astype(np.float32), batch_size=batch_size)
    idx = Mi355xIndex.from_csr(g, device=0)
    idx.set_stream(torch.cuda.current_stream().cuda_stream)
    idx.set_option("update_variant", variant)
    if wave is not None:
        idx.set_option("persistent_wave", wave)
    if mode == "table":
        idx.attach_table(xt)
        prm = idx.make_params(ef=ef, beam=beam, check_relative_distance=check_rel, recompute=False, batch_size=batch_size)
        gd, gi = idx.search(q, k, prm)
    else:
        dp = idx.info.d_padded
        xdev = torch.zeros((x.shape[0], dp), dtype=torch.float32, device="cuda")
        xdev[:, :d] = torch.from_numpy(xt.astype(np.float32)).cuda()
        keep = {}
        seen = []

        def provider(d_ids, n, stream):
            ids = as_tensor(d_ids, (n,), "int32")
            seen.append(ids.cpu().numpy().copy())
            keep["e"] = xdev.index_select(0, ids.long()).contiguous()
            return keep["e"].data_ptr()

        idx.set_provider(provider)
        prm = idx.make_params(ef=ef, beam=beam, check_relative_distance=check_rel, recompute=True, batch_size=batch_size)
        gdt, git = idx.search_device(torch.from_numpy(q).cuda(), k, prm)
        torch.cuda.synchronize()
        gd, gi = gdt.cpu().numpy(), git.cpu().numpy()
        # provider contract: sorted unique ids every round
        for s in seen:
            assert np.all(np.diff(s) > 0)
        # the library default keeps a per-call memo for a call of more than one query: the oracle restates it (oracle.py: memo=)
        want = []
        _, _, pst = orc.search(og, q, k, ef=ef, beam=beam, check_relative_distance=check_rel, memo=q.shape[0] > 1, batch_size=batch_size,
                               provider=lambda idv: (want.append(idv.copy()), xt.astype(np.float32)[idv])[1])
        assert idx.stats()["nunique"] == pst["nunique"]
        assert len(seen) == len(want) and all(np.array_equal(a, b) for a, b in zip(seen, want)), "the provider was not asked for the oracle's ids, round by round"
        if q.shape[0] > 1:
            allids = np.concatenate(seen)
            assert np.unique(allids).shape[0] == allids.shape[0], "a node reached the provider twice in one call"
    st = idx.stats()
    assert np.array_equal(gi, oi), f"ids differ: {np.argwhere(gi != oi)[:5]}"
    assert np.array_equal(gd.view(np.uint32), od.view(np.uint32)), "distances not bit-exact"
    assert np.allclose(gd, od, atol=1e-4, rtol=0)
    assert st["ndis"] == ost["ndis"] and st["nexpand"] == ost["nexpand"] and st["nrounds"] == ost["nrounds"], (st, ost)
    idx.close()
    return gi


@pytest.mark.parametrize("metric", ["mips", "l2"])
@pytest.mark.parametrize("ef,beam", [(16, 1), (64, 1), (64, 4), (200, 2)])
def test_table_mode_parity(env, metric, ef, beam):
    x, g = _build(3000, 384, metric, seed=1)
    q = queries_near(x, 32, seed=2)
    _check(env, x, g, q, 10, ef, beam, "table")


@pytest.mark.parametrize("mode", ["table", "provider"])
@pytest.mark.parametrize("batch_size,ef,beam,check_rel,nq", [(0, 64, 1, True, 24), (16, 64, 1, True, 24), (64, 64, 1, True, 24), (64, 32, 4, True, 24),
                                                            (32, 40, 1, False, 24), (64, 64, 1, True, 1), (500, 24, 2, True, 3)])
def test_dynamic_batching_parity(env, mode, batch_size, ef, beam, check_rel, nq):
    """lm_search_params.batch_size = SearchParametersHNSW.batch_size (hnsw_backend.py:163,181,234): the paper's dynamic batching (section 4.2),
    k_expand's extra pops, against the oracle's restatement -- ids, distance bits, evaluations, expansions, ROUNDS and (provider mode) the
    request list of every round; batch_size 0 is the plain search.  A one-query call (the real caller's batch, api.py:644-796) included."""
    x, g = _build(4000, 384, "mips", seed=5, M=12)
    q = queries_near(x, nq, seed=31)
    _check(env, x, g, q, 10, ef, beam, mode, check_rel=check_rel, batch_size=batch_size)


def test_dynamic_batching_needs_fewer_rounds(env):
    """What the knob is for: a one-query recompute search at efSearch 64 in a third of the rounds or fewer, recall intact."""
    from leann_amd.index import Mi355xIndex

    torch = env
    x, g = _build(20000, 128, "mips", seed=9, M=16)
    q = queries_near(x, 16, seed=33, noise=0.2)
    idx = Mi355xIndex.from_csr(g, device=0)
    idx.attach_table(x)
    from oracle import oracle as orc

    gt, _ = orc.bruteforce_topk(x, q, 10, 0)
    res = {}
    for bs in (0, 64):
        rounds, labels = 0, []
        for i in range(q.shape[0]):
            _, l = idx.search(q[i : i + 1], 10, idx.make_params(ef=64, recompute=False, batch_size=bs))
            rounds += idx.stats()["nrounds"]
            labels.append(l[0])
        res[bs] = (rounds / q.shape[0], recall_at_k(np.stack(labels), gt))
    assert res[64][0] <= 0.5 * res[0][0], res
    assert res[64][1] >= res[0][1] - 0.02, res
    idx.close()


@pytest.mark.parametrize("metric", ["mips", "l2"])
@pytest.mark.parametrize("table_dtype", [np.float32, np.float16])
def test_wave_per_query_table_mode_parity(env, metric, table_dtype):
    """The 64-thread (one wave per query) form of the persistent stored-embedding search -- the kernel behind the
    ">= 60 % of HBM peak" line and the graph builder's searches -- forced by option, against the oracle: ids, distances,
    ndis / nexpand / nrounds."""
    x, g = _build(3000, 384, metric, seed=1)
    q = queries_near(x, 48, seed=21)
    for ef, beam, cr in ((64, 1, True), (16, 1, True), (64, 4, True), (24, 2, False)):
        _check(env, x, g, q, 10, ef, beam, "table", check_rel=cr, table_dtype=table_dtype, wave=1)
    x, g = _build(1500, 100, metric, seed=5)
    _check(env, x, g, queries_near(x, 16, seed=6), 5, 32, 1, "table", table_dtype=table_dtype, wave=1)


@pytest.mark.parametrize("metric", ["mips", "l2"])
def test_wave_per_query_auto_rule_large_batch(env, metric):
    """B >= 2048 on a low-degree graph: the launcher picks the wave-per-query form by itself (lm_search.hip:
    persist_wave_form); same contract, and the same bits as the forced workgroup form."""
    from leann_amd.index import Mi355xIndex

    x, g = _build(3000, 384, metric, seed=1, M=4, efc=40)
    q = queries_near(x, 2304, seed=22)
    deg0 = float(g.level0_degrees().mean())
    assert deg0 * 1 <= 24.0, deg0  # the auto rule's precondition at beam 1
    gi = _check(env, x, g, q, 10, 64, 1, "table")  # auto
    idx = Mi355xIndex.from_csr(g)
    idx.attach_table(x)
    idx.set_option("persistent_wave", 0)
    d0, l0 = idx.search(q, 10, idx.make_params(ef=64, beam=1, recompute=False))
    idx.set_option("persistent_wave", 1)
    d1, l1 = idx.search(q, 10, idx.make_params(ef=64, beam=1, recompute=False))
    assert np.array_equal(l0, gi) and np.array_equal(l1, gi) and np.array_equal(d0, d1)
    idx.close()


@pytest.mark.parametrize("metric", ["mips", "l2"])
def test_recompute_mode_parity(env, metric):
    x, g = _build(3000, 384, metric, seed=1)
    q = queries_near(x, 32, seed=4)
    _check(env, x, g, q, 10, 64, 2, "provider")


@pytest.mark.parametrize("d", [64, 100, 384, 768, 1024])
def test_dimensions_and_padding(env, d):
    x, g = _build(1500, d, "mips", seed=5)
    q = queries_near(x, 16, seed=6)
    _check(env, x, g, q, 5, 32, 1, "table")
    _check(env, x, g, q, 5, 32, 2, "provider")


@pytest.mark.parametrize("variant", [3, 4])
def test_update_kernel_variants_parity(env, variant):
    """The A/B variants of the update step (1: fused + full bitonic sort, 2: split flat-distance +
    merge kernels) obey the same contract as the default."""
    for metric in ("mips", "l2"):
        x, g = _build(3000, 384, metric, seed=1)
        q = queries_near(x, 32, seed=2)
        _check(env, x, g, q, 10, 64, 4, "table", variant=variant)
        _check(env, x, g, q, 10, 16, 1, "table", variant=variant)
        _check(env, x, g, q, 10, 64, 2, "provider", variant=variant)
    x, g = _build(1500, 100, "mips", seed=5)
    _check(env, x, g, queries_near(x, 16, seed=6), 5, 32, 3, "provider", variant=variant, check_rel=False)


def test_recompute_memo_same_results_fewer_recomputes(env):
    """recompute_memo=1: every node is recomputed at most once per call; ids/distances unchanged."""
    from leann_amd.devmem import as_tensor
    from leann_amd.index import Mi355xIndex
    from oracle import oracle as orc

    torch = env
    x, g = _build(3000, 384, "mips", seed=1)
    q = queries_near(x, 64, seed=12)
    xdev = torch.from_numpy(x).cuda()
    idx = Mi355xIndex.from_csr(g)
    idx.set_stream(torch.cuda.current_stream().cuda_stream)
    keep, seen = {}, []

    def provider(d_ids, n, stream):
        ids = as_tensor(d_ids, (n,), "int32")
        seen.append(ids.cpu().numpy().copy())
        keep["e"] = xdev.index_select(0, ids.long()).contiguous()
        return keep["e"].data_ptr()

    idx.set_provider(provider)
    res = {}
    for memo in (False, True):
        seen.clear()
        d, l = idx.search_device(torch.from_numpy(q).cuda(), 10, idx.make_params(ef=64, beam=2, recompute=True, recompute_memo=memo))
        torch.cuda.synchronize()
        allids = np.concatenate(seen)
        res[memo] = (d.cpu().numpy(), l.cpu().numpy(), idx.stats()["nunique"], allids)
    oi, od, _ = orc.search(oracle_graph(g, 384), q, 10, ef=64, beam=2, table=x)
    for memo in (False, True):
        assert np.array_equal(res[memo][1], oi) and np.array_equal(res[memo][0], od)
    assert res[True][2] < res[False][2]
    assert len(np.unique(res[True][3])) == len(res[True][3])  # nothing recomputed twice within the call
    assert set(res[True][3].tolist()) == set(res[False][3].tolist())
    # the memo starts small and doubles on demand (rows kept across the re-allocations): same results, same provider rows
    idx.set_option("memo_initial_rows", 32)
    seen.clear()
    d, l = idx.search_device(torch.from_numpy(q).cuda(), 10, idx.make_params(ef=64, beam=2, recompute=True))  # default: memo on
    torch.cuda.synchronize()
    assert np.array_equal(l.cpu().numpy(), oi) and np.array_equal(d.cpu().numpy(), od)
    assert idx.stats()["nunique"] == res[True][2] and np.array_equal(np.concatenate(seen), res[True][3])
    idx.set_option("memo_initial_rows", 0)


def test_speculative_prefetch_same_results_fewer_provider_calls(env):
    """Option "speculate" (k_speculate): a one-query search embeds the neighbours of its best unexpanded candidates ahead of time.  Labels,
    distances and evaluation counts are the oracle's for every S; the provider is asked less often, never twice for a node."""
    from leann_amd.devmem import as_tensor
    from leann_amd.index import Mi355xIndex
    from oracle import oracle as orc

    torch = env
    x, g = _build(20000, 128, "mips", seed=5)
    q = queries_near(x, 6, seed=31)
    xdev = torch.from_numpy(x).cuda()
    idx = Mi355xIndex.from_csr(g)
    idx.set_stream(torch.cuda.current_stream().cuda_stream)
    keep, seen = {}, []

    def provider(d_ids, n, stream):
        ids = as_tensor(d_ids, (n,), "int32")
        seen.append(ids.cpu().numpy().copy())
        keep["e"] = xdev.index_select(0, ids.long()).contiguous()
        return keep["e"].data_ptr()

    idx.set_provider(provider)
    og = oracle_graph(g, 128)
    calls = {}
    for S in (0, 4, 16):
        idx.set_option("speculate", S)
        calls[S] = 0
        for i in range(q.shape[0]):
            seen.clear()
            d, l = idx.search_device(torch.from_numpy(q[i : i + 1]).cuda(), 10, idx.make_params(ef=96, beam=1, recompute=True))
            torch.cuda.synchronize()
            oi, od, ost = orc.search(og, q[i : i + 1], 10, ef=96, beam=1, table=x)
            assert np.array_equal(l.cpu().numpy(), oi) and np.array_equal(d.cpu().numpy(), od) and idx.stats()["ndis"] == ost["ndis"], (S, i)
            allids = np.concatenate(seen)
            if S:
                assert len(np.unique(allids)) == len(allids)
            calls[S] += len(seen)
    assert calls[16] < calls[4] <= calls[0], calls
    idx.set_option("speculate", 0)


def test_single_query_direct_same_results(env):
    """Option "single_query_direct": a one-query recompute pass hands its new-list to the provider as it is (no k_uniq_* launches).  Labels,
    distances and counts are the oracle's; the provider sees the same ids per round in discovery order."""
    from leann_amd.devmem import as_tensor
    from leann_amd.index import Mi355xIndex
    from oracle import oracle as orc

    torch = env
    x, g = _build(20000, 128, "l2", seed=9, normalize=False)
    q = queries_near(x, 5, seed=33, normalize=False)
    xdev = torch.from_numpy(x).cuda()
    idx = Mi355xIndex.from_csr(g)
    idx.set_stream(torch.cuda.current_stream().cuda_stream)
    keep, seen = {}, []

    def provider(d_ids, n, stream):
        ids = as_tensor(d_ids, (n,), "int32")
        seen.append(ids.cpu().numpy().copy())
        keep["e"] = xdev.index_select(0, ids.long()).contiguous()
        return keep["e"].data_ptr()

    idx.set_provider(provider)
    og = oracle_graph(g, 128)
    per_round = {}
    for direct in (0, 1):
        idx.set_option("single_query_direct", direct)
        per_round[direct] = []
        for i in range(q.shape[0]):
            seen.clear()
            d, l = idx.search_device(torch.from_numpy(q[i : i + 1]).cuda(), 10, idx.make_params(ef=64, beam=2, recompute=True))
            torch.cuda.synchronize()
            oi, od, ost = orc.search(og, q[i : i + 1], 10, ef=64, beam=2, table=x)
            st = idx.stats()
            assert np.array_equal(l.cpu().numpy(), oi) and np.array_equal(d.cpu().numpy(), od) and st["ndis"] == ost["ndis"], (direct, i)
            per_round[direct].append(([s.copy() for s in seen], int(st["nunique"]), int(st["nrounds"])))
    for a, b in zip(per_round[0], per_round[1]):
        assert a[1:] == b[1:] and len(a[0]) == len(b[0])
        assert all(np.array_equal(u, np.sort(v)) for u, v in zip(a[0], b[0]))
    idx.set_option("single_query_direct", 0)


def test_lockstep_table_mode_still_matches(env):
    """Stored-embedding mode defaults to the persistent kernel; the lock-step path must give the same answers."""
    from leann_amd.index import Mi355xIndex

    x, g = _build(3000, 384, "mips", seed=1)
    q = queries_near(x, 32, seed=2)
    idx = Mi355xIndex.from_csr(g)
    idx.attach_table(x)
    res = []
    for persistent in (1, 0):
        idx.set_option("persistent_table", persistent)
        for ef, beam, cr in ((64, 1, True), (48, 4, True), (24, 2, False)):
            d, l = idx.search(q, 10, idx.make_params(ef=ef, beam=beam, recompute=False, check_relative_distance=cr))
            st = idx.stats()
            res.append((persistent, d, l, st["ndis"], st["nexpand"], st["nrounds"]))
    h = len(res) // 2
    for a, b in zip(res[:h], res[h:]):
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3:] == b[3:], (a[3:], b[3:])


def test_hub_cache_same_results_fewer_recomputes(env):
    """Hub-embedding cache: cached nodes never reach the provider; ids/distances unchanged (with and without memo)."""
    from leann_amd.backend import hub_nodes
    from leann_amd.devmem import as_tensor
    from leann_amd.index import Mi355xIndex
    from oracle import oracle as orc

    torch = env
    x, g = _build(3000, 384, "mips", seed=1)
    q = queries_near(x, 64, seed=13)
    xdev = torch.from_numpy(x).cuda()
    hubs = hub_nodes(g, 0.1)
    assert hubs.shape[0] == 300 and np.all(np.diff(hubs) > 0)
    idx = Mi355xIndex.from_csr(g)
    idx.set_stream(torch.cuda.current_stream().cuda_stream)
    keep, seen = {}, []

    def provider(d_ids, n, stream):
        ids = as_tensor(d_ids, (n,), "int32")
        seen.append(ids.cpu().numpy().copy())
        keep["e"] = xdev.index_select(0, ids.long()).contiguous()
        return keep["e"].data_ptr()

    idx.set_provider(provider)
    oi, od, _ = orc.search(oracle_graph(g, 384), q, 10, ef=64, beam=2, table=x)
    base = {}
    for memo in (False, True):
        idx.search_device(torch.from_numpy(q).cuda(), 10, idx.make_params(ef=64, beam=2, recompute=True, recompute_memo=memo))
        base[memo] = idx.stats()["nunique"]
    assert base[True] < base[False]
    idx.set_hub_cache(hubs, xdev[torch.from_numpy(hubs).long().cuda()].contiguous())
    for memo in (False, True):
        seen.clear()
        d, l = idx.search_device(torch.from_numpy(q).cuda(), 10, idx.make_params(ef=64, beam=2, recompute=True, recompute_memo=memo))
        torch.cuda.synchronize()
        assert np.array_equal(l.cpu().numpy(), oi) and np.array_equal(d.cpu().numpy(), od)
        allids = np.concatenate(seen)
        assert not np.isin(allids, hubs).any() and idx.stats()["nunique"] < base[memo]
    idx.set_hub_cache(None)
    seen.clear()
    idx.search_device(torch.from_numpy(q).cuda(), 10, idx.make_params(ef=64, beam=2, recompute=True))  # the default: memo on
    assert idx.stats()["nunique"] == base[True] and np.isin(np.concatenate(seen), hubs).any()


def test_fp16_table(env):
    x, g = _build(1500, 768, "mips", seed=5)
    q = queries_near(x, 24, seed=8)
    _check(env, x, g, q, 10, 64, 1, "table", table_dtype=np.float16)


def test_check_relative_distance_off(env):
    x, g = _build(2000, 128, "mips", seed=9)
    q = queries_near(x, 24, seed=10)
    _check(env, x, g, q, 10, 24, 1, "table", check_rel=False)
    _check(env, x, g, q, 10, 24, 4, "table", check_rel=False)


def test_ties_duplicate_vectors(env):
    """Duplicate vectors -> equal distances -> ordering by id (SURVEY 8c golden case iv)."""
    x, _ = _build(1500, 64, "l2", seed=11, normalize=False)
    x[500:1000] = x[:500]  # every vector in [0,500) has an exact duplicate
    from leann_amd.hnsw_builder import build_hnsw

    g = build_hnsw(x, "l2", M=12, ef_construction=60, seed=11)
    q = x[:32].copy()
    gi = _check(env, x, g, q, 4, 48, 2, "table")
    for i in range(32):
        assert gi[i, 0] == i and gi[i, 1] == i + 500  # distance 0 twice, smaller id first


def test_k_larger_than_reachable_and_tiny_index(env):
    from leann_amd.hnsw_builder import build_hnsw

    x = clustered(7, 64, 12)
    g = build_hnsw(x, "mips", M=4, ef_construction=10)
    q = queries_near(x, 3, 13)
    gi = _check(env, x, g, q, 10, 16, 1, "table")
    assert (gi[:, 7:] == -1).all() and (gi[:, :7] >= 0).all()


def test_single_node_and_empty(env):
    from leann_amd.csr_format import HnswCsr
    from leann_amd.hnsw_builder import build_hnsw
    from leann_amd.index import Mi355xIndex

    x = clustered(1, 64, 14)
    g = build_hnsw(x, "l2", M=4, ef_construction=10)
    _check(env, x, g, x.copy(), 3, 8, 1, "table")
    ge = build_hnsw(np.zeros((0, 64), np.float32), "l2")
    assert isinstance(ge, HnswCsr) and ge.ntotal == 0
    idx = Mi355xIndex.from_csr(ge)
    d, l = idx.search(np.zeros((2, 64), np.float32), 3, idx.make_params(ef=8, recompute=False))
    assert (l == -1).all() and np.isinf(d).all()


def test_recall_full_size_property(env):
    """Size-independent property at a larger N: recall vs exact top-k >= 0.9 at ef=64 and
    monotone in ef; results sorted best-first."""
    from leann_amd.index import Mi355xIndex
    from oracle import oracle as orc

    x, g = _build(30000, 128, "mips", seed=15, M=16, efc=80)
    q = queries_near(x, 200, seed=16)
    gt, _ = orc.bruteforce_topk(x, q, 10, 0)
    idx = Mi355xIndex.from_csr(g)
    idx.attach_table(x)
    recs = []
    for ef in (16, 64, 128):
        d, l = idx.search(q, 10, idx.make_params(ef=ef, beam=2, recompute=False))
        assert np.all(np.diff(d, axis=1) <= 0)  # +IP descending
        recs.append(recall_at_k(l, gt))
    assert recs[1] >= 0.9 and recs[0] <= recs[1] + 1e-9 <= recs[2] + 2e-9, recs
    # idempotence + launch-structure independence: same call twice, persistent vs lock-step -> identical output
    prm = idx.make_params(ef=64, beam=2, recompute=False)
    d1, l1 = idx.search(q, 10, prm)
    d2, l2 = idx.search(q, 10, prm)
    idx.set_option("persistent_table", 0)
    d3, l3 = idx.search(q, 10, prm)
    assert np.array_equal(l1, l2) and np.array_equal(d1, d2) and np.array_equal(l1, l3) and np.array_equal(d1, d3)


def test_dist_gather_kernel(env):
    """lm_dist_gather == oracle orc_dist bit-for-bit (both metrics, f32/f16 rows)."""
    import ctypes as C

    torch = env
    from leann_amd import _lib
    from oracle import oracle as orc

    lib = _lib.load()
    rng = np.random.default_rng(17)
    n, d, nq, npairs = 3000, 384, 32, 5000
    x = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    ids = rng.integers(0, n, npairs).astype(np.int32)
    qidx = rng.integers(0, nq, npairs).astype(np.int32)
    for dtype, code in ((np.float32, 0), (np.float16, 1)):
        xt = x.astype(dtype)
        tx = torch.from_numpy(xt).cuda()
        tq = torch.from_numpy(q).cuda()
        ti, tqi = torch.from_numpy(ids).cuda(), torch.from_numpy(qidx).cuda()
        out = torch.empty(npairs, dtype=torch.float32, device="cuda")
        for metric in (0, 1):
            _lib.check(lib.lm_dist_gather(C.c_void_p(tx.data_ptr()), code, d, metric, C.c_void_p(tq.data_ptr()),
                                          C.c_void_p(tqi.data_ptr()), C.c_void_p(ti.data_ptr()), npairs,
                                          C.c_void_p(out.data_ptr()), None))
            torch.cuda.synchronize()
            got = out.cpu().numpy()
            xf = xt.astype(np.float32)
            exp = np.array([orc.dist(xf[ids[p]], q[qidx[p]], metric) for p in range(0, npairs, 7)], dtype=np.float32)
            assert np.array_equal(got[::7].view(np.uint32), exp.view(np.uint32))
            # and within 1e-4 relative of the reference server's numpy formula (hnsw_embedding_server.py:195-200)
            ref = (np.sum((xf[ids] - q[qidx]) ** 2, axis=1) if metric == 1 else -np.einsum("ij,ij->i", xf[ids], q[qidx]))
            assert np.allclose(got, ref, rtol=1e-4, atol=1e-4)


def test_topk_merge_kernel(env):
    import ctypes as C

    torch = env
    from leann_amd import _lib
    from oracle import oracle as orc

    lib = _lib.load()
    rng = np.random.default_rng(18)
    S, B, k = 8, 37, 10
    for metric in (0, 1):
        ids = rng.integers(0, 10**9, (S, B, k)).astype(np.int64)
        dist = np.sort(rng.standard_normal((S, B, k)).astype(np.float32), axis=2)
        if metric == 0:
            dist = dist[:, :, ::-1].copy()
        ids[3, :, 6:] = -1  # short shard
        dist[5, 4, :] = dist[2, 4, :]  # ties across shards
        ti, td = torch.from_numpy(ids).cuda(), torch.from_numpy(dist).cuda()
        oi = torch.empty((B, k), dtype=torch.int64, device="cuda")
        od = torch.empty((B, k), dtype=torch.float32, device="cuda")
        _lib.check(lib.lm_topk_merge(C.c_void_p(ti.data_ptr()), C.c_void_p(td.data_ptr()), S, B, k, metric,
                                     C.c_void_p(oi.data_ptr()), C.c_void_p(od.data_ptr()), None))
        torch.cuda.synchronize()
        ei, ed = orc.merge_topk(ids, dist, metric)
        assert np.array_equal(oi.cpu().numpy(), ei) and np.array_equal(od.cpu().numpy(), ed)


@pytest.mark.parametrize("strategy", ["global", "local", "proportional"])
def test_two_level_search_parity(env, strategy):
    """prune_ratio / pruning_strategy (hnsw_backend.py:219-231; paper Alg. 2): PQ-ADC ranked pruning before the
    exact evaluation -- bit-exact with the oracle in stored-embedding and recompute (+memo) modes."""
    from leann_amd.devmem import as_tensor
    from leann_amd.index import Mi355xIndex
    from leann_amd.pq import encode_pq, train_pq
    from oracle import oracle as orc

    torch = env
    x, g = _build(3000, 96, "mips", seed=21)
    q = queries_near(x, 40, seed=22)
    cb = train_pq(torch.from_numpy(x), 24, iters=6, seed=1).numpy()
    codes = encode_pq(torch.from_numpy(x), torch.from_numpy(cb)).numpy()
    og = oracle_graph(g, 96)
    idx = Mi355xIndex.from_csr(g)
    idx.set_stream(torch.cuda.current_stream().cuda_stream)
    idx.set_profiling(True)  # exercise the timing buffers across mode switches as well
    with pytest.raises(RuntimeError):  # no PQ attached yet
        idx.attach_table(x)
        idx.search(q, 10, idx.make_params(ef=32, recompute=False, prune_ratio=0.5))
    idx.attach_pq(cb, codes)
    xdev = torch.zeros((3000, idx.info.d_padded), device="cuda")
    xdev[:, :96] = torch.from_numpy(x).cuda()
    keep = {}

    def provider(d_ids, n, stream):
        keep["e"] = xdev.index_select(0, as_tensor(d_ids, (n,), "int32").long()).contiguous()
        return keep["e"].data_ptr()

    idx.set_provider(provider)
    base = None
    for ratio in (0.3, 0.6):
        for ef, beam in ((32, 1), (64, 3)):
            oi, od, ost = orc.search(og, q, 10, ef=ef, beam=beam, table=x, prune_ratio=ratio, pruning_strategy=strategy, pq=(cb, codes))
            kw = dict(ef=ef, beam=beam, prune_ratio=ratio, local_prune=(strategy == "local"),
                      send_neigh_times_ratio=(1.0 if strategy == "proportional" else 0.0))
            for mode in ("table", "provider", "memo"):
                prm = idx.make_params(recompute=(mode != "table"), recompute_memo=(mode == "memo"), **kw)
                gd, gi = idx.search(q, 10, prm)
                st = idx.stats()
                assert np.array_equal(gi, oi) and np.array_equal(gd.view(np.uint32), od.view(np.uint32)), (ratio, ef, beam, mode)
                assert st["ndis"] == ost["ndis"] and st["nadc"] == ost["nadc"] and st["nrounds"] == ost["nrounds"], (st, ost)
                assert st["update_span_launches"] == st["update_launches"] > 0 and 0 < st["update_span_ms"] <= st["update_ms"] * 1.05
            if base is None:
                _, _, bst = orc.search(og, q, 10, ef=ef, beam=beam, table=x)
                assert ost["ndis"] < bst["ndis"]  # fewer exact evaluations than the unpruned search
                base = bst
    idx.close()


def test_random_degenerate_graphs_match_oracle(env):
    """Seeded random tiny graphs with empty neighbour lists, unreachable nodes, upper-level stubs: GPU == oracle in
    stored-embedding (persistent and lock-step) and recompute modes, incl. unfilled result slots."""
    from leann_amd.devmem import as_tensor
    from leann_amd.index import Mi355xIndex
    from oracle import oracle as orc
    from tests.test_oracle_properties import _random_graph

    torch = env
    for seed in range(40):
        rng = np.random.default_rng(1000 + seed)
        n = int(rng.integers(1, 30))
        metric = int(rng.integers(0, 2))
        x, g, _ = _random_graph(rng, n, 64, metric)
        q = rng.standard_normal((5, 64)).astype(np.float32)
        k, ef, beam = int(rng.integers(1, 7)), int(rng.integers(1, n + 5)), int(rng.integers(1, 5))
        oi, od, ost = orc.search(oracle_graph(g, 64), q, k, ef=ef, beam=beam, table=x)
        idx = Mi355xIndex.from_csr(g)
        idx.set_stream(torch.cuda.current_stream().cuda_stream)
        idx.attach_table(x)
        xdev = torch.from_numpy(x).cuda()
        keep = {}

        def provider(d_ids, cnt, stream):
            keep["e"] = xdev.index_select(0, as_tensor(d_ids, (cnt,), "int32").long()).contiguous()
            return keep["e"].data_ptr()

        idx.set_provider(provider)
        for mode in ("persistent", "lockstep", "provider"):
            idx.set_option("persistent_table", 1 if mode == "persistent" else 0)
            d, l = idx.search(q, k, idx.make_params(ef=ef, beam=beam, recompute=(mode == "provider")))
            st = idx.stats()
            assert np.array_equal(l, oi) and np.array_equal(d.view(np.uint32), od.view(np.uint32)), (seed, mode, n, k, ef, beam)
            assert (st["ndis"], st["nexpand"], st["nrounds"]) == (ost["ndis"], ost["nexpand"], ost["nrounds"]), (seed, mode, st, ost)
        idx.close()
