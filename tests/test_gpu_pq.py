"""DiskANN-style path on the GPU (PQ-ADC persistent traversal kernel + deferred rerank) vs the oracle."""
import numpy as np
import pytest

from tests.util import clustered, oracle_graph, queries_near, recall_at_k

pytestmark = pytest.mark.gpu
_C = {}


def _setup(n, d, m, metric, seed):
    import torch

    from leann_amd.hnsw_builder import build_hnsw
    from leann_amd.pq import encode_pq, flat_graph, train_pq

    key = (n, d, m, metric, seed)
    if key not in _C:
        x = clustered(n, d, seed, n_centers=64, sigma=0.5)
        g = flat_graph(build_hnsw(x, metric, M=12, ef_construction=60, seed=seed), x)
        cb = train_pq(torch.from_numpy(x), m, iters=8, seed=seed).numpy()
        codes = encode_pq(torch.from_numpy(x), torch.from_numpy(cb)).numpy()
        _C[key] = (x, g, cb, codes)
    return _C[key]


@pytest.mark.parametrize("metric", ["l2", "mips"])
@pytest.mark.parametrize("L,W", [(32, 1), (64, 4), (100, 64)])
def test_pq_search_parity(metric, L, W):
    import torch

    from leann_amd import _lib
    from leann_amd.devmem import as_tensor
    from leann_amd.index import Mi355xIndex
    from oracle import oracle as orc

    _lib.require_gpu()
    x, g, cb, codes = _setup(4000, 96, 24, metric, 3)
    q = queries_near(x, 40, 4)
    og = oracle_graph(g, 96)
    idx = Mi355xIndex.from_csr(g)
    idx.set_stream(torch.cuda.current_stream().cuda_stream)
    idx.attach_pq(cb, codes)
    # (1) PQ order only
    oi, od, ost = orc.pq_search(og, cb, codes, q, 10, L=L, W=W, skip_search_reorder=True)
    for threads in (1024, 512, 256):  # workgroup width of the traversal kernel (default 1024): a launch-shape choice, identical results
        idx.set_option("pq_threads", threads)
        gi, gd = idx.pq_search(q, 10, idx.make_pq_params(L, W, skip_search_reorder=True))
        st = idx.stats()
        assert np.array_equal(gi, oi) and np.array_equal(gd.view(np.uint32), od.view(np.uint32)), threads
        assert st["ndis"] == ost["n_adc"] and st["nexpand"] == ost["n_expand"] and st["nrounds"] == ost["n_rounds"], (threads, st, ost)
    idx.set_option("pq_threads", 1024)
    # (2) deferred fetch through the provider (one call, sorted unique ids)
    xdev = torch.zeros((x.shape[0], idx.info.d_padded), device="cuda")
    xdev[:, :96] = torch.from_numpy(x).cuda()
    calls, keep = [], {}

    def provider(d_ids, n, stream):
        ids = as_tensor(d_ids, (n,), "int32")
        calls.append(ids.cpu().numpy().copy())
        keep["e"] = xdev.index_select(0, ids.long()).contiguous()
        return keep["e"].data_ptr()

    idx.set_provider(provider)
    oi, od, ost = orc.pq_search(og, cb, codes, q, 10, L=L, W=W, provider=lambda idv: x[idv], use_deferred_fetch=True)
    gi, gd = idx.pq_search(q, 10, idx.make_pq_params(L, W, use_deferred_fetch=True))
    assert len(calls) == 1 and np.all(np.diff(calls[0]) > 0) and len(calls[0]) == ost["n_rerank_unique"]
    assert np.array_equal(gi, oi) and np.array_equal(gd.view(np.uint32), od.view(np.uint32))
    assert np.allclose(gd, od, atol=1e-4, rtol=0)
    # (3) stored embeddings (recompute_embeddings=False on an index that keeps them)
    idx.set_provider(None)
    idx.attach_table(x)
    oi, od, _ = orc.pq_search(og, cb, codes, q, 10, L=L, W=W, table=x)
    gi, gd = idx.pq_search(q, 10, idx.make_pq_params(L, W))
    assert np.array_equal(gi, oi) and np.array_equal(gd.view(np.uint32), od.view(np.uint32))
    gt, _ = orc.bruteforce_topk(x, q, 10, 1 if metric == "l2" else 0)
    if L >= 64:
        assert recall_at_k(gi, gt) > 0.9
    # (4) option "pq_rerank_expanded": the rerank set = every expanded node = upstream DiskANN's full_retset; the second oracle
    # (oracle/lm_oracle_diskann.c, a transcription of PQFlashIndex::cached_beam_search) with rerank_final_list_only off -- through the
    # table and through the deferred fetch (ONE provider call over the union of the expanded sets)
    idx.set_option("pq_rerank_expanded", 1)
    ui, ud, ust = orc.diskann_search(og, cb, codes, q, 10, L=L, W=W, table=x, rerank_final_list_only=False)
    gi2, gd2 = idx.pq_search(q, 10, idx.make_pq_params(L, W))
    assert np.array_equal(gi2, ui) and np.array_equal(gd2.view(np.uint32), ud.view(np.uint32))
    calls.clear()
    idx.set_provider(provider)
    gi3, gd3 = idx.pq_search(q, 10, idx.make_pq_params(L, W, use_deferred_fetch=True))
    assert len(calls) == 1 and np.array_equal(gi3, ui) and np.array_equal(gd3.view(np.uint32), ud.view(np.uint32))
    assert idx.get_option("pq_rerank_overflow") == 0
    assert recall_at_k(gi2, gt) >= recall_at_k(gi, gt)  # a superset of the final list can only rank better
    # the option + skip_search_reorder: no rerank follows, so the PQ-ordered list (case 1) must come back, not the expanded-node record
    oi, od, _ = orc.pq_search(og, cb, codes, q, 10, L=L, W=W, skip_search_reorder=True)
    gi4, gd4 = idx.pq_search(q, 10, idx.make_pq_params(L, W, skip_search_reorder=True))
    assert np.array_equal(gi4, oi) and np.array_equal(gd4.view(np.uint32), od.view(np.uint32))
    idx.set_option("pq_rerank_expanded", 0)
    idx.close()


@pytest.mark.parametrize("L", [256, 512])
def test_pq_search_parity_at_the_shape_c3_is_measured_on(L):
    """The configuration scripts/bench_c3.py times -- D = 384, m = 96 (the 96 KB fp32 lookup table: one 1024-thread workgroup per CU),
    W = 64, L = 256 (the round-5 line) / 512 (with a degree-64 graph a frontier of 64 x 64 new nodes and L = 1024 no longer fit the 160 KB of LDS
    next to the table: the library says so) -- on 200k vectors (GPU-built degree-64 flat graph), against BOTH oracles: oracle/lm_oracle_pq.c (PQ order with
    counts, stored-table rerank) and the DiskANN transcription oracle/lm_oracle_diskann.c (final-list and expanded-node rerank sets).  Ids and
    distance BITS.  (VERDICT r4 missing #3: the parity tests above stop at D = 96, m = 24, L = 100; the run's own 10M index is compared in
    bench_c3.py's parity_check block.)"""
    import torch

    from leann_amd import _lib
    from leann_amd.gpu_graph_build import build_graph_gpu
    from leann_amd.index import Mi355xIndex
    from leann_amd.pq import encode_pq, flat_graph, train_pq
    from oracle import oracle as orc

    _lib.require_gpu()
    key = ("c3-shape",)
    if key not in _C:
        x = clustered(200_000, 384, 11, n_centers=2000, sigma=0.35)
        xt = torch.from_numpy(x).cuda()
        g = flat_graph(build_graph_gpu(xt, "mips", M=32, ef_construction=100), xt)
        cb = train_pq(xt, 96, iters=6, seed=0)
        codes = encode_pq(xt, cb)
        _C[key] = (x, g, cb.cpu().numpy(), codes.cpu().numpy())
    x, g, cb, codes = _C[key]
    q = queries_near(x, 48, 12)
    og = oracle_graph(g, 384)
    idx = Mi355xIndex.from_csr(g)
    idx.set_stream(torch.cuda.current_stream().cuda_stream)
    idx.attach_pq(cb, codes)
    bits = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)  # noqa: E731
    W = 64
    oi, od, ost = orc.pq_search(og, cb, codes, q, 10, L=L, W=W, skip_search_reorder=True)
    gi, gd = idx.pq_search(q, 10, idx.make_pq_params(L, W, skip_search_reorder=True))
    st = idx.stats()
    assert np.array_equal(gi, oi) and np.array_equal(bits(gd), bits(od))
    assert st["ndis"] == ost["n_adc"] and st["nexpand"] == ost["n_expand"] and st["nrounds"] == ost["n_rounds"], (st, ost)
    idx.attach_table(x)
    ti, td, _ = orc.pq_search(og, cb, codes, q, 10, L=L, W=W, table=x)
    gi2, gd2 = idx.pq_search(q, 10, idx.make_pq_params(L, W))
    assert np.array_equal(gi2, ti) and np.array_equal(bits(gd2), bits(td))
    fi, fd, _ = orc.diskann_search(og, cb, codes, q, 10, L=L, W=W, table=x, rerank_final_list_only=True)
    assert np.array_equal(fi, ti) and np.array_equal(bits(fd), bits(td))
    idx.set_option("pq_rerank_expanded", 1)
    ui, ud, _ = orc.diskann_search(og, cb, codes, q, 10, L=L, W=W, table=x, rerank_final_list_only=False)
    gi3, gd3 = idx.pq_search(q, 10, idx.make_pq_params(L, W))
    assert idx.get_option("pq_rerank_overflow") == 0
    assert np.array_equal(gi3, ui) and np.array_equal(bits(gd3), bits(ud))
    idx.set_option("pq_rerank_expanded", 0)
    gt, _ = orc.bruteforce_topk(x, q, 10, 0)
    assert recall_at_k(gi2, gt) > 0.9
    idx.close()


def test_pq_errors_and_edge_cases():
    import torch

    from leann_amd import _lib
    from leann_amd.index import Mi355xIndex

    _lib.require_gpu()
    x, g, cb, codes = _setup(4000, 96, 24, "l2", 3)
    idx = Mi355xIndex.from_csr(g)
    with pytest.raises(RuntimeError):  # no codes attached
        idx.pq_search(x[:1], 3, idx.make_pq_params(16, 1))
    with pytest.raises(ValueError):
        idx.attach_pq(cb[:, :, :3], codes)
    idx.attach_pq(cb, codes)
    with pytest.raises(RuntimeError):  # deferred fetch without provider/table
        idx.pq_search(x[:1], 3, idx.make_pq_params(16, 1, use_deferred_fetch=True))
    with pytest.raises(ValueError):  # beam > 64
        idx.pq_search(x[:1], 3, idx.make_pq_params(16, 65, skip_search_reorder=True))
    with pytest.raises(ValueError, match="recompute_neighbors"):  # the reference passes 0; anything else is refused, not ignored
        prm_rn = idx.make_pq_params(16, 1, skip_search_reorder=True)
        prm_rn.recompute_neighbors = 1
        idx.pq_search(x[:1], 3, prm_rn)
    l, d = idx.pq_search(x[:2], 5000, idx.make_pq_params(16, 1, skip_search_reorder=True))  # k > N reachable
    assert (l[:, 4000:] == -1).all()
