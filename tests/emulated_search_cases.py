"""Scenarios run against libleann_mi355x_emul.so (tests/hip_emul/build_emul_lib.py): the product's C ABI and host code,
its HIP kernels executed on the CPU with a thread per lane.  Imported by tests/test_emulated_search.py and runnable as a
script (the ThreadSanitizer run preloads the TSan runtime into this interpreter):
    python -m tests.emulated_search_cases <path/to/libleann_mi355x_emul.so> [case ...]
Every case compares with the oracle (bit-exact labels AND distances, same number of distance evaluations)."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np


def _load(lib_path: str):
    from leann_amd import _lib

    _lib.LIB_PATH = Path(lib_path)
    _lib._lib = None
    return _lib.load()


class NumpyProvider:
    """lm_provider_fn in the emulated world: 'device' pointers are host pointers."""

    def __init__(self, table: np.ndarray, dp: int, sorted_ids: bool = True):
        self.x = np.zeros((table.shape[0], dp), np.float32)
        self.x[:, : table.shape[1]] = table
        self.keep = None
        self.calls = 0
        self.sorted_ids = sorted_ids  # False: option "single_query_direct" (unique ids in discovery order)

    def __call__(self, d_ids_ptr: int, n: int, stream_ptr: int) -> int:
        ids = np.ctypeslib.as_array(C.cast(d_ids_ptr, C.POINTER(C.c_int32)), shape=(n,))
        assert np.all(ids[1:] > ids[:-1]) if self.sorted_ids else np.unique(ids).shape[0] == n  # sorted unique, as the ABI promises
        self.keep = np.ascontiguousarray(self.x[ids])
        self.calls += 1
        return self.keep.ctypes.data


def _data(n, d, seed, nq=3):
    from tests.util import clustered, queries_near

    x = clustered(n, d, seed)
    return x, queries_near(x, nq, seed + 1)


def _check(tag, got, exp, st_got=None, st_exp=None):
    d, l = got
    el, ed = exp
    ok = np.array_equal(l, el) and np.array_equal(d, ed)
    if st_got is not None:
        ok = ok and int(st_got["ndis"]) == int(st_exp["ndis"])
    print(f"{tag}: {'ok' if ok else 'MISMATCH'}", flush=True)
    assert ok, tag


def case_table(metric="mips", d=64, f16=False):
    from leann_amd.hnsw_builder import build_hnsw
    from leann_amd.index import Mi355xIndex
    from oracle import oracle as orc

    x, q = _data(260, d, 3 + d)
    g = build_hnsw(x, metric, M=6, ef_construction=30)
    og = orc.OracleGraph(g.node_offsets, g.level_ptr, g.neighbors, g.levels, g.entry_point, g.max_level, g.metric_type, d)
    idx = Mi355xIndex.from_csr(g)
    tab = x.astype(np.float16) if f16 else x
    idx.attach_table(tab)
    ref_tab = tab.astype(np.float32)
    for persistent, wave in ((1, 0), (1, 1), (0, 0)):  # persistent kernel: workgroup / wave per query; lock-step rounds
        idx.set_option("persistent_table", persistent)
        idx.set_option("persistent_wave", wave)
        for beam, ef in ((1, 12), (3, 20)):
            got = idx.search(q, 5, idx.make_params(ef=ef, beam=beam, recompute=False))
            exp = orc.search(og, q, 5, ef=ef, beam=beam, table=ref_tab)
            _check(f"table {metric} d={d} f16={f16} persistent={persistent} wave={wave} beam={beam}", got, exp[:2], idx.stats(), exp[2])
    idx.close()


def case_recompute(variant=0, memo=False, initial_rows=0, nq=3):
    """Recompute mode.  memo=None: the library default (per-call memo ON for a pass of more than one query).  With the memo every node
    reaches the provider at most once per call (asserted on the provider's own id log) and the results are those of the plain
    lock-step recompute; initial_rows > 0 starts the memo that small, so that it has to grow (rows kept across the re-allocation)."""
    from leann_amd.hnsw_builder import build_hnsw
    from leann_amd.index import Mi355xIndex
    from oracle import oracle as orc

    x, q = _data(240, 48, 11, nq=nq)  # d = 48 -> padded to 64
    g = build_hnsw(x, "mips", M=6, ef_construction=30)
    og = orc.OracleGraph(g.node_offsets, g.level_ptr, g.neighbors, g.levels, g.entry_point, g.max_level, g.metric_type, 48)
    idx = Mi355xIndex.from_csr(g)
    prov = NumpyProvider(x, idx.info.d_padded)
    seen = []
    inner = prov.__call__

    def logging_provider(d_ids_ptr, n, stream_ptr):
        seen.append(np.ctypeslib.as_array(C.cast(d_ids_ptr, C.POINTER(C.c_int32)), shape=(n,)).copy())
        return inner(d_ids_ptr, n, stream_ptr)

    idx.set_provider(logging_provider)
    idx.set_option("update_variant", variant)
    if initial_rows:
        idx.set_option("memo_initial_rows", initial_rows)
    kw = {} if memo is None else {"recompute_memo": memo}
    got = idx.search(q, 5, idx.make_params(ef=14, beam=2, recompute=True, **kw))
    exp = orc.search(og, q, 5, ef=14, beam=2, table=x)
    _check(f"recompute variant={variant} memo={memo} initial_rows={initial_rows} nq={nq}", got, exp[:2], idx.stats(), exp[2])
    assert prov.calls > 0
    allids = np.concatenate(seen)
    st = idx.stats()
    assert int(st["nunique"]) == allids.shape[0]
    memo_on = (memo is None or memo) and nq > 1
    if memo_on:
        assert np.unique(allids).shape[0] == allids.shape[0], "a node reached the provider twice despite the memo"
        assert allids.shape[0] < int(st["ndis"])
    elif nq > 1:
        assert np.unique(allids).shape[0] < allids.shape[0], "expected repeated nodes without the memo (else this case proves nothing)"
    idx.close()


def case_speculative_prefetch():
    """Option "speculate" (k_speculate): a small recompute batch embeds the neighbours of its best unexpanded candidates ahead of time.
    Labels, distances, rounds and distance-evaluation counts must be those of the oracle and of S = 0 (bit for bit); the provider is called
    less often, never twice for a node, and is asked for at least as many ids."""
    from leann_amd.hnsw_builder import build_hnsw
    from leann_amd.index import Mi355xIndex
    from oracle import oracle as orc

    for metric, nq, beam, check in (("mips", 1, 1, True), ("l2", 1, 2, False), ("mips", 2, 1, True)):
        x, q = _data(400, 48, 23, nq=nq)
        g = build_hnsw(x, metric, M=6, ef_construction=30)
        og = orc.OracleGraph(g.node_offsets, g.level_ptr, g.neighbors, g.levels, g.entry_point, g.max_level, g.metric_type, 48)
        exp = orc.search(og, q, 5, ef=24, beam=beam, check_relative_distance=check, table=x)
        runs = {}
        for S in (0, 1, 3, 64):
            idx = Mi355xIndex.from_csr(g)
            prov = NumpyProvider(x, idx.info.d_padded)
            seen = []
            inner = prov.__call__

            def logging_provider(d_ids_ptr, n, stream_ptr, seen=seen, inner=inner):
                seen.append(np.ctypeslib.as_array(C.cast(d_ids_ptr, C.POINTER(C.c_int32)), shape=(n,)).copy())
                return inner(d_ids_ptr, n, stream_ptr)

            idx.set_provider(logging_provider)
            idx.set_option("speculate", S)
            assert idx.get_option("speculate") == S and idx.get_option("speculate_max_batch") == 2
            got = idx.search(q, 5, idx.make_params(ef=24, beam=beam, recompute=True, check_relative_distance=check))
            st = idx.stats()
            _check(f"speculative prefetch S={S} metric={metric} nq={nq} beam={beam} check={check}", got, exp[:2], st, exp[2])
            allids = np.concatenate(seen)
            assert int(st["nunique"]) == allids.shape[0]
            if S > 0 or nq > 1:
                assert np.unique(allids).shape[0] == allids.shape[0], "a node reached the provider twice"
            runs[S] = (got, int(st["nrounds"]), int(st["ndis"]), len(seen), np.unique(allids))
            idx.close()
        for S in (1, 3, 64):
            assert np.array_equal(runs[S][0][0], runs[0][0][0]) and np.array_equal(runs[S][0][1], runs[0][0][1]) and runs[S][1:3] == runs[0][1:3], S
            assert runs[S][3] <= runs[0][3] and np.isin(runs[0][4], runs[S][4]).all(), (S, runs[S][3], runs[0][3])  # a superset of the ids, in fewer calls
        assert runs[3][3] < runs[0][3], ("the prefetch saved no provider call", {s: r[3] for s, r in runs.items()})
        print(f"speculative prefetch {metric} nq={nq}: (provider calls, distinct ids requested) by S:", {s: (r[3], r[4].shape[0]) for s, r in runs.items()}, flush=True)
    # a batch above speculate_max_batch does not prefetch; the two-level search does not either
    x, q = _data(300, 48, 29, nq=3)
    g = build_hnsw(x, "mips", M=6, ef_construction=30)
    og = orc.OracleGraph(g.node_offsets, g.level_ptr, g.neighbors, g.levels, g.entry_point, g.max_level, g.metric_type, 48)
    idx = Mi355xIndex.from_csr(g)
    idx.set_provider(NumpyProvider(x, idx.info.d_padded))
    base = idx.search(q, 5, idx.make_params(ef=14, beam=1, recompute=True))
    n0 = int(idx.stats()["nunique"])
    idx.set_option("speculate", 4)
    again = idx.search(q, 5, idx.make_params(ef=14, beam=1, recompute=True))
    assert np.array_equal(base[0], again[0]) and np.array_equal(base[1], again[1]) and int(idx.stats()["nunique"]) == n0
    idx.set_option("speculate_max_batch", 8)
    spec = idx.search(q, 5, idx.make_params(ef=14, beam=1, recompute=True))
    assert np.array_equal(base[0], spec[0]) and np.array_equal(base[1], spec[1]) and int(idx.stats()["nunique"]) >= n0
    idx.close()


def case_single_query_direct():
    """Option "single_query_direct": a one-query recompute pass hands its new-list to the provider as it is (discovery order, no request
    bitmap, no k_uniq_* launches, k_expand writes the count and the live flag).  Labels, distances, rounds and counts must be the oracle's
    and those of the default path; the provider sees the same ids per round, in a different order; batches of more than one query and
    passes with a memo (hub cache, speculative prefetch) do not take the path."""
    from leann_amd.hnsw_builder import build_hnsw
    from leann_amd.index import Mi355xIndex
    from oracle import oracle as orc

    for metric, beam, check, k, ef in (("mips", 1, True, 5, 24), ("l2", 3, False, 5, 12), ("mips", 2, True, 9, 4)):
        x, q = _data(400, 48, 37, nq=2)
        g = build_hnsw(x, metric, M=6, ef_construction=30)
        og = orc.OracleGraph(g.node_offsets, g.level_ptr, g.neighbors, g.levels, g.entry_point, g.max_level, g.metric_type, 48)
        exp = orc.search(og, q[:1], k, ef=ef, beam=beam, check_relative_distance=check, table=x)
        runs = {}
        for direct in (0, 1):
            idx = Mi355xIndex.from_csr(g)
            prov = NumpyProvider(x, idx.info.d_padded, sorted_ids=not direct)
            seen = []
            inner = prov.__call__

            def logging_provider(d_ids_ptr, n, stream_ptr, seen=seen, inner=inner):
                seen.append(np.ctypeslib.as_array(C.cast(d_ids_ptr, C.POINTER(C.c_int32)), shape=(n,)).copy())
                return inner(d_ids_ptr, n, stream_ptr)

            idx.set_provider(logging_provider)
            idx.set_option("single_query_direct", direct)
            assert idx.get_option("single_query_direct") == direct
            got = idx.search(q[:1], k, idx.make_params(ef=ef, beam=beam, recompute=True, check_relative_distance=check))
            st = idx.stats()
            _check(f"single_query_direct={direct} metric={metric} beam={beam} check={check} k={k} ef={ef}", got, exp[:2], st, exp[2])
            runs[direct] = (got, {f: int(st[f]) for f in ("ndis", "nunique", "nrounds", "nexpand")}, [s.copy() for s in seen])
            if direct:  # more than one query: the ordinary path (sorted unique ids per round), same answers as without the option
                seen.clear()
                got2 = idx.search(q, k, idx.make_params(ef=ef, beam=beam, recompute=True, check_relative_distance=check))
                exp2 = orc.search(og, q, k, ef=ef, beam=beam, check_relative_distance=check, table=x)
                _check("single_query_direct set, two queries", got2, exp2[:2], idx.stats(), exp2[2])
                assert all((np.diff(s) > 0).all() for s in seen)
                prov.sorted_ids = True
                idx.set_option("speculate", 4)  # a memo pass: the ordinary path again
                got3 = idx.search(q[:1], k, idx.make_params(ef=ef, beam=beam, recompute=True, check_relative_distance=check))
                _check("single_query_direct + speculate", got3, exp[:2], idx.stats(), exp[2])
            idx.close()
        assert runs[0][1] == runs[1][1], (runs[0][1], runs[1][1])
        assert len(runs[0][2]) == len(runs[1][2])
        assert all(np.array_equal(a, np.sort(b)) for a, b in zip(runs[0][2], runs[1][2]))  # the same ids every round ...
        assert any(not np.array_equal(a, b) for a, b in zip(runs[0][2], runs[1][2]))      # ... in discovery order instead of id order
    print("single_query_direct: ok", flush=True)


def case_dynamic_batching():
    """lm_search_params.batch_size (SearchParametersHNSW.batch_size, hnsw_backend.py:234): the paper's dynamic batching, k_expand's extra pops.
    Against the oracle's restatement: labels, distances, distance evaluations, expansions AND rounds, for batch_size 0 / small / larger than any
    list, both stop rules, beam 1 and 3, stored table (lock-step rounds: the persistent kernel declines a batching call) and provider with /
    without the per-call memo, single_query_direct, two-level search; the provider sees the oracle's id lists round by round.  batch_size = 0
    equals the answers every other case pins; a batching search must need FEWER rounds than batch_size 0 (else the case proves nothing)."""
    from leann_amd.hnsw_builder import build_hnsw
    from leann_amd.index import Mi355xIndex
    from oracle import oracle as orc

    x, q = _data(500, 48, 41, nq=4)
    for metric, M in (("mips", 6), ("l2", 10)):
        g = build_hnsw(x, metric, M=M, ef_construction=40)
        og = orc.OracleGraph(g.node_offsets, g.level_ptr, g.neighbors, g.levels, g.entry_point, g.max_level, g.metric_type, 48)
        idx = Mi355xIndex.from_csr(g)
        idx.attach_table(x)
        rounds_at = {}
        for bs in ((0, 1, 7, 24, 400) if metric == "mips" else (0, 24)):
            for beam, ef, k, check in (((1, 24, 5, True), (3, 16, 5, True), (1, 9, 5, False), (2, 4, 9, True)) if bs in (0, 7, 24) else ((1, 24, 5, True),)):
                tag = f"dynamic batching {metric} bs={bs} beam={beam} ef={ef} k={k} check={check}"
                exp = orc.search(og, q, k, ef=ef, beam=beam, check_relative_distance=check, table=x, batch_size=bs)
                rounds_at[(bs, beam, ef, k, check)] = exp[2]["nrounds"]
                # stored table: a batching call runs the lock-step kernels whatever "persistent_table" says
                for persistent in (1, 0):
                    idx.set_option("persistent_table", persistent)
                    got = idx.search(q, k, idx.make_params(ef=ef, beam=beam, recompute=False, check_relative_distance=check, batch_size=bs))
                    st = idx.stats()
                    _check(tag + f" table persistent={persistent}", got, exp[:2], st, exp[2])
                    assert int(st["nexpand"]) == exp[2]["nexpand"] and int(st["nrounds"]) == exp[2]["nrounds"], (tag, dict(st), exp[2])
                # provider, with and without the per-call memo: the oracle's request lists, round by round
                for memo in (True, False):
                    want = []
                    orc.search(og, q, k, ef=ef, beam=beam, check_relative_distance=check, batch_size=bs, memo=memo,
                               provider=lambda idv, want=want: (want.append(idv.copy()), x[idv])[1])
                    prov = NumpyProvider(x, idx.info.d_padded)
                    seen = []

                    def logging_provider(d_ids_ptr, n, stream_ptr, seen=seen, inner=prov.__call__):
                        seen.append(np.ctypeslib.as_array(C.cast(d_ids_ptr, C.POINTER(C.c_int32)), shape=(n,)).copy())
                        return inner(d_ids_ptr, n, stream_ptr)

                    idx.set_provider(logging_provider)
                    got = idx.search(q, k, idx.make_params(ef=ef, beam=beam, recompute=True, check_relative_distance=check, batch_size=bs, recompute_memo=memo))
                    st = idx.stats()
                    _check(tag + f" provider memo={memo}", got, exp[:2], st, exp[2])
                    assert int(st["nexpand"]) == exp[2]["nexpand"] and int(st["nrounds"]) == exp[2]["nrounds"]
                    assert len(seen) == len(want) and all(np.array_equal(a, b) for a, b in zip(seen, want)), tag + " request lists"
        # one query, handed over directly (k_expand writes the list length itself)
        idx.set_option("single_query_direct", 1)
        prov = NumpyProvider(x, idx.info.d_padded, sorted_ids=False)
        idx.set_provider(prov)
        for bs in (0, 24):
            exp = orc.search(og, q[:1], 5, ef=24, table=x, batch_size=bs)
            got = idx.search(q[:1], 5, idx.make_params(ef=24, recompute=True, batch_size=bs))
            _check(f"dynamic batching {metric} single_query_direct bs={bs}", got, exp[:2], idx.stats(), exp[2])
            assert int(idx.stats()["nrounds"]) == exp[2]["nrounds"]
        idx.set_option("single_query_direct", 0)
        assert rounds_at[(24, 1, 24, 5, True)] < rounds_at[(0, 1, 24, 5, True)], rounds_at
        assert metric != "mips" or rounds_at[(400, 1, 24, 5, True)] <= rounds_at[(24, 1, 24, 5, True)]
        try:
            idx.search(q, 5, idx.make_params(ef=8, recompute=False, batch_size=-1))
            raise AssertionError("negative batch_size accepted")
        except ValueError:
            pass
        idx.close()
    print("dynamic batching: ok", flush=True)


def case_stop_rules():
    """Both faiss stop rules against the oracle (itself pinned by the literal transcription, tests/test_oracle_faiss.py):
    k > efSearch (count_below(d0) >= efSearch ends the search although the pool holds k entries) and
    check_relative_distance = 0 (at most efSearch + 1 expansions), persistent and lock-step, table and provider."""
    from leann_amd.hnsw_builder import build_hnsw
    from leann_amd.index import Mi355xIndex
    from oracle import oracle as orc

    x, q = _data(300, 64, 17, nq=4)
    g = build_hnsw(x, "l2", M=4, ef_construction=30)
    og = orc.OracleGraph(g.node_offsets, g.level_ptr, g.neighbors, g.levels, g.entry_point, g.max_level, g.metric_type, 64)
    idx = Mi355xIndex.from_csr(g)
    idx.attach_table(x)
    idx.set_provider(NumpyProvider(x, idx.info.d_padded))
    for k, ef, beam, check in ((12, 3, 1, True), (9, 4, 2, True), (5, 6, 1, False), (8, 3, 2, False)):
        exp = orc.search(og, q, k, ef=ef, beam=beam, check_relative_distance=check, table=x)
        if beam == 1:
            fi, fd, _ = orc.faiss_search(og, q, k, ef=ef, check_relative_distance=check, table=x)
            assert np.array_equal(fi, exp[0]) and np.array_equal(fd, exp[1])
        for persistent in (1, 0):
            idx.set_option("persistent_table", persistent)
            got = idx.search(q, k, idx.make_params(ef=ef, beam=beam, recompute=False, check_relative_distance=check))
            _check(f"stop rules table k={k} ef={ef} beam={beam} check={check} persistent={persistent}", got, exp[:2], idx.stats(), exp[2])
        got = idx.search(q, k, idx.make_params(ef=ef, beam=beam, recompute=True, check_relative_distance=check))
        _check(f"stop rules provider k={k} ef={ef} beam={beam} check={check}", got, exp[:2], idx.stats(), exp[2])
    idx.close()


def case_pq(deferred=True):
    import torch

    from leann_amd.hnsw_builder import build_hnsw
    from leann_amd.index import Mi355xIndex
    from leann_amd.pq import encode_pq, flat_graph, train_pq
    from oracle import oracle as orc

    x, q = _data(300, 32, 21)
    g = flat_graph(build_hnsw(x, "mips", M=6, ef_construction=30), x)
    xt = torch.from_numpy(x)
    cb = train_pq(xt, 8, iters=4, seed=0)
    codes = encode_pq(xt, cb)
    idx = Mi355xIndex.from_csr(g)
    idx.attach_pq(cb.numpy(), codes.numpy())
    if deferred:
        # 256 threads per query here (the emulation pays for every barrier with as many OS threads; the stored-table case below runs
        # the default 1024, `pytest -m gpu` all three widths on hardware)
        idx.set_option("pq_threads", 256)
        idx.set_provider(NumpyProvider(x, idx.info.d_padded))
    else:
        idx.attach_table(x)
    prm = idx.make_pq_params(12, 2, use_deferred_fetch=deferred)
    l, d = idx.pq_search(q, 5, prm)
    og = orc.OracleGraph(g.node_offsets, g.level_ptr, g.neighbors, g.levels, g.entry_point, g.max_level, g.metric_type, 32)
    if deferred:
        el, ed, _ = orc.pq_search(og, cb.numpy(), codes.numpy(), q, 5, L=12, W=2, provider=lambda ids: x[ids], use_deferred_fetch=True)
    else:
        el, ed, _ = orc.pq_search(og, cb.numpy(), codes.numpy(), q, 5, L=12, W=2, table=x)
    _check(f"pq traversal deferred={deferred}", (d, l), (el, ed))
    if not deferred:  # wide beam on a full list: > 256 fresh nodes per hop (several compaction passes) and the below-threshold filter in action
        l2, d2 = idx.pq_search(q, 5, idx.make_pq_params(24, 40))
        st2 = idx.stats()
        el2, ed2, ost2 = orc.pq_search(og, cb.numpy(), codes.numpy(), q, 5, L=24, W=40, table=x)
        _check("pq traversal, beam 40 on a 24-entry list", (d2, l2), (el2, ed2))
        assert int(st2["ndis"]) == int(ost2["n_adc"]), (st2, ost2)
        # option "pq_rerank_expanded": the exact rerank over EVERY expanded node = upstream DiskANN's full_retset -- the second oracle
        # (oracle/lm_oracle_diskann.c, the transcription of cached_beam_search) with rerank_final_list_only off, ids and distances;
        # with the option off the same oracle's final-list form (= lm_oracle_pq.c) must come back
        idx.set_option("pq_rerank_expanded", 1)
        differs = 0
        for L, Wb in ((12, 2), (24, 40), (6, 1)):
            l3, d3 = idx.pq_search(q, 5, idx.make_pq_params(L, Wb))
            ui, ud, ust = orc.diskann_search(og, cb.numpy(), codes.numpy(), q, 5, L=L, W=Wb, table=x, rerank_final_list_only=False)
            _check(f"pq traversal, rerank set = expanded nodes (upstream full_retset) L={L} W={Wb}", (d3, l3), (ui, ud))
            differs += ust["n_final_differs"]
        assert differs > 0  # (on at least one query the expanded set is a strict superset of the final list: the option is exercised)
        assert idx.get_option("pq_rerank_overflow") == 0 and idx.get_option("pq_rerank_expanded") == 1
        # the option together with skip_search_reorder (no rerank follows): the PQ-ordered final list is the result -- the expanded-node
        # record must NOT replace it (round-4 advisor finding: the first k expanded nodes came back with distance 0)
        l5, d5 = idx.pq_search(q, 5, idx.make_pq_params(12, 2, skip_search_reorder=True))
        si, sd, _ = orc.pq_search(og, cb.numpy(), codes.numpy(), q, 5, L=12, W=2, skip_search_reorder=True)
        _check("pq traversal, pq_rerank_expanded + skip_search_reorder = PQ order", (d5, l5), (si, sd))
        idx.set_option("pq_rerank_expanded", 0)
        l4, d4 = idx.pq_search(q, 5, idx.make_pq_params(12, 2))
        fi, fd, _ = orc.diskann_search(og, cb.numpy(), codes.numpy(), q, 5, L=12, W=2, table=x, rerank_final_list_only=True)
        _check("pq traversal, rerank set = final list again", (d4, l4), (fi, fd))
    idx.close()


def case_pq_stock_bundle():
    """The DiskANN-style traversal over a STOCK bundle (tests/golden/stock_diskann: public DiskANN file layout, unequal PQ chunks, MIPS
    augmentation undone by leann_amd/diskann_files.py) through lm_pq_attach_chunked, against the oracle with the same chunking."""
    from leann_amd.diskann_files import load_stock_bundle
    from leann_amd.index import Mi355xIndex
    from oracle import oracle as orc

    fx = Path(__file__).resolve().parent / "golden" / "stock_diskann"
    b = load_stock_bundle(fx / "fx", 24, "mips")
    g = b.graph()
    x = b.vectors
    q = x[[3, 77, 150]] + 0.05 * np.random.default_rng(2).standard_normal((3, 24)).astype(np.float32)
    idx = Mi355xIndex.from_csr(g)
    idx.attach_table(x)
    idx.attach_pq(b.codebooks, b.codes, b.chunk_offsets)
    idx.set_option("pq_threads", 256)  # (emulation speed; see case_pq)
    og = orc.OracleGraph(g.node_offsets, g.level_ptr, g.neighbors, g.levels, g.entry_point, g.max_level, g.metric_type, 24)
    for L, W in ((16, 2), (40, 4)):
        l, d = idx.pq_search(q, 5, idx.make_pq_params(L, W))
        el, ed, _ = orc.pq_search(og, b.codebooks, b.codes, q, 5, L=L, W=W, table=x, chunk_off=b.chunk_offsets)
        _check(f"pq traversal over the stock DiskANN fixture L={L} W={W}", (d, l), (el, ed))
    gt, _ = orc.bruteforce_topk(x, q, 5, 0)
    assert np.mean([len(set(l[i].tolist()) & set(gt[i].tolist())) / 5 for i in range(3)]) >= 0.8  # L = 40: a working index
    idx.close()


def case_two_level():
    import torch

    from leann_amd.hnsw_builder import build_hnsw
    from leann_amd.index import Mi355xIndex
    from leann_amd.pq import encode_pq, train_pq
    from oracle import oracle as orc

    x, q = _data(260, 32, 31)
    g = build_hnsw(x, "mips", M=6, ef_construction=30)
    xt = torch.from_numpy(x)
    cb = train_pq(xt, 8, iters=4, seed=0)
    codes = encode_pq(xt, cb)
    idx = Mi355xIndex.from_csr(g)
    idx.attach_pq(cb.numpy(), codes.numpy())
    idx.set_provider(NumpyProvider(x, idx.info.d_padded))
    og = orc.OracleGraph(g.node_offsets, g.level_ptr, g.neighbors, g.levels, g.entry_point, g.max_level, g.metric_type, 32)
    for strategy in ("global", "local"):
        got = idx.search(q, 5, idx.make_params(ef=14, beam=2, recompute=True, prune_ratio=0.5, local_prune=(strategy == "local")))
        exp = orc.search(og, q, 5, ef=14, beam=2, table=x, pq=(cb.numpy(), codes.numpy()), prune_ratio=0.5, pruning_strategy=strategy)
        _check(f"two-level search {strategy}", got, exp[:2], idx.stats(), exp[2])
        # ... and with dynamic batching on top (the extra pops count the fresh neighbours BEFORE pruning; k_prune finishes the longer list)
        got = idx.search(q, 5, idx.make_params(ef=14, beam=2, recompute=True, prune_ratio=0.5, local_prune=(strategy == "local"), batch_size=20))
        exp = orc.search(og, q, 5, ef=14, beam=2, table=x, pq=(cb.numpy(), codes.numpy()), prune_ratio=0.5, pruning_strategy=strategy, batch_size=20)
        _check(f"two-level search {strategy} + batch_size 20", got, exp[:2], idx.stats(), exp[2])
        assert int(idx.stats()["nadc"]) == exp[2]["nadc"] and int(idx.stats()["nrounds"]) == exp[2]["nrounds"]
    idx.close()


def case_degenerate_graphs(n_graphs=12):
    """Seeded random tiny graphs with empty neighbour lists, unreachable nodes, upper-level stubs, k > reachable set
    (unfilled result slots): the same generator as tests/test_gpu_parity.py::test_random_degenerate_graphs_match_oracle."""
    from leann_amd.index import Mi355xIndex
    from oracle import oracle as orc
    from tests.test_oracle_properties import _random_graph
    from tests.util import oracle_graph

    for seed in range(n_graphs):
        rng = np.random.default_rng(1000 + seed)
        n = int(rng.integers(1, 30))
        metric = int(rng.integers(0, 2))
        x, g, _ = _random_graph(rng, n, 64, metric)
        q = rng.standard_normal((3, 64)).astype(np.float32)
        k, ef, beam = int(rng.integers(1, 7)), int(rng.integers(1, n + 5)), int(rng.integers(1, 5))
        oi, od, ost = orc.search(oracle_graph(g, 64), q, k, ef=ef, beam=beam, table=x)
        idx = Mi355xIndex.from_csr(g)
        idx.attach_table(x)
        idx.set_provider(NumpyProvider(x, idx.info.d_padded))
        for mode in ("persistent", "lockstep", "provider"):
            idx.set_option("persistent_table", 1 if mode == "persistent" else 0)
            d, l = idx.search(q, k, idx.make_params(ef=ef, beam=beam, recompute=(mode == "provider")))
            st = idx.stats()
            ok = (np.array_equal(l, oi) and np.array_equal(d.view(np.uint32), od.view(np.uint32))
                  and (st["ndis"], st["nexpand"], st["nrounds"]) == (ost["ndis"], ost["nexpand"], ost["nrounds"]))
            assert ok, ("degenerate", seed, mode, n, k, ef, beam, st, ost)
        idx.close()
    print(f"degenerate graphs x{n_graphs} (persistent / lock-step / provider): ok", flush=True)


def case_hub_cache_and_helpers():
    """Hub-embedding cache (results identical to plain recompute, fewer provider rows), lm_topk_merge, lm_dist_gather."""
    from leann_amd import _lib
    from leann_amd.backend import hub_nodes
    from leann_amd.hnsw_builder import build_hnsw
    from leann_amd.index import Mi355xIndex
    from oracle import oracle as orc

    lib = _lib.load()
    x, q = _data(240, 64, 41)
    g = build_hnsw(x, "mips", M=6, ef_construction=30)
    og = orc.OracleGraph(g.node_offsets, g.level_ptr, g.neighbors, g.levels, g.entry_point, g.max_level, g.metric_type, 64)
    idx = Mi355xIndex.from_csr(g)
    prov = NumpyProvider(x, 64)
    rows = {"n": 0}
    base_call = prov.__call__

    def counting(d_ids, n, stream):
        rows["n"] += n
        return base_call(d_ids, n, stream)

    idx.set_provider(counting)
    exp = orc.search(og, q, 5, ef=14, beam=2, table=x)
    got = idx.search(q, 5, idx.make_params(ef=14, beam=2, recompute=True))
    plain_rows = rows["n"]
    _check("recompute (for the hub comparison)", got, exp[:2], idx.stats(), exp[2])
    hubs = hub_nodes(g, 0.2)
    emb = np.ascontiguousarray(x[hubs])
    _lib.check(lib.lm_index_set_hub_cache(idx._h, hubs.ctypes.data_as(C.c_void_p), len(hubs), emb.ctypes.data_as(C.c_void_p)), "hub")
    rows["n"] = 0
    got = idx.search(q, 5, idx.make_params(ef=14, beam=2, recompute=True))
    _check("recompute with a 20 % hub cache", got, exp[:2], idx.stats(), exp[2])
    assert 0 < rows["n"] < plain_rows, (rows, plain_rows)
    idx.close()
    # per-shard top-k merge == oracle merge
    rng = np.random.default_rng(5)
    S, B, k = 3, 4, 5
    ids = rng.permutation(1000)[: S * B * k].reshape(S, B, k).astype(np.int64)
    ids[0, 0, 3:] = -1
    dist = np.sort(rng.standard_normal((S, B, k)).astype(np.float32), axis=2)[:, :, ::-1].copy()
    dist[0, 0, 3:] = -np.inf
    oi, od = np.empty((B, k), np.int64), np.empty((B, k), np.float32)
    _lib.check(lib.lm_topk_merge(ids.ctypes.data_as(C.c_void_p), dist.ctypes.data_as(C.c_void_p), S, B, k, 0,
                                 oi.ctypes.data_as(C.c_void_p), od.ctypes.data_as(C.c_void_p), None), "merge")
    ei, ed = orc.merge_topk(ids, dist, 0)
    assert np.array_equal(oi, ei) and np.array_equal(od, ed)
    # pair distances == canonical oracle distance
    qi = rng.integers(0, q.shape[0], 50).astype(np.int32)
    vi = rng.integers(0, x.shape[0], 50).astype(np.int32)
    out = np.empty(50, np.float32)
    _lib.check(lib.lm_dist_gather(x.ctypes.data_as(C.c_void_p), 0, 64, 0, q.ctypes.data_as(C.c_void_p), qi.ctypes.data_as(C.c_void_p),
                                  vi.ctypes.data_as(C.c_void_p), 50, out.ctypes.data_as(C.c_void_p), None), "dist")
    ref = np.array([orc.dist(q[a], x[b], 0) for a, b in zip(qi, vi)], np.float32)
    assert np.array_equal(out, ref), np.abs(out - ref).max()
    print("hub cache / top-k merge / pair distances: ok", flush=True)


def case_dims_and_batches():
    """The padded-dimension instantiations the bench / configs use (NCH = 6, 12, 16) and a search split over several passes."""
    from leann_amd.hnsw_builder import build_hnsw
    from leann_amd.index import Mi355xIndex
    from oracle import oracle as orc

    for d in (384, 768, 1024):
        x, q = _data(120, d, 50 + d, nq=5)
        g = build_hnsw(x, "mips", M=6, ef_construction=30)
        og = orc.OracleGraph(g.node_offsets, g.level_ptr, g.neighbors, g.levels, g.entry_point, g.max_level, g.metric_type, d)
        exp = orc.search(og, q, 5, ef=12, beam=2, table=x)
        idx = Mi355xIndex.from_csr(g)
        idx.attach_table(x)
        idx.set_provider(NumpyProvider(x, idx.info.d_padded))
        for mode, kw in (("persistent", dict(recompute=False)), ("recompute, 3 passes of <= 2 queries", dict(recompute=True, max_batch=2))):
            got = idx.search(q, 5, idx.make_params(ef=12, beam=2, **kw))
            ok = np.array_equal(got[1], exp[0]) and np.array_equal(got[0], exp[1]) and int(idx.stats()["ndis"]) == int(exp[2]["ndis"])
            assert ok, (d, mode)
        idx.close()
    print("D = 384 / 768 / 1024, multi-pass batches: ok", flush=True)


def case_encoder_abi():
    """The encoder entry points through the C ABI (launchers included: argument checks, grids, LDS attributes, the
    environment-selected kernel generations) on host buffers, against numpy references."""
    import os

    import torch
    from scipy.special import erf

    from leann_amd import _lib
    from leann_amd.encoder import pack_w2_fused_mlp

    lib = _lib.load()
    rng = np.random.default_rng(7)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    T, H = 130, 384
    x = rng.standard_normal((T, H)).astype(np.float16)
    gamma = (1 + 0.1 * rng.standard_normal(H)).astype(np.float16)
    beta = (0.1 * rng.standard_normal(H)).astype(np.float16)

    def ln(z):
        mu = z.mean(1, keepdims=True)
        var = ((z - mu) ** 2).mean(1, keepdims=True)
        return (z - mu) / np.sqrt(var + 1e-12) * gamma.astype(np.float64) + beta.astype(np.float64)

    b2 = (0.2 * rng.standard_normal(H)).astype(np.float32)
    w = (rng.standard_normal((3 * H, H)) / np.sqrt(H)).astype(np.float16)
    b = (0.2 * rng.standard_normal(3 * H)).astype(np.float32)
    res = rng.standard_normal((T, H)).astype(np.float16)
    wo, bo = np.ascontiguousarray(w[:H]), np.ascontiguousarray(b[:H])
    # weight-stationary form (lm_gemm_ws_h384.hip): plain nn.Linear weight layout, 8 waves per workgroup; ragged token count
    Tw = 300
    xw = rng.standard_normal((Tw, H)).astype(np.float16)
    for wmat, bvec in ((w, b), (wo, bo)):
        ow = np.zeros((Tw, wmat.shape[0]), np.float16)
        _lib.check(lib.lm_gemm_ws_h384_f16(vp(xw), vp(np.ascontiguousarray(wmat)), vp(bvec), wmat.shape[0], vp(ow), Tw, None), "gemm_ws")
        assert np.abs(ow.astype(np.float64) - (xw.astype(np.float64) @ wmat.astype(np.float64).T + bvec)).max() < 6e-3, wmat.shape
    assert lib.lm_gemm_ws_h384_f16(vp(xw), vp(w), vp(b), 200, vp(ow), Tw, None) == -1  # n_out % 192
    # weight-streaming form (lm_qkv_h384.hip: x read once, W through a four-stage ring, 8 waves = two per SIMD): ragged token counts, several workgroups
    for Tq, Nq in ((300, 1152), (513, 256), (31, 384)):
        xq = rng.standard_normal((Tq, H)).astype(np.float16)
        wq = (rng.standard_normal((Nq, H)) / np.sqrt(H)).astype(np.float16)
        bq = (0.3 * rng.standard_normal(Nq)).astype(np.float32)
        wqi = np.zeros_like(wq)
        _lib.check(lib.lm_qkv_pack_h384(vp(wq), Nq, vp(wqi), None), "qkv image")
        oq = np.full((Tq, Nq), 7.0, np.float16)
        _lib.check(lib.lm_qkv_h384_f16(vp(xq), vp(wqi), vp(bq), Nq, vp(oq), Tq, None), "qkv")
        assert np.abs(oq.astype(np.float64) - (xq.astype(np.float64) @ wq.astype(np.float64).T + bq)).max() < 6e-3, (Tq, Nq)
    assert lib.lm_qkv_h384_f16(vp(xq), vp(wqi), vp(bq), 200, vp(oq), Tq, None) == -1  # n_out % 128
    # attention output projection + LayerNorm + feed-forward block + LayerNorm in one kernel (lm_layer_tail_h384.hip: k_layer_tail_h384)
    from leann_amd.encoder import pack_w1_acc_order, pack_wo_slabs

    gamma1 = (1 + 0.1 * rng.standard_normal(H)).astype(np.float16)
    beta1 = (0.1 * rng.standard_normal(H)).astype(np.float16)

    def ln_with(z, gm, bt):
        mu = z.mean(1, keepdims=True)
        var = ((z - mu) ** 2).mean(1, keepdims=True)
        return (z - mu) / np.sqrt(var + 1e-12) * gm.astype(np.float64) + bt.astype(np.float64)

    # weight images, alternating products; ffn a multiple of 192; ragged token counts
    for Tt, F4 in ((130, 192), (257, 384)):
        at = rng.standard_normal((Tt, H)).astype(np.float16)
        rs = rng.standard_normal((Tt, H)).astype(np.float16)
        w1t = (rng.standard_normal((F4, H)) / np.sqrt(H)).astype(np.float16)
        w2t = (rng.standard_normal((H, F4)) / np.sqrt(F4)).astype(np.float16)
        b1t = (0.2 * rng.standard_normal(F4)).astype(np.float32)
        x1 = ln_with(rs.astype(np.float64) + at.astype(np.float64) @ wo.astype(np.float64).T + bo, gamma1, beta1).astype(np.float16)
        hidt = x1.astype(np.float64) @ w1t.astype(np.float64).T + b1t
        p16t = (0.5 * hidt * (1 + erf(hidt / np.sqrt(2)))).astype(np.float16).astype(np.float64)
        reft = ln(p16t @ w2t.astype(np.float64).T + b2 + x1.astype(np.float64))
        wos = pack_wo_slabs(torch.from_numpy(wo)).numpy()
        w1a = pack_w1_acc_order(torch.from_numpy(w1t)).numpy()
        w2pt = pack_w2_fused_mlp(torch.from_numpy(w2t)).numpy()
        woi, w1i, w2i = np.zeros_like(wos), np.zeros_like(w1a), np.zeros_like(w2pt)
        _lib.check(lib.lm_layer_tail_pack_h384(vp(wos), vp(w1a), vp(w2pt), F4, vp(woi), vp(w1i), vp(w2i), None), "tail images")
        for src, img in ((wos, woi), (w1a, w1i), (w2pt, w2i)):  # an image is a permutation of 16-byte chunks
            assert np.array_equal(np.sort(src.view(np.uint64).reshape(-1, 2), axis=0), np.sort(img.view(np.uint64).reshape(-1, 2), axis=0))
        o4 = np.zeros((Tt, H), np.float16)
        _lib.check(lib.lm_layer_tail_h384_f16(vp(at), vp(rs), vp(woi), vp(bo), vp(gamma1), vp(beta1), 1e-12, vp(w1i), vp(b1t), vp(w2i), vp(b2),
                                              vp(gamma), vp(beta), vp(o4), Tt, F4, 1e-12, None), "tail4")
        err4 = np.abs(o4.astype(np.float64) - reft).max()
        assert err4 < 1.2e-2, (Tt, F4, err4)
        if F4 % 128 == 0:  # the unfused form (LEANN_MI355X_TAIL=0) through the same library: agreement to fp16 rounding of the intermediates
            y0, x1k, hm, y2, o5 = (np.zeros((Tt, n), np.float16) for n in (H, H, F4, H, H))
            _lib.check(lib.lm_gemm_ws_h384_f16(vp(at), vp(np.ascontiguousarray(wo)), vp(bo), H, vp(y0), Tt, None), "gemm_ws")
            _lib.check(lib.lm_add_layernorm_f16(vp(y0), vp(rs), vp(gamma1), vp(beta1), vp(x1k), Tt, H, 1e-12, None), "ln")
            _lib.check(lib.lm_gemm_f16(vp(x1k), vp(w1t), vp(b1t), None, 1, F4, H, vp(hm), Tt, None), "fc1")
            _lib.check(lib.lm_gemm_f16(vp(hm), vp(w2t), vp(b2), vp(x1k), 2, H, F4, vp(y2), Tt, None), "fc2")
            _lib.check(lib.lm_add_layernorm_f16(vp(y2), None, vp(gamma), vp(beta), vp(o5), Tt, H, 1e-12, None), "ln2")
            assert np.abs(o5.astype(np.float64) - o4.astype(np.float64)).max() < 2e-2, (Tt, F4)
    assert lib.lm_layer_tail_h384_f16(vp(at), vp(rs), vp(woi), vp(bo), vp(gamma1), vp(beta1), 1e-12, vp(w1i), vp(b1t), vp(w2i), vp(b2),
                                      vp(gamma), vp(beta), vp(o4), Tt, 224, 1e-12, None) == -1  # ffn % 192
    # add + LayerNorm: both generations through the same entry point
    for gen in ("1", "2"):
        os.environ["LEANN_MI355X_LN"] = gen
        o = np.zeros((T, H), np.float16)
        _lib.check(lib.lm_add_layernorm_f16(vp(x), vp(res), vp(gamma), vp(beta), vp(o), T, H, 1e-12, None), "ln")
        assert np.abs(o.astype(np.float64) - ln(x.astype(np.float64) + res.astype(np.float64))).max() < 4e-3, gen
    os.environ.pop("LEANN_MI355X_LN")
    # attention; mean pooling
    heads, lens = 2, np.array([70, 1, 33, 64], np.int32)
    cu = np.zeros(5, np.int32)
    cu[1:] = np.cumsum(lens)
    tot, Hh = int(cu[-1]), heads * 32
    qkv = (1.5 * rng.standard_normal((tot, 3 * Hh))).astype(np.float16)
    q3 = qkv.astype(np.float64).reshape(tot, 3, heads, 32)
    refa = np.zeros((tot, Hh))
    for i in range(4):
        a_, b_ = cu[i], cu[i + 1]
        for h in range(heads):
            sc = q3[a_:b_, 0, h] @ q3[a_:b_, 1, h].T / np.sqrt(32)
            pr = np.exp(sc - sc.max(1, keepdims=True))
            refa[a_:b_, h * 32:(h + 1) * 32] = (pr / pr.sum(1, keepdims=True)) @ q3[a_:b_, 2, h]
    oa = np.zeros((tot, Hh), np.float16)
    _lib.check(lib.lm_attn_varlen_hd32_f16(vp(qkv), vp(cu), 4, heads, int(lens.max()), vp(oa), None), "attn")
    assert np.abs(oa.astype(np.float64) - refa).max() < 4e-3
    xs = rng.standard_normal((tot, H)).astype(np.float16)
    po = np.zeros((4, H), np.float32)
    _lib.check(lib.lm_meanpool_varlen_f16(vp(xs), vp(cu), 4, H, 1, vp(po), None), "pool")
    m = np.stack([xs[cu[i]:cu[i + 1]].astype(np.float64).mean(0) for i in range(4)])
    assert np.abs(po - m / np.linalg.norm(m, axis=1, keepdims=True)).max() < 1e-5
    print("encoder entry points through the C ABI: ok", flush=True)


def case_attention_v3():
    """Generation 3 of the head_dim-32 attention kernel (csrc/lm_attn_v3.hip: K / V by LDS-DMA, transposing V reads, the running maximum
    as the score MFMA's C operand, deferred rescaling) against float64 numpy and against
    generation 2 (LEANN_MI355X_ATTN=2).  Inputs that FORCE the rescale branch (cdna_hip_programming.md T13: the branch is rare and data
    dependent -- a passing check on bounded random data says nothing): a key in a LATER tile whose score exceeds every earlier one by far
    (threshold 8 in log2 units), per head and only for some query rows; rows whose maximum sits in the masked last tile; lengths 1, 31,
    32, 33, 64, 65 and 200 (two query blocks per wave, seven key tiles)."""
    import os

    from leann_amd import _lib

    lib = _lib.load()
    vp = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
    rng = np.random.default_rng(77)
    heads = 2
    for lens in (np.array([200, 1, 33], np.int32), np.array([31, 65, 32, 64], np.int32)):
        n = lens.shape[0]
        cu = np.zeros(n + 1, np.int32)
        cu[1:] = np.cumsum(lens)
        tot, Hh = int(cu[-1]), heads * 32
        qkv = (1.2 * rng.standard_normal((tot, 3 * Hh))).astype(np.float32)
        # spikes: in sequence 0 / head 0, key 170 (tile 5) is strongly aligned with query rows 3 and 40; key 199 (the masked last tile) with row 77;
        # in the second batch, key 64 (the one-key last tile of the 65-token sequence) with its query row 0
        def spike(tok_q, tok_k, h, gain):
            qkv[tok_k, Hh + 32 * h: Hh + 32 * h + 32] = gain * qkv[tok_q, 32 * h: 32 * h + 32]
        if lens[0] == 200:
            spike(3, 170, 0, 3.0); spike(40, 170, 0, 3.0); spike(77, 199, 0, 4.0); spike(150, 40, 1, 3.5)
        else:
            spike(31, 31 + 64, 0, 4.0); spike(31 + 10, 31 + 33, 1, 3.0)
        qkv = qkv.astype(np.float16)
        q3 = qkv.astype(np.float64).reshape(tot, 3, heads, 32)
        ref = np.zeros((tot, Hh))
        grew = 0
        for i in range(n):
            a_, b_ = cu[i], cu[i + 1]
            for h in range(heads):
                sc = q3[a_:b_, 0, h] @ q3[a_:b_, 1, h].T / np.sqrt(32)
                pr = np.exp(sc - sc.max(1, keepdims=True))
                ref[a_:b_, h * 32:(h + 1) * 32] = (pr / pr.sum(1, keepdims=True)) @ q3[a_:b_, 2, h]
                s2 = sc * 1.4426950408889634
                if s2.shape[1] > 32:  # rows whose later tiles exceed the first tile's maximum by more than the threshold: the branch is exercised
                    grew += int(((s2[:, 32:].max(1) - s2[:, :32].max(1)) > 8.0).sum())
        assert grew > 0, "the test data does not reach the rescale branch"
        outs = {}
        for var in ("0",):
            o = np.zeros((tot, Hh), np.float16)
            _lib.check(lib.lm_attn_varlen_hd32_f16(vp(qkv), vp(cu), n, heads, int(lens.max()), vp(o), None), "attn v3")
            err = np.abs(o.astype(np.float64) - ref).max()
            assert err < 4e-3, (var, lens.tolist(), err)
            outs[var] = o
        os.environ["LEANN_MI355X_ATTN"] = "2"
        o2 = np.zeros((tot, Hh), np.float16)
        _lib.check(lib.lm_attn_varlen_hd32_f16(vp(qkv), vp(cu), n, heads, int(lens.max()), vp(o2), None), "attn v2")
        os.environ.pop("LEANN_MI355X_ATTN")
        assert np.abs(o2.astype(np.float64) - ref).max() < 4e-3
        assert np.abs(o2.astype(np.float32) - outs["0"].astype(np.float32)).max() < 4e-3
        print(f"attention generation 3, lengths {lens.tolist()}: max |err| vs float64 {max(np.abs(v.astype(np.float64) - ref).max() for v in outs.values()):.2e}, {grew} rows through the rescale branch: ok", flush=True)


def case_qkv_attn_fused(lens_list=((70, 1, 33), (256, 31), (150, 200))):
    """The QKV projection fused into attention (csrc/lm_qkv_attn_h384.hip, round 6) against float64 numpy: x [T][384] -> attention output
    [T][384] for 12 heads x 32, with Q, K, V rounded where the kernel rounds them (Q after the softmax scale, once; K and V to fp16).  Weights
    scaled so that scores spread over many powers of two: later key tiles exceed the running maximum by more than the deferred-rescaling
    threshold for many rows (counted: the branch must be exercised), and the last tile is masked at lengths that are not multiples of 32.
    Lengths 1, 31, 33, 70, 256: one to eight active waves, idle waves that only stream weights, a full workgroup.  Also against the
    stand-alone pair (lm_qkv_h384_f16 -> lm_attn_varlen_hd32_f16) on the same operands."""
    from leann_amd import _lib

    lib = _lib.load()
    vp = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
    rng = np.random.default_rng(91)
    H, heads = 384, 12
    w = rng.standard_normal((3 * H, H)).astype(np.float32)
    w[: 2 * H] *= 0.26  # q, k entries ~ N(0, 2.5^2): scores ~ N(0, (6.3 / sqrt 32 x 5.6 ...)) -- wide enough for the rescale branch
    w[2 * H:] *= 0.1
    w = w.astype(np.float16)
    b = (0.3 * rng.standard_normal(3 * H)).astype(np.float32)
    img = np.zeros_like(w)
    _lib.check(lib.lm_qkv_pack_h384(vp(w), 3 * H, vp(img), None), "lm_qkv_pack_h384")
    c = 1.4426950408889634 / np.sqrt(32.0)
    for lens in lens_list:
        lens = np.asarray(lens, np.int32)
        n = lens.shape[0]
        cu = np.zeros(n + 1, np.int32)
        cu[1:] = np.cumsum(lens)
        tot = int(cu[-1])
        x = (0.5 * rng.standard_normal((tot, H))).astype(np.float16)
        qkv = x.astype(np.float64) @ w.astype(np.float64).T + b.astype(np.float64)
        q2 = (qkv[:, :H] * c).astype(np.float16).astype(np.float64)  # (log2 units)
        k = qkv[:, H: 2 * H].astype(np.float16).astype(np.float64)
        v = qkv[:, 2 * H:].astype(np.float16).astype(np.float64)
        ref = np.zeros((tot, H))
        grew = 0
        for i in range(n):
            a_, b_ = cu[i], cu[i + 1]
            for h in range(heads):
                s2 = q2[a_:b_, 32 * h: 32 * h + 32] @ k[a_:b_, 32 * h: 32 * h + 32].T
                pr = np.exp2(s2 - s2.max(1, keepdims=True))
                ref[a_:b_, 32 * h: 32 * h + 32] = (pr / pr.sum(1, keepdims=True)) @ v[a_:b_, 32 * h: 32 * h + 32]
                if s2.shape[1] > 32:
                    grew += int(((s2[:, 32:].max(1) - s2[:, :32].max(1)) > 8.0).sum())
        assert max(lens) <= 32 or grew > 0, "the test data does not reach the rescale branch"
        o = np.zeros((tot, H), np.float16)
        _lib.check(lib.lm_qkv_attn_h384_f16(vp(x), vp(img), vp(b), vp(cu), n, int(lens.max()), tot, vp(o), None), "lm_qkv_attn_h384_f16")
        err = np.abs(o.astype(np.float64) - ref).max()
        assert np.isfinite(o.astype(np.float32)).all() and err < 6e-3, (lens.tolist(), err)
        # the stand-alone pair on the same operands (Q rounded twice there: fp16-close, not bit-equal)
        qkv16 = np.zeros((tot, 3 * H), np.float16)
        _lib.check(lib.lm_qkv_h384_f16(vp(x), vp(img), vp(b), 3 * H, vp(qkv16), tot, None), "lm_qkv_h384_f16")
        o2 = np.zeros((tot, H), np.float16)
        _lib.check(lib.lm_attn_varlen_hd32_f16(vp(qkv16), vp(cu), n, heads, int(lens.max()), vp(o2), None), "attn v3")
        pair = np.abs(o2.astype(np.float64) - o.astype(np.float64)).max()
        assert pair < 3e-2, (lens.tolist(), pair)
        print(f"fused QKV + attention, lengths {lens.tolist()}: max |err| vs float64 {err:.2e}, vs the stand-alone pair {pair:.2e}, {grew} rows through the rescale branch: ok", flush=True)


def case_layer_tail_small():
    """The fused layer tail alone (small enough for the ThreadSanitizer build): every LDS stage hand-over of lm_layer_tail_h384.hip -- the
    six-stage W_o ring with the residual rows behind it, the W1 / W2 rings, the continuous fragment ring, the output tiles -- with real
    threads per lane."""
    import torch
    from scipy.special import erf

    from leann_amd import _lib
    from leann_amd.encoder import pack_w1_acc_order, pack_w2_fused_mlp, pack_wo_slabs

    lib = _lib.load()
    rng = np.random.default_rng(11)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    T, F, H = 161, 192, 384
    f64 = np.float64

    def ln(z, gm, bt):
        mu = z.mean(1, keepdims=True)
        var = ((z - mu) ** 2).mean(1, keepdims=True)
        return (z - mu) / np.sqrt(var + 1e-12) * gm.astype(f64) + bt.astype(f64)

    def mlp(xh, w1, b1, w2, b2, gm, bt):
        hid = xh.astype(f64) @ w1.astype(f64).T + b1
        p16 = (0.5 * hid * (1 + erf(hid / np.sqrt(2)))).astype(np.float16).astype(f64)
        return ln(p16 @ w2.astype(f64).T + b2 + xh.astype(f64), gm, bt)

    at, rs = rng.standard_normal((T, H)).astype(np.float16), rng.standard_normal((T, H)).astype(np.float16)
    g1, g2 = [(1 + 0.1 * rng.standard_normal(H)).astype(np.float16) for _ in range(2)]
    be1, be2 = [(0.1 * rng.standard_normal(H)).astype(np.float16) for _ in range(2)]
    wo = (rng.standard_normal((H, H)) / np.sqrt(H)).astype(np.float16)
    w1 = (rng.standard_normal((F, H)) / np.sqrt(H)).astype(np.float16)
    w2 = (rng.standard_normal((H, F)) / np.sqrt(F)).astype(np.float16)
    bo, b1, b2 = [(0.2 * rng.standard_normal(n)).astype(np.float32) for n in (H, F, H)]
    srcs = (pack_wo_slabs(torch.from_numpy(wo)).numpy(), pack_w1_acc_order(torch.from_numpy(w1)).numpy(), pack_w2_fused_mlp(torch.from_numpy(w2)).numpy())
    imgs = tuple(np.zeros_like(s) for s in srcs)
    _lib.check(lib.lm_layer_tail_pack_h384(*(vp(s) for s in srcs), F, *(vp(i) for i in imgs), None), "tail images")
    x1 = ln(rs.astype(f64) + at.astype(f64) @ wo.astype(f64).T + bo, g1, be1).astype(np.float16)
    ot = np.zeros((T, H), np.float16)
    _lib.check(lib.lm_layer_tail_h384_f16(vp(at), vp(rs), vp(imgs[0]), vp(bo), vp(g1), vp(be1), 1e-12, vp(imgs[1]), vp(b1), vp(imgs[2]), vp(b2), vp(g2),
                                          vp(be2), vp(ot), T, F, 1e-12, None), "tail")
    assert np.abs(ot.astype(f64) - mlp(x1, w1, b1, w2, b2, g2, be2)).max() < 1.2e-2
    print("fused layer tail (small): ok", flush=True)


CASES = {
    "layer_tail_small": case_layer_tail_small,
    "table_mips": lambda: case_table("mips", 64),
    "table_l2_d100": lambda: case_table("l2", 100),
    "table_f16": lambda: case_table("mips", 64, f16=True),
    "recompute": lambda: case_recompute(0, memo=False),
    "recompute_memo": lambda: case_recompute(0, memo=True),
    "recompute_default_is_memo": lambda: case_recompute(0, memo=None, nq=6),
    "recompute_memo_grows": lambda: case_recompute(0, memo=None, initial_rows=4, nq=6),
    "recompute_one_query_skips_memo": lambda: case_recompute(0, memo=True, nq=1),
    "recompute_wave_variant": lambda: case_recompute(3),
    "stop_rules": case_stop_rules,
    "dynamic_batching": case_dynamic_batching,
    "speculative_prefetch": case_speculative_prefetch,
    "single_query_direct": case_single_query_direct,
    "pq_deferred": lambda: case_pq(True),
    "pq_table": lambda: case_pq(False),
    "two_level": case_two_level,
    "pq_stock_bundle": case_pq_stock_bundle,
    "degenerate_graphs": case_degenerate_graphs,
    "hub_cache_and_helpers": case_hub_cache_and_helpers,
    "encoder_abi": case_encoder_abi,
    "dims_and_batches": case_dims_and_batches,
    "attention_v3": case_attention_v3,
    "qkv_attn_fused": case_qkv_attn_fused,
}

def case_gemm_f16():
    """lm_gemm_f16 (csrc/lm_gemm_f16.hip: the general linear layer of the hidden-768 path) vs numpy: both tile shapes, every
    epilogue, ragged token counts (clamped row loads, masked stores), more than eight row blocks (XCD-order padding)."""
    import ctypes as C

    from leann_amd import _lib
    from scipy.special import erf

    lib = _lib.load()
    rng = np.random.default_rng(17)

    def run(t, n, k, epi):
        x = (rng.standard_normal((t, k)) * 0.5).astype(np.float16)
        w = (rng.standard_normal((n, k)) * 0.1).astype(np.float16)
        b = rng.standard_normal(n).astype(np.float32)
        res = rng.standard_normal((t, n)).astype(np.float16)
        out = np.full((t, n), 7.0, np.float16)
        vp = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
        _lib.check(lib.lm_gemm_f16(vp(x), vp(w), vp(b), vp(res) if epi & 2 else None, epi, n, k, vp(out), t, None), "lm_gemm_f16")
        ref = x.astype(np.float32) @ w.astype(np.float32).T + b
        if epi & 1:
            ref = 0.5 * ref * (1.0 + erf(ref / np.sqrt(2.0)))
        ref = ref.astype(np.float16)
        if epi & 2:
            ref = (ref.astype(np.float32) + res.astype(np.float32)).astype(np.float16)
        err = float(np.abs(out.astype(np.float32) - ref.astype(np.float32)).max())
        tol = 4e-3 * max(1.0, float(np.abs(ref.astype(np.float32)).max()))
        print(f"gemm_f16 T={t} N={n} K={k} epilogue={epi}: max|diff| {err:.2e} (tol {tol:.2e}) {'ok' if err <= tol else 'MISMATCH'}", flush=True)
        assert err <= tol

    for epi in (0, 1, 2, 3):
        run(300, 256, 128, epi)      # 256 x 256 tiles, two row blocks, the second one ragged
    run(257, 512, 256, 3)            # two column tiles, 1 token in the last row block, four K-tiles
    run(130, 384, 128, 0)            # 128 x 128 tiles (N % 256 != 0)
    run(100, 256, 128, 2)            # short launch: small tiles although N % 256 == 0
    run(1200, 128, 128, 1)           # ten row blocks of 128: the last group of eight is padded
    run(300, 1152, 128, 2)           # 256-wide tiles with a half-empty last column tile (the QKV width of the 384-wide models)
    run(700, 768, 768, 2)            # twelve K-tiles, three column tiles, tile lists of several tiles per workgroup (the out-projection of a 768-wide model)
    run(200, 384, 384, 3)            # six K-tiles on the 128 x 128 shape
    try:
        lib.lm_gemm_f16.restype = C.c_int
        x = np.zeros((4, 100), np.float16)
        rc = lib.lm_gemm_f16(C.c_void_p(x.ctypes.data), C.c_void_p(x.ctypes.data), C.c_void_p(x.ctypes.data), None, 0, 128, 100, C.c_void_p(x.ctypes.data), 4, None)
        assert rc == _lib.LM_EINVAL  # k_in not a multiple of 128
    finally:
        pass


CASES["gemm_f16"] = case_gemm_f16


def case_hidden768():
    """The hidden-768 path (bge-base / contriever shape: 12 heads x 64): lm_attn_varlen_f16 at head_dim 64 vs numpy, then
    leann_amd/encoder.py's packed forward on the general kernels (lm_gemm_f16 x 4, attention, lm_add_layernorm_f16 x 2 per layer; no
    library GEMM left) vs the fp32 torch forward of the same weights."""
    import ctypes as C
    import os
    from unittest import mock

    import torch

    from leann_amd import _lib
    from leann_amd.encoder import BertEncoder, EncoderConfig

    lib = _lib.load()
    rng = np.random.default_rng(23)
    heads, hd = 3, 64
    lens = np.array([70, 1, 33, 64], np.int32)  # three 32-key tiles with a ragged tail, a single token, exact tiles
    cu = np.zeros(5, np.int32)
    cu[1:] = np.cumsum(lens)
    tot, Hh = int(cu[-1]), heads * hd
    qkv = (rng.standard_normal((tot, 3 * Hh)) * 1.2).astype(np.float16)
    q3 = qkv.astype(np.float64).reshape(tot, 3, heads, hd)
    ref = np.zeros((tot, Hh))
    for i in range(4):
        a, b = cu[i], cu[i + 1]
        for h in range(heads):
            sc = q3[a:b, 0, h] @ q3[a:b, 1, h].T / np.sqrt(hd)
            pr = np.exp(sc - sc.max(1, keepdims=True))
            ref[a:b, h * hd:(h + 1) * hd] = (pr / pr.sum(1, keepdims=True)) @ q3[a:b, 2, h]
    out = np.zeros((tot, Hh), np.float16)
    vp = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
    _lib.check(lib.lm_attn_varlen_f16(vp(qkv), vp(cu), 4, heads, hd, int(lens.max()), vp(out), None), "lm_attn_varlen_f16")
    err = float(np.abs(out.astype(np.float64) - ref).max())
    print(f"attention head_dim 64: max|diff| vs numpy {err:.2e}", flush=True)
    assert err < 4e-3
    out32 = np.zeros((tot, Hh), np.float16)  # the same entry point at head_dim 32 = the hd32 kernel (6 heads x 32 on the same buffer)
    _lib.check(lib.lm_attn_varlen_f16(vp(qkv), vp(cu), 4, 6, 32, int(lens.max()), vp(out32), None), "lm_attn_varlen_f16")
    ref32 = np.zeros((tot, Hh), np.float16)
    _lib.check(lib.lm_attn_varlen_hd32_f16(vp(qkv), vp(cu), 4, 6, int(lens.max()), vp(ref32), None), "lm_attn_varlen_hd32_f16")
    assert np.array_equal(out32, ref32)

    torch.manual_seed(0)
    cfg = EncoderConfig(vocab_size=300, hidden=768, layers=2, heads=12, ffn=256, max_pos=64, max_seq_length=40, pooling="cls", normalize=True)
    e32 = BertEncoder.random_init(cfg, 7).eval()
    with torch.no_grad():
        for L in e32.layers:  # biases and LayerNorm parameters that are not the identity
            for lin in (L.qkv, L.out, L.fc1, L.fc2):
                lin.bias.copy_(0.1 * torch.randn(lin.bias.shape))
            for ln in (L.ln1, L.ln2):
                ln.weight.copy_(1 + 0.1 * torch.randn(768))
                ln.bias.copy_(0.1 * torch.randn(768))
    e16 = BertEncoder.random_init(cfg, 7).eval()
    e16.load_state_dict(e32.state_dict())
    e16 = e16.half()
    n, t = 7, 40
    lens2 = rng.integers(1, t + 1, n).astype(np.int32)
    lens2[0], lens2[1] = t, 1
    ids = np.zeros((n, t), np.int32)
    for i in range(n):
        ids[i, : lens2[i]] = rng.integers(1, cfg.vocab_size, lens2[i])
    ti, tl = torch.from_numpy(ids), torch.from_numpy(lens2)
    with torch.no_grad():
        ref_e = e32(ti, tl).float()

    class _Stream:
        cuda_stream = 0

    used = []
    real_check = _lib.check

    def recording_check(rc, what=""):
        used.append(what)
        return real_check(rc, what)

    env = {k: v for k, v in os.environ.items() if not k.startswith("LEANN_MI355X_")}
    with mock.patch.object(torch.Tensor, "is_cuda", new=property(lambda self: True)), \
            mock.patch("torch.cuda.current_stream", new=lambda *a, **k: _Stream()), mock.patch.dict(os.environ, env, clear=True), \
            mock.patch.object(_lib, "check", new=recording_check):
        with torch.no_grad():
            one = e16.encode_tokens_packed(ti, tl, 4096)  # the default: the whole forward as ONE library call (lm_bert_forward_packed)
        assert "lm_bert_forward_packed" in used and "lm_gemm_f16" not in used, sorted(set(used))
        used.clear()
        with mock.patch.dict(os.environ, {"LEANN_MI355X_ONECALL": "0"}), torch.no_grad():
            got = e16.encode_tokens_packed(ti, tl, 4096)  # ... and kernel by kernel: the same launches from the Python side
    want = {"lm_gemm_f16": 4 * cfg.layers, "lm_attn_varlen_f16": cfg.layers, "lm_add_layernorm_f16": 2 * cfg.layers, "lm_embed_layernorm_f16": 1}
    counts = {k: used.count(k) for k in want}
    assert counts == want and "lm_bert_forward_packed" not in used, (counts, sorted(set(used)))
    assert torch.equal(one, got), float((one - got).abs().max())
    err = float((got.float() - ref_e).abs().max())
    print(f"hidden 768 packed forward on the general kernels: max|diff| vs fp32 torch = {err:.2e}", flush=True)
    assert err < 6e-3, err


CASES["hidden768"] = case_hidden768


def case_encoder_python_wiring():
    """leann_amd/encoder.py's kernel paths -- the unfused A/B form (LEANN_MI355X_TAIL=0), the default set, the one-call forward -- run on CPU tensors through the emulated library:
    the wrappers' argument wiring (weight packing caches, bias / LayerNorm parameters, cu_seqlens, call order) is what
    the A/B switch sets exercise on the GPU.  Only test code pretends the tensors are device tensors
    (Tensor.is_cuda / current_stream are patched HERE); the product has no such switch."""
    import os
    from unittest import mock

    import torch

    from leann_amd.encoder import BertEncoder, EncoderConfig

    torch.manual_seed(0)
    cfg = EncoderConfig(vocab_size=500, hidden=384, layers=2, heads=12, ffn=128, max_pos=64, max_seq_length=48)
    enc = BertEncoder.random_init(cfg, 3).eval()
    rng = np.random.default_rng(9)
    n, t = 9, 48
    lens = rng.integers(1, t + 1, n).astype(np.int32)
    lens[0], lens[1] = t, 1
    ids = np.zeros((n, t), np.int32)
    for i in range(n):
        ids[i, : lens[i]] = rng.integers(1, cfg.vocab_size, lens[i])
    ti, tl = torch.from_numpy(ids), torch.from_numpy(lens)
    with torch.no_grad():
        ref = enc(ti, tl).float()  # fp32 padded reference path (plain torch)
    enc16 = BertEncoder.random_init(cfg, 3).eval().half()

    class _Stream:
        cuda_stream = 0

    # the unfused form of a layer (a feed-forward width outside the fused tail's envelope; LEANN_MI355X_TAIL=0 selects it at any width)
    switches = {"LEANN_MI355X_LN": "2", "LEANN_MI355X_POOL": "1", "LEANN_MI355X_EMBED": "1", "LEANN_MI355X_TAIL": "0", "LEANN_MI355X_PACK": "1",
                "LEANN_MI355X_ONECALL": "0", "LEANN_MI355X_SMALL_TOKENS": "0"}  # (0: the hidden-384 kernels also for this small forward)
    from leann_amd import _lib

    used = []
    real_check = _lib.check

    def recording_check(rc, what=""):
        used.append(what)
        return real_check(rc, what)

    with mock.patch.object(torch.Tensor, "is_cuda", new=property(lambda self: True)), \
            mock.patch("torch.cuda.current_stream", new=lambda *a, **k: _Stream()), mock.patch.dict(os.environ, switches), \
            mock.patch.object(_lib, "check", new=recording_check):
        with torch.no_grad():
            got = enc16.encode_tokens_packed(ti, tl, 4096)
    expected = {"lm_pack_tokens": 1, "lm_embed_layernorm_f16": 1, "lm_qkv_attn_h384_f16": 0, "lm_qkv_h384_f16": cfg.layers, "lm_gemm_ws_h384_f16": cfg.layers,
                "lm_attn_varlen_hd32_f16": cfg.layers, "lm_gemm_f16": 2 * cfg.layers, "lm_add_layernorm_f16": 2 * cfg.layers, "lm_meanpool_varlen_f16": 1}
    counts = {k: used.count(k) for k in expected}
    assert counts == expected, (counts, sorted(set(used)))  # no library GEMM, no torch op left
    # mixed configuration: packing kernel + torch pooling (needs the lazily built token -> sequence map) + torch embedding
    mixed = {k: v for k, v in switches.items() if k not in ("LEANN_MI355X_POOL", "LEANN_MI355X_EMBED")}
    with mock.patch.object(torch.Tensor, "is_cuda", new=property(lambda self: True)), \
            mock.patch("torch.cuda.current_stream", new=lambda *a, **k: _Stream()), mock.patch.dict(os.environ, mixed):
        for k in ("LEANN_MI355X_POOL", "LEANN_MI355X_EMBED"):
            os.environ.pop(k, None)
        with torch.no_grad():
            got2 = enc16.encode_tokens_packed(ti, tl, 4096)
    assert float((got2.float() - ref).abs().max()) < 6e-3
    # the DEFAULT kernel set of a batch of short sequences (weight-streaming QKV projection, attention, fused layer tail), then the QKV projection fused
    # into attention (the library's choice for long sequences, forced here), then the weight-stationary projection
    for ffn, extra, want in ((384, {}, {"lm_qkv_attn_h384_f16": 0, "lm_qkv_h384_f16": 2, "lm_attn_varlen_hd32_f16": 2, "lm_layer_tail_h384_f16": 2}),
                             (384, {"LEANN_MI355X_FUSED_QKV_ATTN": "1"}, {"lm_qkv_attn_h384_f16": 2, "lm_qkv_h384_f16": 0, "lm_attn_varlen_hd32_f16": 0, "lm_layer_tail_h384_f16": 2}),
                             (384, {"LEANN_MI355X_QKV": "0"}, {"lm_gemm_ws_h384_f16": 2, "lm_attn_varlen_hd32_f16": 2, "lm_layer_tail_h384_f16": 2})):
        cfg3 = EncoderConfig(vocab_size=500, hidden=384, layers=2, heads=12, ffn=ffn, max_pos=64, max_seq_length=48)
        e32 = BertEncoder.random_init(cfg3, 5).eval()
        with torch.no_grad():
            ref3 = e32(ti, tl).float()
        e16 = BertEncoder.random_init(cfg3, 5).eval().half()
        used.clear()
        env3 = {k: v for k, v in os.environ.items() if not k.startswith("LEANN_MI355X_")}
        env3["LEANN_MI355X_ONECALL"] = "0"  # the per-kernel launch path (the one-call default follows below)
        env3["LEANN_MI355X_SMALL_TOKENS"] = "0"  # ... of the LARGE-forward form (a forward this small takes the general kernels by default)
        env3.update(extra)
        with mock.patch.object(torch.Tensor, "is_cuda", new=property(lambda self: True)), \
                mock.patch("torch.cuda.current_stream", new=lambda *a, **k: _Stream()), mock.patch.dict(os.environ, env3, clear=True), \
                mock.patch.object(_lib, "check", new=recording_check):
            with torch.no_grad():
                got3 = e16.encode_tokens_packed(ti, tl, 4096)
        counts3 = {k: used.count(k) for k in want}
        assert counts3 == want and "lm_add_layernorm_f16" not in used, (ffn, extra, counts3, sorted(set(used)))
        err3 = float((got3.float() - ref3).abs().max())
        assert err3 < 6e-3, (ffn, err3)
    # the default launch path: the whole forward as ONE library call (csrc/lm_encoder_forward.cpp) -- same kernels, same result as the default path
    cfg1 = EncoderConfig(vocab_size=500, hidden=384, layers=2, heads=12, ffn=384, max_pos=64, max_seq_length=48)
    e1 = BertEncoder.random_init(cfg1, 5).eval().half()
    outs = {}
    for onecall, small in (("0", "0"), ("1", "0"), ("0", None), ("1", None), ("0", "fused"), ("1", "fused"), ("0", "hybrid"), ("1", "hybrid")):  # large-forward form, small-forward form (general kernels), large form with the fused first half, large form with the QKV projection from the general GEMM
        used.clear()
        env1 = {k: v for k, v in os.environ.items() if not k.startswith("LEANN_MI355X_")}
        env1["LEANN_MI355X_ONECALL"] = onecall
        if small is not None:
            env1["LEANN_MI355X_SMALL_TOKENS"] = "0" if small != "hybrid" else "1"  # hybrid: not a small forward, but below LM_BERT_QKV_GEMM_TOKENS
        if small == "fused":
            env1["LEANN_MI355X_FUSED_QKV_ATTN"] = "1"
        with mock.patch.object(torch.Tensor, "is_cuda", new=property(lambda self: True)), \
                mock.patch("torch.cuda.current_stream", new=lambda *a, **k: _Stream()), mock.patch.dict(os.environ, env1, clear=True), \
                mock.patch.object(_lib, "check", new=recording_check):
            with torch.no_grad():
                outs[(onecall, small)] = e1.encode_tokens_packed(ti, tl, 4096)
        if onecall == "1":
            assert used.count("lm_bert_h384_forward_packed") == 1 and "lm_gemm_ws_h384_f16" not in used, sorted(set(used))
        elif small is None:
            assert used.count("lm_gemm_f16") == 4 * cfg1.layers and "lm_layer_tail_h384_f16" not in used, sorted(set(used))
        elif small == "hybrid":  # (round 6) a forward that fills a fraction of the chip: QKV from the general GEMM, attention, the fused tail
            assert used.count("lm_layer_tail_h384_f16") == cfg1.layers and used.count("lm_gemm_f16") == cfg1.layers and used.count("lm_attn_varlen_hd32_f16") == cfg1.layers
            assert "lm_qkv_h384_f16" not in used and "lm_qkv_attn_h384_f16" not in used, sorted(set(used))
        else:
            assert used.count("lm_layer_tail_h384_f16") == cfg1.layers and "lm_gemm_f16" not in used, sorted(set(used))
            assert used.count("lm_qkv_attn_h384_f16") == (cfg1.layers if small == "fused" else 0), sorted(set(used))
    print("one-call vs per-kernel: large form max|diff|", float((outs[("0", "0")] - outs[("1", "0")]).abs().max()), "small form", float((outs[("0", None)] - outs[("1", None)]).abs().max()), flush=True)
    assert torch.equal(outs[("0", "0")], outs[("1", "0")]) and torch.equal(outs[("0", None)], outs[("1", None)]) and torch.equal(outs[("0", "fused")], outs[("1", "fused")])
    assert float((outs[("1", "fused")].float() - outs[("1", "0")].float()).abs().max()) < 5e-3
    assert torch.equal(outs[("0", "hybrid")], outs[("1", "hybrid")]) and float((outs[("1", "hybrid")].float() - outs[("1", "0")].float()).abs().max()) < 5e-3
    with torch.no_grad():
        ref1 = BertEncoder.random_init(cfg1, 5).eval()(ti, tl).float()
    assert float((outs[("1", None)].float() - ref1).abs().max()) < 6e-3 and float((outs[("1", "0")].float() - ref1).abs().max()) < 6e-3
    err = float((got.float() - ref).abs().max())
    print(f"encoder.py packed forward, unfused form, through the emulated library: max|diff| vs fp32 torch = {err:.2e}", flush=True)
    assert err < 6e-3, err


CASES["encoder_python_wiring"] = case_encoder_python_wiring


def case_native_recompute(hidden=384, heads=12, pooling="mean", searches=True):
    """The built-in recompute provider (csrc/lm_recompute.hip) against the Python one (leann_amd/recompute.py: RecomputeProvider.__call__)
    on the same encoder, token store and graph: (1) embeddings of a ragged id list, one forward and -- with a 192-token budget -- several
    sub-batched forwards, bit-identical between the two forms (same token batches into the same kernels); (2) a recompute-mode search
    whose provider is the library's (lm_index_set_recompute: lengths computed by the search loop's own round, ONE synchronisation per
    round) equals the search over the Python provider AND the oracle over the table of those embeddings: labels, distances, evaluation
    counts; one query (no memo) and several (default memo).  Model shapes: hidden 384 / mean pooling (all-MiniLM: fused kernels,
    lm_recompute_create), hidden 384 / CLS (bge-small), and a general width with head_dim 64 and CLS pooling (the bge-base form:
    lm_bert_forward_packed, lm_recompute_create_general; head_dim 32 / mean pooling at a general width passes as well: run it by hand
    with case_native_recompute(256, 8, "mean")).  Device tensors are pretended as in case_encoder_python_wiring."""
    import os
    from unittest import mock

    import torch

    from leann_amd.encoder import BertEncoder, EncoderConfig
    from leann_amd.hnsw_builder import build_hnsw
    from leann_amd.index import Mi355xIndex
    from leann_amd.recompute import RecomputeProvider
    from leann_amd.token_store import TokenStore
    from oracle import oracle as orc

    class _Stream:
        cuda_stream = 0

    torch.manual_seed(1)
    cfg = EncoderConfig(vocab_size=300, hidden=hidden, layers=1, heads=heads, ffn=384 if hidden == 384 else 128, max_pos=32, max_seq_length=24, pooling=pooling)
    enc = BertEncoder.random_init(cfg, 7).eval().half()
    rng = np.random.default_rng(5)
    n = 72
    lens = rng.integers(1, 20, n)
    lens[3], lens[10] = 24, 1
    seqs = [rng.integers(1, cfg.vocab_size, int(l)).tolist() for l in lens]
    store = TokenStore.from_lists(seqs)
    dev = torch.device("cpu")
    tag = f"native provider h={hidden} {pooling}"
    env = {k: v for k, v in os.environ.items() if not k.startswith("LEANN_MI355X_")}
    from leann_amd import _lib

    used = []
    real_check = _lib.check

    def recording_check(rc, what=""):
        used.append(what)
        return real_check(rc, what)

    with mock.patch.object(torch.Tensor, "is_cuda", new=property(lambda self: True)), \
            mock.patch("torch.cuda.current_stream", new=lambda *a, **k: _Stream()), mock.patch.dict(os.environ, env, clear=True), \
            mock.patch.object(_lib, "check", new=recording_check):
        py = RecomputeProvider(enc, store, hidden, dev)
        with mock.patch.dict(os.environ, {"LEANN_MI355X_NATIVE_PROVIDER": "0"}):
            assert py.native() is None
            ids_all = torch.arange(n, dtype=torch.int32)
            X = py.embed_ids(ids_all).clone()  # Python form: gather + encode_tokens (one-call forward)
            assert ("lm_bert_h384_forward_packed" if hidden == 384 else "lm_bert_forward_packed") in used, sorted(set(used))
            with mock.patch.dict(os.environ, {"LEANN_MI355X_ONECALL": "0"}):  # ... and the per-kernel launch path: same kernels, same bits
                Xk = py.embed_ids(ids_all).clone()
            assert torch.equal(X, Xk), float((X - Xk).abs().max())
        nat = RecomputeProvider(enc, store, hidden, dev)
        assert nat.native() is not None
        assert ("lm_recompute_create" if hidden == 384 else "lm_recompute_create_general") in used
        Xn = nat.embed_ids(ids_all)
        print(f"{tag}: vs Python provider, all chunks in one forward: max|diff|", float((X - Xn).abs().max()), flush=True)
        assert torch.equal(X, Xn)
        if pooling == "cls":  # the pooling kernel against torch: first token's row, normalised
            with torch.no_grad():
                ref = BertEncoder.random_init(cfg, 7).eval()(torch.tensor([s + [0] * (24 - len(s)) for s in seqs], dtype=torch.int32),
                                                            torch.tensor([len(s) for s in seqs], dtype=torch.int32)).float()
            assert float((X - ref).abs().max()) < 6e-3, float((X - ref).abs().max())
        pick = torch.from_numpy(np.sort(rng.choice(n, 23, replace=False)).astype(np.int32))
        small_py = RecomputeProvider(enc, store, hidden, dev, batch_size=1)   # 192 tokens per forward: several forwards
        small_nat = RecomputeProvider(enc, store, hidden, dev, batch_size=1)
        with mock.patch.dict(os.environ, {"LEANN_MI355X_NATIVE_PROVIDER": "0"}):
            a = small_py.embed_ids(pick).clone()
        b = small_nat.embed_ids(pick)
        st = small_nat.native_stats()
        assert st["forwards"] > 1 and st["chunks"] == 23, st
        assert torch.equal(a, b) and torch.equal(a, X[pick.long()]), float((a - b).abs().max())
        # ---- search: library provider vs Python provider vs oracle over the table of the same embeddings
        x = X.numpy().astype(np.float32)
        g = build_hnsw(x, "mips", M=4, ef_construction=24)
        og = orc.OracleGraph(g.node_offsets, g.level_ptr, g.neighbors, g.levels, g.entry_point, g.max_level, g.metric_type, hidden)
        q = (x[[5, 40, 61]] + 0.05 * rng.standard_normal((3, hidden))).astype(np.float32)
        for nq in ((1, 3) if searches else (3,)):
            exp = orc.search(og, q[:nq], 4, ef=8, beam=2, table=x)
            idx = Mi355xIndex.from_csr(g)
            idx.set_provider(nat)
            assert idx.native_provider
            s0 = nat.native_stats()
            got = idx.search(q[:nq], 4, idx.make_params(ef=8, beam=2, recompute=True))
            st_n, s1 = idx.stats(), nat.native_stats()
            _check(f"{tag} nq={nq} vs oracle", got, exp[:2], st_n, exp[2])
            # one synchronisation per round: the provider itself never synchronised during the search (buffers were already grown)
            assert s1["host_syncs"] == s0["host_syncs"], (s0, s1)
            assert s1["chunks"] - s0["chunks"] == int(st_n["nunique"])
            if nq == 1:  # option "single_query_direct": the new-list goes to the provider as it is (no k_uniq_*): same answer, same counts, still one sync per round
                idx.set_option("single_query_direct", 1)
                s2 = nat.native_stats()
                got_d = idx.search(q[:1], 4, idx.make_params(ef=8, beam=2, recompute=True))
                st_d, s3 = idx.stats(), nat.native_stats()
                _check(f"{tag} nq=1 single_query_direct vs oracle", got_d, exp[:2], st_d, exp[2])
                assert s3["host_syncs"] == s2["host_syncs"] and s3["chunks"] - s2["chunks"] == int(st_d["nunique"]) == int(st_n["nunique"])
                assert int(st_d["nrounds"]) == int(st_n["nrounds"])
            idx.close()
            if not searches or nq == 1:
                continue
            idx2 = Mi355xIndex.from_csr(g)
            with mock.patch.dict(os.environ, {"LEANN_MI355X_NATIVE_PROVIDER": "0"}):
                idx2.set_provider(py)
                assert not idx2.native_provider
                got2 = idx2.search(q[:nq], 4, idx2.make_params(ef=8, beam=2, recompute=True))
            _check(f"python provider nq={nq} vs oracle", got2, exp[:2], idx2.stats(), exp[2])
            assert int(idx2.stats()["nunique"]) == int(st_n["nunique"])
            idx2.close()
        if searches:
            # as a plain lm_provider_fn (lm_index_set_provider with the exported function): same result, the provider synchronises itself
            lib = _lib.load()
            idx3 = Mi355xIndex.from_csr(g)
            fn = C.cast(lib.lm_recompute_provider, _lib.PROVIDER_FN)
            _lib.check(lib.lm_index_set_provider(idx3._h, fn, nat.native()), "lm_index_set_provider")
            got3 = idx3.search(q, 4, idx3.make_params(ef=8, beam=2, recompute=True))
            exp = orc.search(og, q, 4, ef=8, beam=2, table=x)
            _check(f"{tag} as a plain lm_provider_fn", got3, exp[:2], idx3.stats(), exp[2])
            idx3.close()
            # a provider of another width is refused by the index
            other = Mi355xIndex.from_csr(build_hnsw(x[:, :64].copy(), "mips", M=4, ef_construction=24))
            try:
                _lib.check(lib.lm_index_set_recompute(other._h, nat.native()), "lm_index_set_recompute")
                raise AssertionError("a 64-d index accepted a provider of another width")
            except ValueError:
                pass
            other.close()
            # closing the provider while an index still holds its handle detaches it there first: the next recompute search fails loudly
            idx4 = Mi355xIndex.from_csr(g)
            idx4.set_provider(nat)
            assert idx4.native_provider
            nat.close()
            assert not idx4.native_provider
            try:
                idx4.search(q[:1], 4, idx4.make_params(ef=8, beam=2, recompute=True))
                raise AssertionError("search over a closed provider did not fail")
            except RuntimeError as ex:
                assert "provider" in str(ex), ex
            idx4.close()
        for p_ in (nat, small_nat, py, small_py):
            p_.close()
    store.close()


CASES["native_recompute"] = case_native_recompute
CASES["native_recompute_h384_cls"] = lambda: case_native_recompute(384, 12, "cls", searches=False)
CASES["native_recompute_general_hd64_cls"] = lambda: case_native_recompute(128, 2, "cls", searches=False)


def case_native_recompute_long_id_list():
    """An id list longer than one pass of the built-in provider's length scan (k_rc_lengths_scan walks 1024 ids per pass: carry between
    passes, the next pass's lengths requested early) and, with a 768-token budget, cut into several forwards: 1100 ids (with repeats) over
    very short chunks; embeddings bit-identical to the Python provider's, token count exact."""
    import os
    from unittest import mock

    import torch

    from leann_amd.encoder import BertEncoder, EncoderConfig
    from leann_amd.recompute import RecomputeProvider
    from leann_amd.token_store import TokenStore

    class _Stream:
        cuda_stream = 0

    cfg = EncoderConfig(vocab_size=200, hidden=128, layers=1, heads=2, ffn=128, max_pos=16, max_seq_length=8, pooling="mean")
    enc = BertEncoder.random_init(cfg, 4).eval().half()
    rng = np.random.default_rng(8)
    lens = rng.integers(1, 3, 40)
    lens[7] = 0  # an empty chunk: contributes no tokens, pools to the zero vector in both forms
    seqs = [rng.integers(1, cfg.vocab_size, int(l)).tolist() for l in lens]
    store = TokenStore.from_lists(seqs)
    ids = torch.from_numpy(rng.integers(0, 40, 1100).astype(np.int32))
    env = {k: v for k, v in os.environ.items() if not k.startswith("LEANN_MI355X_")}
    with mock.patch.object(torch.Tensor, "is_cuda", new=property(lambda self: True)), \
            mock.patch("torch.cuda.current_stream", new=lambda *a, **k: _Stream()), mock.patch.dict(os.environ, env, clear=True):
        for bs in (5461, 4):
            nat = RecomputeProvider(enc, store, 128, torch.device("cpu"), batch_size=bs)
            py = RecomputeProvider(enc, store, 128, torch.device("cpu"), batch_size=bs)
            a = nat.embed_ids(ids)
            st = nat.native_stats()
            with mock.patch.dict(os.environ, {"LEANN_MI355X_NATIVE_PROVIDER": "0"}):
                b = py.embed_ids(ids)
            assert st["chunks"] == 1100 and st["tokens"] == int(lens[ids.numpy()].sum()), st
            assert (st["forwards"] == 1) == (bs == 5461), st
            assert torch.equal(a, b), float((a - b).abs().max())
            assert not torch.isnan(a).any() and float(a[ids.numpy() == 7].abs().max()) == 0.0
            nat.close()
    store.close()
    print("native provider, 1100-id list (2 scan passes), one forward and 768-token forwards: ok", flush=True)


CASES["native_recompute_long_id_list"] = case_native_recompute_long_id_list


def case_embedding_service():
    """SURVEY 8 rows a4 / a5 / f1 on the CPU box: the wire handlers of the ZMQ-compatible server (leann_amd/embedding_server.py: the
    reference's hnsw_embedding_server.py:119-284 msgpack protocol and diskann_embedding_server.py:258-334 protobuf protocol) over the
    emulated library -- embeddings by id through the built-in recompute provider (general-width model), distances through lm_dist_gather,
    unknown ids, malformed requests."""
    import os
    from unittest import mock

    import msgpack
    import torch

    from leann_amd import embedding_server as es
    from leann_amd.encoder import BertEncoder, EncoderConfig
    from leann_amd.token_store import TokenStore

    class _Stream:
        cuda_stream = 0

    cfg = EncoderConfig(vocab_size=300, hidden=128, layers=1, heads=2, ffn=128, max_pos=32, max_seq_length=16, pooling="cls")
    enc = BertEncoder.random_init(cfg, 6).eval().half()
    rng = np.random.default_rng(2)
    seqs = [rng.integers(1, cfg.vocab_size, int(l)).tolist() for l in rng.integers(1, 12, 30)]
    store = TokenStore.from_lists(seqs)
    with torch.no_grad():
        ref = BertEncoder.random_init(cfg, 6).eval()(torch.tensor([s_ + [0] * (16 - len(s_)) for s_ in seqs], dtype=torch.int32),
                                                    torch.tensor([len(s_) for s_ in seqs], dtype=torch.int32)).float().numpy()
    env = {k: v for k, v in os.environ.items() if not k.startswith("LEANN_MI355X_")}
    with mock.patch.object(torch.Tensor, "is_cuda", new=property(lambda self: True)), \
            mock.patch("torch.cuda.current_stream", new=lambda *a, **k: _Stream()), mock.patch.dict(os.environ, env, clear=True):
        svc = es.Mi355xEmbeddingService("test-model", enc, store, None, "mips", device=torch.device("cpu"))
        assert msgpack.unpackb(svc.handle_msgpack(msgpack.packb(["__QUERY_MODEL__"]))) == ["test-model"]
        dims, flat = msgpack.unpackb(svc.handle_msgpack(msgpack.packb([[3, 29, 1000]])))  # an unknown id -> zero row
        e = np.asarray(flat, np.float32).reshape(dims)
        assert dims == [3, 128] and np.allclose(e[:2], ref[[3, 29]], atol=4e-3) and np.all(e[2] == 0)
        assert svc.provider.native_stats()["chunks"] == 2  # the embeddings came from the library-side provider
        q = ref[7].tolist()
        (d,) = msgpack.unpackb(svc.handle_msgpack(msgpack.packb([[7, 8, 12345], q])))  # distances: -e.q for mips, 1e9 for the unknown id
        assert np.allclose(d[:2], -(ref[[7, 8]] @ ref[7]), atol=8e-3) and abs(d[2] - 1e9) < 1.0
        svc.distance_metric = "l2"
        (d2,) = msgpack.unpackb(svc.handle_msgpack(msgpack.packb([[[7, 8]], q])))
        assert np.allclose(d2, ((ref[[7, 8]] - ref[7]) ** 2).sum(1), atol=8e-3)
        assert msgpack.unpackb(svc.handle_msgpack(b"\xc1")) == [[0, 128], []]  # malformed request -> shape-correct fallback
        data, dims, missing = es.decode_node_embedding_response(svc.handle_diskann(es.encode_node_embedding_request([1, 2, 3])))
        assert dims == [3, 128] and missing == [] and np.allclose(np.frombuffer(data, np.float32).reshape(3, 128), ref[1:4], atol=4e-3)
        assert svc.handle_diskann(b"") == b""
        svc.provider.close()
    store.close()
    print("embedding service handlers over the emulated library (msgpack + protobuf protocols): ok", flush=True)


CASES["embedding_service"] = case_embedding_service


def case_gpu_graph_builder():
    """Row f4 on the CPU box: leann_amd/gpu_graph_build.build_graph_gpu with its DEFAULT candidate search -- the product's stored-embedding
    search kernel (hip_search_fn: lm_index_search_device, persistent k_search_table) -- over the emulated library, then the paper's
    Algorithm 3 pruning: the graph is valid, and a search over it finds the exact neighbours."""
    from unittest import mock

    import torch

    from leann_amd.gpu_graph_build import build_graph_gpu, prune_preserving_hubs
    from oracle import oracle as orc
    from tests.util import clustered, oracle_graph, queries_near, recall_at_k

    class _Stream:
        cuda_stream = 0

    x = clustered(320, 32, 4, n_centers=4, sigma=0.6)
    with mock.patch.object(torch.Tensor, "is_cuda", new=property(lambda self: True)), \
            mock.patch("torch.cuda.current_stream", new=lambda *a, **k: _Stream()), mock.patch("torch.cuda.synchronize", new=lambda *a, **k: None):
        g = build_graph_gpu(torch.from_numpy(x), "mips", M=8, ef_construction=40, seed_nodes=128)
    g.validate()
    assert g.ntotal == 320 and g.level0_degrees().max() <= 16 and g.level0_degrees().min() >= 1
    q = queries_near(x, 30, 5)
    gt, _ = orc.bruteforce_topk(x, q, 10, 0)
    ids, _, _ = orc.search(oracle_graph(g, 32), q, 10, ef=48, table=x)
    r = recall_at_k(ids, gt)
    g2 = prune_preserving_hubs(g, torch.from_numpy(x), M=8, m_low=4, hub_fraction=0.05)
    g2.validate()
    ids2, _, _ = orc.search(oracle_graph(g2, 32), q, 10, ef=48, table=x)
    r2 = recall_at_k(ids2, gt)
    print(f"graph builder over the emulated search kernel: recall@10 {r:.3f}; after hub-preserving pruning ({g2.neighbors.shape[0]} of {g.neighbors.shape[0]} links) {r2:.3f}", flush=True)
    assert r > 0.92 and r2 > 0.9 and g2.neighbors.shape[0] < g.neighbors.shape[0]


CASES["gpu_graph_builder"] = case_gpu_graph_builder


if __name__ == "__main__":
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    _load(sys.argv[1])
    from oracle import oracle as _orc

    import torch

    torch.set_num_threads(1)
    _orc.set_num_threads(1)  # libgomp's own synchronisation is invisible to ThreadSanitizer (false positives in the ORACLE)
    import time

    for name in (sys.argv[2:] or list(CASES)):
        t0 = time.time()
        CASES[name]()
        print(f"[case {name}: {time.time() - t0:.1f} s]", flush=True)
    print("ALL CASES OK")
