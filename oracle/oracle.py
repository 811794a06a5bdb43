"""ctypes front-end of the CPU oracle (oracle/lm_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, bench.py's ``cpu_baseline`` leg and
``__graft_entry__.smoke()``.  Nothing under ``leann_amd/`` may import this module.
Parity status: "parity unpinned" at the faiss boundary (see lm_oracle.c header).
"""

from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path
from typing import Callable, Optional

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "_build" / "liblm_oracle.so"

METRIC_IP = 0
METRIC_L2 = 1


def build(force: bool = False) -> Path:
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force or not _LIB_PATH.exists():
        subprocess.run(["make", "-C", str(_HERE)] + (["-B"] if force else []), check=True, capture_output=True)
    return _LIB_PATH


class _Graph(C.Structure):
    _fields_ = [
        ("N", C.c_int64),
        ("D", C.c_int32),
        ("Dp", C.c_int32),
        ("max_level", C.c_int32),
        ("entry_point", C.c_int32),
        ("metric", C.c_int32),
        ("node_offsets", C.c_void_p),
        ("level_ptr", C.c_void_p),
        ("neighbors", C.c_void_p),
        ("levels", C.c_void_p),
    ]


class _Params(C.Structure):
    _fields_ = [("ef", C.c_int32), ("beam", C.c_int32), ("k", C.c_int32), ("check_relative_distance", C.c_int32),
                ("prune_ratio", C.c_float), ("prune_strategy", C.c_int32), ("batch_size", C.c_int32)]


class _Stats(C.Structure):
    _fields_ = [("ndis", C.c_int64), ("nunique", C.c_int64), ("nrounds", C.c_int64), ("nexpand", C.c_int64), ("nadc", C.c_int64)]


PRUNE_STRATEGY = {"global": 0, "local": 1, "proportional": 2}


class _Pq(C.Structure):
    _fields_ = [("m", C.c_int32), ("dsub", C.c_int32), ("codebooks", C.c_void_p), ("codes", C.c_void_p), ("chunk_off", C.c_void_p)]


class _PqParams(C.Structure):
    _fields_ = [("L", C.c_int32), ("W", C.c_int32), ("k", C.c_int32), ("use_deferred_fetch", C.c_int32),
                ("skip_search_reorder", C.c_int32)]


class _PqStats(C.Structure):
    _fields_ = [("n_adc", C.c_int64), ("n_rerank_unique", C.c_int64), ("n_rounds", C.c_int64), ("n_expand", C.c_int64)]


class _DiskannStats(C.Structure):
    _fields_ = [("n_cmps", C.c_int64), ("n_hops", C.c_int64), ("n_expanded", C.c_int64), ("max_hops", C.c_int64), ("n_final_differs", C.c_int64)]


class _FaissStats(C.Structure):
    _fields_ = [("ndis", C.c_int64), ("ndis_upper", C.c_int64), ("nstep", C.c_int64)]


_PROVIDER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_float))

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(_LIB_PATH))
        _lib.orc_dist.restype = C.c_float
        _lib.orc_dist.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        _lib.orc_search.restype = C.c_int
        _lib.orc_search.argtypes = [
            C.POINTER(_Graph), C.c_void_p, _PROVIDER, C.c_void_p, C.c_void_p, C.c_int32,
            C.POINTER(_Params), C.c_void_p, C.c_void_p, C.POINTER(_Stats),
        ]
        _lib.orc_search_pq.restype = C.c_int
        _lib.orc_search_pq.argtypes = [
            C.POINTER(_Graph), C.c_void_p, _PROVIDER, C.c_void_p, C.c_void_p, C.c_int32,
            C.POINTER(_Params), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(_Stats),
        ]
        _lib.orc_bruteforce_topk.restype = C.c_int
        _lib.orc_bruteforce_topk.argtypes = [
            C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
        ]
        _lib.orc_merge_topk.restype = C.c_int
        _lib.orc_merge_topk.argtypes = [
            C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
        ]
        _lib.orc_num_threads.restype = C.c_int
        _lib.orc_pq_search.restype = C.c_int
        _lib.orc_pq_search.argtypes = [C.POINTER(_Graph), C.POINTER(_Pq), C.c_void_p, _PROVIDER, C.c_void_p, C.c_void_p,
                                       C.c_int32, C.POINTER(_PqParams), C.c_void_p, C.c_void_p, C.POINTER(_PqStats)]
        _lib.orc_pq_lut.restype = None
        _lib.orc_pq_lut.argtypes = [C.POINTER(_Pq), C.c_void_p, C.c_int32, C.c_void_p]
        _lib.orc_pq_adc.restype = C.c_float
        _lib.orc_pq_adc.argtypes = [C.POINTER(_Pq), C.c_void_p, C.c_int64]
        _lib.orcf_search.restype = C.c_int
        _lib.orcf_search.argtypes = [C.POINTER(_Graph), C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_void_p, C.c_void_p, C.POINTER(_FaissStats)]
        if hasattr(_lib, "orcd_search"):  # (a library built before lm_oracle_diskann.c existed still serves every other test)
            _lib.orcd_search.restype = C.c_int
            _lib.orcd_search.argtypes = [C.POINTER(_Graph), C.POINTER(_Pq), C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(_DiskannStats)]
        _lib.orc_set_num_threads.argtypes = [C.c_int]
        _lib.orc_set_num_threads.restype = None
        _lib.orc_set_num_threads(usable_cores())
    return _lib


def pad64(x: np.ndarray) -> np.ndarray:
    """Zero-pad the last dim to a multiple of 64 floats (the canonical row layout)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    d = x.shape[-1]
    dp = (d + 63) // 64 * 64
    if dp == d:
        return x
    out = np.zeros(x.shape[:-1] + (dp,), dtype=np.float32)
    out[..., :d] = x
    return out


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def dist(e: np.ndarray, q: np.ndarray, metric: int) -> float:
    e, q = pad64(e), pad64(q)
    return float(lib().orc_dist(_ptr(e), _ptr(q), e.shape[-1], metric))


class OracleGraph:
    """Holds the compact-CSR arrays (convert_to_csr.py:182-237 layout) for the oracle."""

    def __init__(self, node_offsets, level_ptr, neighbors, levels, entry_point, max_level, metric, D):
        self.node_offsets = np.ascontiguousarray(node_offsets, dtype=np.uint64)
        self.level_ptr = np.ascontiguousarray(level_ptr, dtype=np.uint64)
        self.neighbors = np.ascontiguousarray(neighbors, dtype=np.int32)
        self.levels = np.ascontiguousarray(levels, dtype=np.int32)
        self.N = int(self.levels.shape[0])
        self.D = int(D)
        self.Dp = (self.D + 63) // 64 * 64
        self.entry_point, self.max_level, self.metric = int(entry_point), int(max_level), int(metric)
        if self.neighbors.size == 0:  # keep a valid pointer
            self.neighbors = np.zeros(1, dtype=np.int32)

    def cstruct(self) -> _Graph:
        return _Graph(self.N, self.D, self.Dp, self.max_level, self.entry_point, self.metric,
                      _ptr(self.node_offsets), _ptr(self.level_ptr), _ptr(self.neighbors), _ptr(self.levels))


def search(graph: OracleGraph, queries: np.ndarray, k: int, ef: int = 64, beam: int = 1,
           check_relative_distance: bool = True, table: Optional[np.ndarray] = None,
           provider: Optional[Callable[[np.ndarray], np.ndarray]] = None, prune_ratio: float = 0.0,
           pruning_strategy: str = "global", pq=None, memo: bool = False, batch_size: int = 0):
    """Run the oracle search.  Exactly one of ``table`` (N x D, stored embeddings) or
    ``provider`` (callable: sorted unique int32 ids -> (n, D) float32) must be given.
    ``memo`` restates the product's per-call recompute memo (lm_search_params.recompute_memo, the library default for a call of
    more than one query): a node's embedding is requested from ``provider`` at most once per call and kept until the call returns;
    ids, distances and ndis are unchanged by construction, ``stats["nunique"]`` becomes the number of rows actually requested.
    ``batch_size`` > 0: dynamic batching (lm_oracle.c header, "batching": a query keeps popping while its round's new-list is shorter).
    Returns (ids int64 (B,k), dist float32 (B,k), stats dict)."""
    assert (table is None) != (provider is None)
    q = pad64(np.atleast_2d(queries))
    B = q.shape[0]
    assert q.shape[1] == graph.Dp, (q.shape, graph.Dp)
    ids = np.empty((B, k), dtype=np.int64)
    dd = np.empty((B, k), dtype=np.float32)
    st = _Stats()
    prm = _Params(ef, beam, k, 1 if check_relative_distance else 0, float(prune_ratio), PRUNE_STRATEGY[pruning_strategy], int(batch_size))
    g = graph.cstruct()
    pqs = None
    if pq is not None:  # (codebooks [m,256,dsub], codes [N,m]) for the two-level search
        cbk = np.ascontiguousarray(pq[0], dtype=np.float32)
        cds = np.ascontiguousarray(pq[1], dtype=np.uint8)
        pqs = _Pq(cbk.shape[0], cbk.shape[2], _ptr(cbk), _ptr(cds), None)
    tab = None
    err: list = []
    if table is not None:
        tab = pad64(table)
        assert tab.shape == (graph.N, graph.Dp)
        cb = _PROVIDER()
    else:
        Dp = graph.Dp
        slot = np.full(graph.N, -1, dtype=np.int64) if memo else None  # memo: row of every node fetched so far in this call
        rows: list = []
        memo_rows = [0]

        def _cb(_user, ids_p, n, out_p):
            try:
                idv = np.ctypeslib.as_array(ids_p, shape=(n,)).copy()
                out = np.ctypeslib.as_array(out_p, shape=(n, Dp))
                if slot is None:
                    e = pad64(np.asarray(provider(idv), dtype=np.float32))
                    assert e.shape == (n, Dp), e.shape
                    out[:] = e
                    return 0
                fresh = idv[slot[idv] < 0]
                if fresh.size:
                    e = pad64(np.asarray(provider(fresh), dtype=np.float32))
                    assert e.shape == (fresh.size, Dp), e.shape
                    slot[fresh] = memo_rows[0] + np.arange(fresh.size)
                    memo_rows[0] += fresh.size
                    rows.append(e)
                    if len(rows) > 64:  # keep the lookup below a handful of blocks
                        rows[:] = [np.concatenate(rows)]
                allrows = rows[0] if len(rows) == 1 else np.concatenate(rows)
                rows[:] = [allrows]
                out[:] = allrows[slot[idv]]
                return 0
            except Exception as ex:  # noqa: BLE001 - surfaced after the C call returns
                err.append(ex)
                return 1

        cb = _PROVIDER(_cb)
    rc = lib().orc_search_pq(C.byref(g), _ptr(tab), cb, None, _ptr(q), B, C.byref(prm), C.byref(pqs) if pqs is not None else None,
                             _ptr(ids), _ptr(dd), C.byref(st))
    if err:
        raise err[0]
    if rc:
        raise RuntimeError(f"orc_search failed rc={rc}")
    stats = {f: int(getattr(st, f)) for f, _ in _Stats._fields_}
    if provider is not None and memo:
        stats["nunique"] = int(memo_rows[0])
    return ids, dd, stats


def _make_provider(provider, Dp, err):
    def _cb(_user, ids_p, n, out_p):
        try:
            idv = np.ctypeslib.as_array(ids_p, shape=(n,)).copy()
            e = pad64(np.asarray(provider(idv), dtype=np.float32))
            assert e.shape == (n, Dp), e.shape
            np.ctypeslib.as_array(out_p, shape=(n, Dp))[:] = e
            return 0
        except Exception as ex:  # noqa: BLE001
            err.append(ex)
            return 1

    return _PROVIDER(_cb)


def _pq_struct(codebooks: np.ndarray, codes: np.ndarray, chunk_off=None):
    """(_Pq, keep-alive tuple).  Uniform: codebooks [m, 256, dsub]; chunked (public DiskANN layout): flat codebooks of
    256 * chunk_off[m] floats + chunk_off int32[m + 1]."""
    cb = np.ascontiguousarray(codebooks, dtype=np.float32)
    cd = np.ascontiguousarray(codes, dtype=np.uint8)
    if chunk_off is None:
        m, _, dsub = cb.shape
        return _Pq(m, dsub, _ptr(cb), _ptr(cd), None), (cb, cd), m
    co = np.ascontiguousarray(chunk_off, dtype=np.int32)
    m = co.shape[0] - 1
    assert cb.size == 256 * int(co[-1]) and cd.shape[1] == m and np.all(np.diff(co) >= 0) and co[0] == 0
    return _Pq(m, 0, _ptr(cb), _ptr(cd), _ptr(co)), (cb, cd, co), m


def pq_search(graph: OracleGraph, codebooks: np.ndarray, codes: np.ndarray, queries: np.ndarray, k: int, L: int = 64,
              W: int = 1, table: Optional[np.ndarray] = None, provider=None, use_deferred_fetch: bool = False,
              skip_search_reorder: bool = False, chunk_off=None):
    """DiskANN-style oracle: PQ-ADC traversal (+ exact rerank from `table`, or from `provider` when
    use_deferred_fetch).  Returns (ids, dist, stats)."""
    pq, _keep, m = _pq_struct(codebooks, codes, chunk_off)
    assert np.asarray(codes).shape == (graph.N, m)
    q = pad64(np.atleast_2d(queries))
    B = q.shape[0]
    ids = np.empty((B, k), dtype=np.int64)
    dd = np.empty((B, k), dtype=np.float32)
    prm = _PqParams(L, W, k, 1 if use_deferred_fetch else 0, 1 if skip_search_reorder else 0)
    st = _PqStats()
    g = graph.cstruct()
    tab = pad64(table) if table is not None else None
    err: list = []
    cbf = _make_provider(provider, graph.Dp, err) if provider is not None else _PROVIDER()
    rc = lib().orc_pq_search(C.byref(g), C.byref(pq), _ptr(tab), cbf, None, _ptr(q), B, C.byref(prm), _ptr(ids), _ptr(dd),
                             C.byref(st))
    if err:
        raise err[0]
    if rc:
        raise RuntimeError(f"orc_pq_search failed rc={rc}")
    return ids, dd, {f: int(getattr(st, f)) for f, _ in _PqStats._fields_}


def pq_lut_adc(codebooks: np.ndarray, codes: np.ndarray, query: np.ndarray, metric: int, ids: np.ndarray, chunk_off=None):
    """Canonical LUT + ADC distances of `ids` (for kernel-level tests)."""
    pq, _keep, m = _pq_struct(codebooks, codes, chunk_off)
    qv = np.ascontiguousarray(query, dtype=np.float32)
    lut = np.empty((m, 256), np.float32)
    lib().orc_pq_lut(C.byref(pq), _ptr(qv), metric, _ptr(lut))
    return lut, np.array([lib().orc_pq_adc(C.byref(pq), _ptr(lut), int(v)) for v in ids], np.float32)


def bruteforce_topk(table: np.ndarray, queries: np.ndarray, k: int, metric: int):
    tab, q = pad64(table), pad64(np.atleast_2d(queries))
    B = q.shape[0]
    ids = np.empty((B, k), dtype=np.int64)
    dd = np.empty((B, k), dtype=np.float32)
    lib().orc_bruteforce_topk(_ptr(tab), tab.shape[0], tab.shape[1], metric, _ptr(q), B, k, _ptr(ids), _ptr(dd))
    return ids, dd


def merge_topk(ids: np.ndarray, dist_: np.ndarray, metric: int):
    """ids/dist: (S, B, k) -> (B, k) merged by (internal distance, id)."""
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    dist_ = np.ascontiguousarray(dist_, dtype=np.float32)
    S, B, k = ids.shape
    oi = np.empty((B, k), dtype=np.int64)
    od = np.empty((B, k), dtype=np.float32)
    lib().orc_merge_topk(_ptr(ids), _ptr(dist_), S, B, k, metric, _ptr(oi), _ptr(od))
    return oi, od


def usable_cores() -> int:
    """Cores this process may actually use (affinity mask and cgroup cpu.max aware)."""
    import os

    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, p_ = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p_))))
    except Exception:  # noqa: BLE001
        pass
    return max(1, n)


def set_num_threads(n: int) -> None:
    lib().orc_set_num_threads(int(n))


def num_threads() -> int:
    return int(lib().orc_num_threads())


def faiss_search(graph: OracleGraph, queries: np.ndarray, k: int, ef: int = 64, check_relative_distance: bool = True,
                 table: Optional[np.ndarray] = None):
    """The second oracle (oracle/lm_oracle_faiss.c): a literal heap-based transcription of upstream faiss'
    HNSW::search / search_from_candidates / MinimaxHeap, one query at a time, stored embeddings.
    Returns (ids int64 (B,k), dist float32 (B,k), stats dict: ndis (level 0), ndis_upper, nstep)."""
    q = pad64(np.atleast_2d(queries))
    B = q.shape[0]
    assert q.shape[1] == graph.Dp, (q.shape, graph.Dp)
    tab = pad64(table)
    assert tab.shape == (graph.N, graph.Dp)
    ids = np.empty((B, k), dtype=np.int64)
    dd = np.empty((B, k), dtype=np.float32)
    st = _FaissStats()
    g = graph.cstruct()
    rc = lib().orcf_search(C.byref(g), _ptr(tab), _ptr(q), B, k, ef, 1 if check_relative_distance else 0, _ptr(ids), _ptr(dd),
                           C.byref(st))
    if rc != 0:
        raise RuntimeError(f"orcf_search failed rc={rc}")
    return ids, dd, {"ndis": st.ndis, "ndis_upper": st.ndis_upper, "nstep": st.nstep}


def diskann_search(graph: OracleGraph, codebooks: np.ndarray, codes: np.ndarray, queries: np.ndarray, k: int, L: int = 64, W: int = 1,
                   table: Optional[np.ndarray] = None, rerank_final_list_only: bool = True, chunk_off=None):
    """The second oracle of the DiskANN-style path (oracle/lm_oracle_diskann.c): a literal transcription of upstream DiskANN's
    PQFlashIndex::cached_beam_search / NeighborPriorityQueue, one query at a time, exact distances from ``table``.
    ``rerank_final_list_only`` = True ranks the final candidate list (the product's deferred fetch; must equal ``pq_search``),
    False ranks upstream's full_retset (every expanded node).  Returns (ids int64 (B,k), dist float32 (B,k), stats dict)."""
    pq, _keep, m = _pq_struct(codebooks, codes, chunk_off)
    assert np.asarray(codes).shape == (graph.N, m)
    q = pad64(np.atleast_2d(queries))
    B = q.shape[0]
    tab = pad64(table)
    assert tab.shape == (graph.N, graph.Dp) and q.shape[1] == graph.Dp
    ids = np.empty((B, k), dtype=np.int64)
    dd = np.empty((B, k), dtype=np.float32)
    st = _DiskannStats()
    g = graph.cstruct()
    rc = lib().orcd_search(C.byref(g), C.byref(pq), _ptr(tab), _ptr(q), B, k, L, W, 1 if rerank_final_list_only else 0, _ptr(ids), _ptr(dd),
                           C.byref(st))
    if rc != 0:
        raise RuntimeError(f"orcd_search failed rc={rc}")
    return ids, dd, {f: int(getattr(st, f)) for f, _ in _DiskannStats._fields_}
