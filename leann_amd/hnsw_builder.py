"""Host-side HNSW graph construction (index build time; not the query hot path).

Python front-end of ``csrc/hnsw_build.cpp``: the published HNSW insertion algorithm with the
reference's build parameters (M=32, efConstruction=200: hnsw_backend.py:54-55) emitting the
compact-CSR arrays of convert_to_csr.py:494-548.
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from .csr_format import METRIC_INNER_PRODUCT, METRIC_L2, HnswCsr

_blib = None


def _load():
    global _blib
    if _blib is None:
        if not _lib.BUILD_LIB_PATH.exists():
            raise _lib.LeannMi355xError(f"{_lib.BUILD_LIB_PATH} is missing: run __graft_entry__.build()")
        b = C.CDLL(str(_lib.BUILD_LIB_PATH))
        b.lm_hnsw_build.restype = C.c_void_p
        b.lm_hnsw_build.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_int32]
        b.lm_hnsw_build_sizes.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        b.lm_hnsw_build_sizes.restype = None
        b.lm_hnsw_build_export.argtypes = [C.c_void_p] * 5
        b.lm_hnsw_build_export.restype = None
        b.lm_hnsw_build_free.argtypes = [C.c_void_p]
        b.lm_hnsw_build_free.restype = None
        _blib = b
    return _blib


def default_threads(cap: int = 16) -> int:
    """Usable cores (affinity / cgroup aware), capped: the builder takes per-node spin locks and
    oversubscribed threads make it crawl."""
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
    return max(1, min(n, cap))


def build_hnsw(data: np.ndarray, metric: str = "mips", M: int = 32, ef_construction: int = 200,
               seed: int = 12345, num_threads: int = 0) -> HnswCsr:
    """Build an HNSW graph over ``data`` (N, D) float32.  ``metric``: "mips" | "cosine" | "l2"
    (cosine data must already be L2-normalised, as hnsw_backend.py:86-87 does)."""
    metric = metric.lower()
    if metric not in ("mips", "cosine", "l2"):
        raise ValueError(f"Unsupported distance_metric '{metric}'.")
    mt = METRIC_L2 if metric == "l2" else METRIC_INNER_PRODUCT
    x = np.ascontiguousarray(data, dtype=np.float32)
    n, d = x.shape
    if n == 0:
        return HnswCsr(d=d, ntotal=0, metric_type=mt, levels=np.zeros(0, np.int32), level_ptr=np.zeros(0, np.uint64),
                       node_offsets=np.zeros(1, np.uint64), neighbors=np.zeros(0, np.int32), entry_point=-1, max_level=-1,
                       ef_construction=ef_construction)
    b = _load()
    h = b.lm_hnsw_build(x.ctypes.data, n, d, mt, M, ef_construction, seed, num_threads or default_threads())
    if not h:
        raise ValueError("lm_hnsw_build rejected its arguments")
    try:
        nptr, nedge, ep, ml = C.c_int64(), C.c_int64(), C.c_int32(), C.c_int32()
        b.lm_hnsw_build_sizes(h, C.byref(nptr), C.byref(nedge), C.byref(ep), C.byref(ml))
        levels = np.empty(n, np.int32)
        node_offsets = np.empty(n + 1, np.uint64)
        level_ptr = np.empty(nptr.value, np.uint64)
        neighbors = np.empty(max(nedge.value, 1), np.int32)
        b.lm_hnsw_build_export(h, levels.ctypes.data, node_offsets.ctypes.data, level_ptr.ctypes.data, neighbors.ctypes.data)
        neighbors = neighbors[: nedge.value]
    finally:
        b.lm_hnsw_build_free(h)
    cum = np.array([0, 2 * M] + [2 * M + M * (i + 1) for i in range(ml.value + 1)], dtype=np.int32)
    return HnswCsr(d=d, ntotal=n, metric_type=mt, levels=levels, level_ptr=level_ptr, node_offsets=node_offsets,
                   neighbors=neighbors, entry_point=ep.value, max_level=ml.value, ef_construction=ef_construction,
                   cum_nneighbor_per_level=cum)
