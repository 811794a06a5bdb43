"""Python handle on an HBM-resident ``lm_index`` (graph + search workspace).

Thin: every call goes straight through the C ABI of include/leann_mi355x.h.
"""

from __future__ import annotations

import ctypes as C
from typing import Callable, Optional

import numpy as np

from . import _lib
from ._lib import PqSearchParams, SearchParams, SearchStats, check
from .csr_format import HnswCsr


def _np_ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class Mi355xIndex:
    """Owns one ``lm_index``.  Create with :meth:`from_csr` or :meth:`read`."""

    def __init__(self, handle: C.c_void_p):
        self._h = handle
        self._lib = _lib.load()
        self._provider_keepalive = None
        self._table_keepalive = None
        info = _lib.IndexInfo()
        check(self._lib.lm_index_info(self._h, C.byref(info)), "lm_index_info")
        self.info = info

    # ---- construction -------------------------------------------------------------------
    @classmethod
    def from_csr(cls, g: HnswCsr, device: int = 0) -> "Mi355xIndex":
        lib = _lib.load()
        h = C.c_void_p()
        no = np.ascontiguousarray(g.node_offsets, np.uint64)
        lp = np.ascontiguousarray(g.level_ptr, np.uint64)
        nb = np.ascontiguousarray(g.neighbors, np.int32)
        lv = np.ascontiguousarray(g.levels, np.int32)
        check(lib.lm_index_create_from_csr(g.ntotal, g.d, g.metric_type, _np_ptr(no), _np_ptr(lp), lp.shape[0],
                                           _np_ptr(nb), nb.shape[0], _np_ptr(lv), g.entry_point, g.max_level,
                                           device, C.byref(h)), "lm_index_create_from_csr")
        idx = cls(h)
        if g.storage is not None:
            idx.attach_table(g.storage)
        return idx

    @classmethod
    def read(cls, path: str, device: int = 0) -> "Mi355xIndex":
        """faiss.read_index(path, IO_FLAG_MMAP, HNSWIndexConfig) equivalent (hnsw_backend.py:145-151)."""
        lib = _lib.load()
        h = C.c_void_p()
        check(lib.lm_index_read(str(path).encode(), device, C.byref(h)), "lm_index_read")
        return cls(h)

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.lm_index_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def _refresh(self):
        check(self._lib.lm_index_info(self._h, C.byref(self.info)))

    # ---- embedding sources ----------------------------------------------------------------
    def attach_table(self, table) -> None:
        """Stored embeddings: numpy (N, D) f32/f16 is uploaded; a CUDA/HIP torch tensor (N, d_padded)
        is borrowed in place."""
        if isinstance(table, np.ndarray):
            dt = _lib.DTYPE_F16 if table.dtype == np.float16 else _lib.DTYPE_F32
            t = np.ascontiguousarray(table, dtype=np.float16 if dt == _lib.DTYPE_F16 else np.float32)
            check(self._lib.lm_index_attach_table(self._h, _np_ptr(t), dt, t.shape[0], t.shape[1], 0), "lm_index_attach_table")
        else:  # torch tensor on device
            import torch

            assert table.is_cuda and table.is_contiguous()
            dt = _lib.DTYPE_F16 if table.dtype == torch.float16 else _lib.DTYPE_F32
            if table.shape[1] != self.info.d_padded:
                raise ValueError(f"device table must have row stride d_padded={self.info.d_padded}")
            check(self._lib.lm_index_attach_table(self._h, C.c_void_p(table.data_ptr()), dt, table.shape[0], self.info.d, 1),
                  "lm_index_attach_table")
            self._table_keepalive = table
        self._refresh()

    def set_provider(self, fn: Optional[Callable[[int, int, int], int]]) -> None:
        """``fn(d_ids_ptr, n, stream_ptr) -> device pointer (int) of fp32 [n][d_padded]`` (or raises).  A provider that offers a
        library-side form (``fn.native()`` -> ``lm_recompute`` handle: recompute.py) is attached as that -- the search loop then calls
        library code directly, no interpreter per round (lm_index_set_recompute); ``fn`` itself is what runs otherwise."""
        self.native_provider = False
        if fn is None:
            self._provider_keepalive = None
            check(self._lib.lm_index_set_recompute(self._h, None), "lm_index_set_recompute")
            check(self._lib.lm_index_set_provider(self._h, _lib.PROVIDER_FN(), None))
            self._refresh()
            return
        self._provider_error = None
        h = fn.native() if hasattr(fn, "native") else None
        if h is not None and self.info.d_padded == getattr(fn, "dp", -1):
            self._provider_keepalive = fn
            check(self._lib.lm_index_set_recompute(self._h, h), "lm_index_set_recompute")
            self.native_provider = True
            if hasattr(fn, "_attached"):  # the provider detaches itself from the indexes that hold its handle before it frees it
                fn._attached.add(self)
            self._refresh()
            return
        check(self._lib.lm_index_set_recompute(self._h, None), "lm_index_set_recompute")

        def _cb(_user, d_ids, n, out_pp, stream):
            try:
                out_pp[0] = int(fn(int(d_ids or 0), int(n), int(stream or 0)))
                return 0
            except BaseException as ex:  # noqa: BLE001 - re-raised after the C call returns
                self._provider_error = ex
                return 1

        cb = _lib.PROVIDER_FN(_cb)
        self._provider_keepalive = cb
        check(self._lib.lm_index_set_provider(self._h, cb, None))
        self._refresh()

    def set_hub_cache(self, ids: Optional[np.ndarray], embeddings=None) -> None:
        """Cache the embeddings (device torch tensor fp32 [n, d_padded]) of the nodes ``ids``; ``None`` clears it."""
        if ids is None or len(ids) == 0:
            check(self._lib.lm_index_set_hub_cache(self._h, None, 0, None), "lm_index_set_hub_cache")
            return
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        assert embeddings.is_cuda and embeddings.is_contiguous() and tuple(embeddings.shape) == (ids.shape[0], self.info.d_padded)
        check(self._lib.lm_index_set_hub_cache(self._h, _np_ptr(ids), ids.shape[0], C.c_void_p(embeddings.data_ptr())),
              "lm_index_set_hub_cache")

    def set_stream(self, stream_ptr: int) -> None:
        check(self._lib.lm_index_set_stream(self._h, C.c_void_p(stream_ptr)))

    def set_option(self, name: str, value: int) -> None:
        check(self._lib.lm_index_set_option(self._h, name.encode(), int(value)), "lm_index_set_option")

    def get_option(self, name: str) -> int:
        out = C.c_int64()
        check(self._lib.lm_index_get_option(self._h, name.encode(), C.byref(out)), "lm_index_get_option")
        return int(out.value)

    def event_overhead_us(self) -> float:
        out = C.c_double()
        check(self._lib.lm_index_event_overhead_us(self._h, C.byref(out)), "lm_index_event_overhead_us")
        return float(out.value)

    def set_profiling(self, on: bool) -> None:
        check(self._lib.lm_index_set_profiling(self._h, 1 if on else 0))

    # ---- search ---------------------------------------------------------------------------
    @staticmethod
    def make_params(ef: int = 64, beam: int = 1, check_relative_distance: bool = True, recompute: bool = True,
                    prune_ratio: float = 0.0, local_prune: bool = False, send_neigh_times_ratio: float = 0.0,
                    batch_size: int = 0, zmq_port: int = 0, max_batch: int = 0, recompute_memo: bool = True) -> SearchParams:
        """recompute_memo (default on, = lm_search_params_default): within one pass every node is recomputed at most once; identical results."""
        return SearchParams(ef, beam, 1 if check_relative_distance else 0, prune_ratio, 1 if local_prune else 0,
                            send_neigh_times_ratio, batch_size, zmq_port or 0, 1 if recompute else 0, max_batch,
                            1 if recompute_memo else 0)

    def _raise_provider(self, rc: int, what: str):
        err = getattr(self, "_provider_error", None)
        if rc == _lib.LM_EPROVIDER and err is not None:
            self._provider_error = None
            raise err
        check(rc, what)

    def search(self, queries: np.ndarray, k: int, params: SearchParams):
        """Host-pointer search == index.search(n, x, k, D, I, params) (hnsw_backend.py:241-248)."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != self.info.d:
            raise ValueError(f"query must be (B, {self.info.d}) float32")
        n = q.shape[0]
        dist = np.empty((n, k), dtype=np.float32)
        labels = np.empty((n, k), dtype=np.int64)
        rc = self._lib.lm_index_search(self._h, n, _np_ptr(q), k, _np_ptr(dist), _np_ptr(labels), C.byref(params))
        self._raise_provider(rc, "lm_index_search")
        return dist, labels

    def search_device(self, queries, k: int, params: SearchParams):
        """Device-resident search: ``queries`` is a CUDA/HIP torch tensor (B, D) f32; returns torch tensors."""
        import torch

        assert queries.is_cuda and queries.dtype == torch.float32 and queries.is_contiguous()
        n = queries.shape[0]
        dist = torch.empty((n, k), dtype=torch.float32, device=queries.device)
        labels = torch.empty((n, k), dtype=torch.int64, device=queries.device)
        rc = self._lib.lm_index_search_device(self._h, n, C.c_void_p(queries.data_ptr()), k, C.c_void_p(dist.data_ptr()),
                                              C.c_void_p(labels.data_ptr()), C.byref(params))
        self._raise_provider(rc, "lm_index_search_device")
        return dist, labels

    # ---- DiskANN-style path -----------------------------------------------------------------
    def attach_pq(self, codebooks: np.ndarray, codes: np.ndarray, chunk_offsets: Optional[np.ndarray] = None) -> None:
        """codebooks: (m, 256, d/m) float32; codes: (N, m) uint8.  With ``chunk_offsets`` (int32 [m + 1], the chunking of a stock
        DiskANN bundle: leann_amd/diskann_files.py) the codebooks are the flat per-chunk tables of lm_pq_attach_chunked."""
        cb = np.ascontiguousarray(codebooks, dtype=np.float32)
        cd = np.ascontiguousarray(codes, dtype=np.uint8)
        if chunk_offsets is not None:
            co = np.ascontiguousarray(chunk_offsets, dtype=np.int32)
            m = co.shape[0] - 1
            if m <= 0 or cd.shape != (self.info.ntotal, m) or cb.size != 256 * int(co[-1]):
                raise ValueError("chunk_offsets (m + 1), flat codebooks (256 * chunk_offsets[m]) and codes (N, m) do not fit together")
            check(self._lib.lm_pq_attach_chunked(self._h, m, _np_ptr(co), _np_ptr(cb), _np_ptr(cd), cd.shape[0]), "lm_pq_attach_chunked")
            return
        if cb.ndim != 3 or cb.shape[1] != 256 or cb.shape[0] * cb.shape[2] != self.info.d or cd.shape != (self.info.ntotal, cb.shape[0]):
            raise ValueError("codebooks must be (m, 256, d/m) and codes (N, m)")
        check(self._lib.lm_pq_attach(self._h, cb.shape[0], _np_ptr(cb), _np_ptr(cd), cd.shape[0]), "lm_pq_attach")

    @staticmethod
    def make_pq_params(complexity: int = 64, beam_width: int = 1, use_deferred_fetch: bool = False,
                       skip_search_reorder: bool = False, num_threads: int = 8, dedup_node_dis: bool = False,
                       prune_ratio: float = 0.0, batch_recompute: bool = False, use_global_pruning: bool = True) -> PqSearchParams:
        return PqSearchParams(complexity, beam_width, num_threads, 1 if use_deferred_fetch else 0,
                              1 if skip_search_reorder else 0, 0, 1 if dedup_node_dis else 0, prune_ratio,
                              1 if batch_recompute else 0, 1 if use_global_pruning else 0)

    def pq_search(self, queries: np.ndarray, k: int, params: PqSearchParams):
        """== StaticDiskFloatIndex.batch_search(...) -> (labels, distances) (diskann_backend.py:453-467)."""
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != self.info.d:
            raise ValueError(f"query must be (B, {self.info.d}) float32")
        n = q.shape[0]
        dist = np.empty((n, k), dtype=np.float32)
        labels = np.empty((n, k), dtype=np.int64)
        rc = self._lib.lm_pq_batch_search(self._h, n, _np_ptr(q), k, C.byref(params), _np_ptr(labels), _np_ptr(dist))
        self._raise_provider(rc, "lm_pq_batch_search")
        return labels, dist

    def pq_search_device(self, queries, k: int, params: PqSearchParams):
        import torch

        assert queries.is_cuda and queries.dtype == torch.float32 and queries.is_contiguous()
        n = queries.shape[0]
        dist = torch.empty((n, k), dtype=torch.float32, device=queries.device)
        labels = torch.empty((n, k), dtype=torch.int64, device=queries.device)
        rc = self._lib.lm_pq_batch_search_device(self._h, n, C.c_void_p(queries.data_ptr()), k, C.byref(params),
                                                 C.c_void_p(labels.data_ptr()), C.c_void_p(dist.data_ptr()))
        self._raise_provider(rc, "lm_pq_batch_search_device")
        return labels, dist

    def stats(self) -> dict:
        st = SearchStats()
        check(self._lib.lm_index_get_stats(self._h, C.byref(st)))
        return {f: getattr(st, f) for f, _ in SearchStats._fields_}
