"""On-disk graph formats of the LEANN HNSW backend.

Reads and writes the *compact CSR* HNSW index file that the reference produces with
``convert_hnsw_graph_to_csr`` and loads with ``faiss.read_index(path, IO_FLAG_MMAP,
HNSWIndexConfig{is_compact, is_recompute})``
(reference: packages/leann-backend-hnsw/leann_backend_hnsw/convert_to_csr.py:182-237 for the
field order, :494-548 for the CSR construction, hnsw_backend.py:145-151 for the load call).
Also reads the *original* (non-compact) faiss ``IHNf`` layout that the converter consumes
(convert_to_csr.py:264-301,411-479) and turns it into the same in-memory CSR.

Layout (little endian), SURVEY.md Appendix A.1::

    u32 "IHNf" | i32 d | i64 ntotal | i64 dummy | i64 dummy | u8 is_trained | i32 metric_type
    [f32 metric_arg if metric_type > 1]
    vec<f64> assign_probas | vec<i32> cum_nneighbor_per_level | vec<i32> levels
    u8 storage_is_compact(=1)
    vec<u64> compact_level_ptr | vec<u64> compact_node_offsets
    i32 entry_point | i32 max_level | i32 efConstruction | i32 efSearch | i32 upper_beam
    u32 storage_fourcc ("null" when embeddings are pruned)
    vec<i32> compact_neighbors_data
    [flat storage index payload when storage_fourcc != "null"]

``vec<T>`` is ``u64 count`` followed by ``count`` elements.  Neighbours of node ``i`` at level
``l`` are ``neighbors[level_ptr[p] : level_ptr[p+1]]`` with ``p = node_offsets[i] + l``.
"""

from __future__ import annotations

import struct
from dataclasses import dataclass, field
from pathlib import Path
from typing import BinaryIO, Optional

import numpy as np

FOURCC_IHNF = int.from_bytes(b"IHNf", "little")
FOURCC_NULL = int.from_bytes(b"null", "little")
FOURCC_FLAT_IP = int.from_bytes(b"IxFI", "little")
FOURCC_FLAT_L2 = int.from_bytes(b"IxF2", "little")
FOURCC_FLAT_LEGACY = int.from_bytes(b"IxFl", "little")

METRIC_INNER_PRODUCT = 0  # faiss enum; "mips" and "cosine" both map here (hnsw_backend.py:25-29)
METRIC_L2 = 1


class IndexFormatError(ValueError):
    pass


@dataclass
class HnswCsr:
    """In-memory compact-CSR HNSW graph (plus optional flat embedding storage)."""

    d: int
    ntotal: int
    metric_type: int
    levels: np.ndarray  # i32[N]   number of levels of node i (>= 1)
    level_ptr: np.ndarray  # u64[sum(levels+1)]
    node_offsets: np.ndarray  # u64[N+1]
    neighbors: np.ndarray  # i32[E]
    entry_point: int
    max_level: int
    ef_construction: int = 200
    ef_search: int = 16
    upper_beam: int = 1
    assign_probas: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float64))
    cum_nneighbor_per_level: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    metric_arg: float = 0.0
    is_trained: bool = True
    storage: Optional[np.ndarray] = None  # f32[N, d] when the file carries flat embeddings

    @property
    def is_pruned(self) -> bool:
        return self.storage is None

    def neighbors_of(self, node: int, level: int) -> np.ndarray:
        p = int(self.node_offsets[node]) + level
        return self.neighbors[int(self.level_ptr[p]) : int(self.level_ptr[p + 1])]

    def level0_degrees(self) -> np.ndarray:
        p = self.node_offsets[:-1].astype(np.int64)
        return (self.level_ptr[p + 1] - self.level_ptr[p]).astype(np.int64)

    def validate(self) -> None:
        n = self.ntotal
        if self.levels.shape[0] != n or self.node_offsets.shape[0] != n + 1:
            raise IndexFormatError("levels / node_offsets size does not match ntotal")
        if n:
            span = np.diff(self.node_offsets.astype(np.int64))
            if not np.array_equal(span, self.levels.astype(np.int64) + 1):
                raise IndexFormatError("node_offsets[i+1]-node_offsets[i] != levels[i]+1")
            if int(self.node_offsets[-1]) != self.level_ptr.shape[0]:
                raise IndexFormatError("level_ptr length mismatch")
            if np.any(np.diff(self.level_ptr.astype(np.int64)) < 0):
                raise IndexFormatError("level_ptr not monotone")
            if int(self.level_ptr[-1]) > self.neighbors.shape[0]:
                raise IndexFormatError("level_ptr points past neighbors")
            if self.neighbors.size and (self.neighbors.min() < 0 or self.neighbors.max() >= n):
                raise IndexFormatError("neighbor id out of range")
            if not (0 <= self.entry_point < n):
                raise IndexFormatError("entry_point out of range")
            if int(self.levels[self.entry_point]) != self.max_level + 1:
                raise IndexFormatError("entry_point level != max_level")


# ----------------------------------------------------------------------------------------
# primitive IO
# ----------------------------------------------------------------------------------------


def _rd(f: BinaryIO, fmt: str):
    size = struct.calcsize(fmt)
    b = f.read(size)
    if len(b) != size:
        raise IndexFormatError(f"unexpected end of file reading '{fmt}'")
    return struct.unpack(fmt, b)[0]


def _rd_vec(f: BinaryIO, dtype) -> np.ndarray:
    count = _rd(f, "<Q")
    dt = np.dtype(dtype)
    if count > (1 << 40):
        raise IndexFormatError(f"implausible vector length {count}")
    a = np.fromfile(f, dtype=dt, count=count)
    if a.shape[0] != count:
        raise IndexFormatError("truncated vector")
    return a


def _wr_vec(f: BinaryIO, a: np.ndarray, dtype) -> None:
    a = np.ascontiguousarray(a, dtype=np.dtype(dtype))
    f.write(struct.pack("<Q", a.shape[0]))
    a.tofile(f)


def _read_flat_storage(f: BinaryIO, fourcc: int, ntotal: int) -> np.ndarray:
    """faiss IndexFlat payload: header (d, ntotal, 2 dummies, is_trained, metric) + vec<float>.
    [recalled upstream faiss write_index; the reference only copies these bytes verbatim,
    convert_to_csr.py:619-634]"""
    if fourcc not in (FOURCC_FLAT_IP, FOURCC_FLAT_L2, FOURCC_FLAT_LEGACY):
        raise IndexFormatError(f"unsupported storage fourcc 0x{fourcc:08x}")
    d = _rd(f, "<i")
    n = _rd(f, "<q")
    _rd(f, "<q"), _rd(f, "<q")
    _rd(f, "<?")
    mt = _rd(f, "<i")
    if mt > 1:
        _rd(f, "<f")
    count = _rd(f, "<Q")
    if count == n * d:  # counted in floats
        x = np.fromfile(f, dtype=np.float32, count=n * d)
    elif count == n * d * 4:  # counted in bytes
        x = np.fromfile(f, dtype=np.float32, count=n * d)
    else:
        raise IndexFormatError("flat storage size mismatch")
    if x.shape[0] != n * d or n != ntotal:
        raise IndexFormatError("flat storage truncated")
    return x.reshape(n, d)


def _write_flat_storage(f: BinaryIO, x: np.ndarray, metric_type: int) -> None:
    n, d = x.shape
    f.write(struct.pack("<i", d))
    f.write(struct.pack("<q", n))
    f.write(struct.pack("<q", 1 << 20))
    f.write(struct.pack("<q", 1 << 20))
    f.write(struct.pack("<?", True))
    f.write(struct.pack("<i", metric_type))
    f.write(struct.pack("<Q", n * d))
    np.ascontiguousarray(x, dtype=np.float32).tofile(f)


# ----------------------------------------------------------------------------------------
# reader
# ----------------------------------------------------------------------------------------


def read_index(path) -> HnswCsr:
    """Read a compact-CSR or original-layout ``IHNf`` file into an :class:`HnswCsr`."""
    path = Path(path)
    if not path.exists():
        raise FileNotFoundError(f"HNSW index file not found at {path}")
    with open(path, "rb") as f:
        fourcc = _rd(f, "<I")
        if fourcc != FOURCC_IHNF:
            raise IndexFormatError(f"expected fourcc IHNf, got 0x{fourcc:08x}")
        d = _rd(f, "<i")
        ntotal = _rd(f, "<q")
        _rd(f, "<q"), _rd(f, "<q")
        is_trained = _rd(f, "<?")
        metric_type = _rd(f, "<i")
        metric_arg = _rd(f, "<f") if metric_type > 1 else 0.0
        assign_probas = _rd_vec(f, np.float64)
        cum_nn = _rd_vec(f, np.int32)
        levels = _rd_vec(f, np.int32)
        ntotal = int(levels.shape[0])
        pos = f.tell()
        flag = f.read(1)
        if flag == b"\x01":
            level_ptr = _rd_vec(f, np.uint64)
            node_offsets = _rd_vec(f, np.uint64)
            ep, ml, efc, efs, ub = (_rd(f, "<i") for _ in range(5))
            storage_fourcc = _rd(f, "<I")
            neighbors = _rd_vec(f, np.int32)
            storage = None
            if storage_fourcc != FOURCC_NULL:
                storage = _read_flat_storage_after_fourcc(f, storage_fourcc, ntotal)
        else:
            # original layout: an optional 0x00 flag byte, then offsets / padded neighbours
            if flag != b"\x00":
                f.seek(pos)
            offsets = _rd_vec(f, np.uint64)
            nb = _rd_vec(f, np.int32)
            ep, ml, efc, efs, ub = (_rd(f, "<i") for _ in range(5))
            level_ptr, node_offsets, neighbors = _csr_from_padded(levels, cum_nn, offsets, nb)
            storage = None
            tail = f.read(4)
            if len(tail) == 4:
                storage_fourcc = struct.unpack("<I", tail)[0]
                if storage_fourcc != FOURCC_NULL:
                    storage = _read_flat_storage_after_fourcc(f, storage_fourcc, ntotal)
    g = HnswCsr(
        d=d, ntotal=ntotal, metric_type=metric_type, levels=levels, level_ptr=level_ptr,
        node_offsets=node_offsets, neighbors=neighbors, entry_point=ep, max_level=ml,
        ef_construction=efc, ef_search=efs, upper_beam=ub, assign_probas=assign_probas,
        cum_nneighbor_per_level=cum_nn, metric_arg=metric_arg, is_trained=is_trained, storage=storage,
    )
    if ntotal:
        g.validate()
    return g


def _read_flat_storage_after_fourcc(f: BinaryIO, fourcc: int, ntotal: int) -> np.ndarray:
    return _read_flat_storage(f, fourcc, ntotal)


def _csr_from_padded(levels, cum_nn, offsets, nb):
    """Vectorised equivalent of the reference's per-node loop (convert_to_csr.py:494-548):
    drop the -1 padding of every (node, level) slot range and build level_ptr/node_offsets."""
    n = int(levels.shape[0])
    lv = levels.astype(np.int64)
    node_offsets = np.zeros(n + 1, dtype=np.uint64)
    node_offsets[1:] = np.cumsum(lv + 1)
    cum = cum_nn.astype(np.int64)

    def cum_at(level):
        level = np.asarray(level)
        idx = np.minimum(level, len(cum) - 1)
        return np.where(level < 0, 0, cum[idx]) if len(cum) else np.zeros_like(level)

    total_ptr = int(node_offsets[-1])
    level_ptr = np.zeros(total_ptr, dtype=np.uint64)
    # one entry per (node, level): slot begin/end in the padded array
    node_of = np.repeat(np.arange(n, dtype=np.int64), lv)
    first = np.repeat(node_offsets[:-1].astype(np.int64), lv)
    pidx = np.arange(node_of.shape[0], dtype=np.int64)
    starts = np.zeros(n + 1, dtype=np.int64)
    starts[1:] = np.cumsum(lv)
    lev_of = pidx - np.repeat(starts[:-1], lv)
    base = offsets.astype(np.int64)[node_of]
    beg = np.clip(base + cum_at(lev_of), 0, nb.shape[0])
    end = np.clip(base + cum_at(lev_of + 1), beg, nb.shape[0])
    valid = nb >= 0
    csum = np.zeros(nb.shape[0] + 1, dtype=np.int64)
    csum[1:] = np.cumsum(valid)
    cnt = csum[end] - csum[beg]
    # level_ptr position of (node, level) is node_offsets[node] + level
    ptr_pos = first + lev_of
    run = np.zeros(node_of.shape[0] + 1, dtype=np.int64)
    run[1:] = np.cumsum(cnt)
    level_ptr[ptr_pos] = run[:-1].astype(np.uint64)
    # closing pointer of each node = start of the next node's data
    last_pos = node_offsets[1:].astype(np.int64) - 1
    level_ptr[last_pos] = run[starts[1:]].astype(np.uint64)
    # gather valid neighbours slot by slot (slots of one node are contiguous and ordered)
    keep = np.zeros(nb.shape[0], dtype=bool)
    if node_of.shape[0]:
        # mark the union of [beg, end) ranges
        delta = np.zeros(nb.shape[0] + 1, dtype=np.int64)
        np.add.at(delta, beg, 1)
        np.add.at(delta, end, -1)
        keep = np.cumsum(delta[:-1]) > 0
    neighbors = nb[keep & valid].astype(np.int32)
    return level_ptr, node_offsets, neighbors


# ----------------------------------------------------------------------------------------
# writer
# ----------------------------------------------------------------------------------------


def write_index(path, g: HnswCsr, *, prune_embeddings: bool = True) -> None:
    """Write ``g`` in the compact-CSR layout (field order of convert_to_csr.py:196-237)."""
    M2 = int(g.cum_nneighbor_per_level[1]) if g.cum_nneighbor_per_level.shape[0] > 1 else 64
    with open(path, "wb") as f:
        f.write(struct.pack("<I", FOURCC_IHNF))
        f.write(struct.pack("<i", g.d))
        f.write(struct.pack("<q", g.ntotal))
        f.write(struct.pack("<q", 1 << 20))
        f.write(struct.pack("<q", 1 << 20))
        f.write(struct.pack("<?", g.is_trained))
        f.write(struct.pack("<i", g.metric_type))
        if g.metric_type > 1:
            f.write(struct.pack("<f", g.metric_arg))
        _wr_vec(f, g.assign_probas, np.float64)
        cum = g.cum_nneighbor_per_level
        if cum.shape[0] == 0:
            cum = np.array([0, M2], dtype=np.int32)
        _wr_vec(f, cum, np.int32)
        _wr_vec(f, g.levels, np.int32)
        f.write(struct.pack("<?", True))
        _wr_vec(f, g.level_ptr, np.uint64)
        _wr_vec(f, g.node_offsets, np.uint64)
        for v in (g.entry_point, g.max_level, g.ef_construction, g.ef_search, g.upper_beam):
            f.write(struct.pack("<i", int(v)))
        if prune_embeddings or g.storage is None:
            f.write(struct.pack("<I", FOURCC_NULL))
            _wr_vec(f, g.neighbors, np.int32)
        else:
            f.write(struct.pack("<I", FOURCC_FLAT_L2 if g.metric_type == METRIC_L2 else FOURCC_FLAT_IP))
            _wr_vec(f, g.neighbors, np.int32)
            _write_flat_storage(f, g.storage, g.metric_type)


def csr_from_adjacency(adj_levels: list[list[np.ndarray]], d: int, metric_type: int,
                       entry_point: int, M: int = 32, ef_construction: int = 200) -> HnswCsr:
    """Build an :class:`HnswCsr` from per-node, per-level neighbour arrays
    (``adj_levels[i][l]`` = neighbours of node i at level l)."""
    n = len(adj_levels)
    levels = np.array([len(a) for a in adj_levels], dtype=np.int32)
    node_offsets = np.zeros(n + 1, dtype=np.uint64)
    node_offsets[1:] = np.cumsum(levels.astype(np.int64) + 1)
    level_ptr = np.zeros(int(node_offsets[-1]), dtype=np.uint64)
    chunks = []
    pos = 0
    p = 0
    for a in adj_levels:
        for nb in a:
            level_ptr[p] = pos
            nb = np.asarray(nb, dtype=np.int32)
            chunks.append(nb)
            pos += nb.shape[0]
            p += 1
        level_ptr[p] = pos
        p += 1
    neighbors = np.concatenate(chunks).astype(np.int32) if chunks else np.zeros(0, np.int32)
    max_level = int(levels.max()) - 1 if n else -1
    cum = np.array([0, 2 * M] + [2 * M + M * (i + 1) for i in range(max(max_level, 0) + 1)], dtype=np.int32)
    return HnswCsr(d=d, ntotal=n, metric_type=metric_type, levels=levels, level_ptr=level_ptr,
                   node_offsets=node_offsets, neighbors=neighbors, entry_point=entry_point,
                   max_level=max_level, ef_construction=ef_construction, cum_nneighbor_per_level=cum)
