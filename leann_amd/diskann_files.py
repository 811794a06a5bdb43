"""Readers (and fixture writers) for the files of a STOCK DiskANN bundle, so that an index built by ``leann-backend-diskann``
can be served by ``mi355x_diskann``.

File names: ``diskann_backend.py:151-162,220,328-329`` (reference).  Layouts: the reference tree holds none of them -- its
native code is an empty submodule (``yichuan-w/DiskANN``) -- so what is read here is the PUBLIC microsoft/DiskANN ``bin`` layout,
restated; the two small files the reference's own tests spell out are checked against that description
(``tests/test_diskann_partition.py:258-281``: medoids = u32 1, u32 1, u32 id; max_base_norm = u32 1, u32 1, f32 norm):

* every ``.bin``: ``i32 npts, i32 ndims`` then ``npts * ndims`` values, row major;
* ``<p>_pq_pivots.bin``: a 4096-byte metadata block holding a bin of u64 section offsets, then three (four in older writers) bins:
  the 256 x dim table of FULL-dimension pivots (f32), the dim-vector centroid the data was centred by (f32), the
  ``nchunks + 1`` chunk offsets (u32): chunk c of a code addresses dimensions ``[off[c], off[c+1])`` of pivot ``code[c]``;
* ``<p>_pq_compressed.bin``: bin of u8, ``npts x nchunks``;
* ``<p>_disk.index`` (kept by the reference only when the index is NOT built for recompute, ``diskann_backend.py:268-284``): sector 0
  = a bin of u64 metadata ``[npts, dim, medoid, max_node_len, nnodes_per_sector, ...]``, then 4096-byte sectors of
  ``nnodes_per_sector`` node records ``[dim x f32 coordinates | u32 degree | u32 neighbours ...]`` of ``max_node_len`` bytes
  (a record larger than a sector takes ``ceil(max_node_len / 4096)`` sectors of its own);
* inner-product indexes are stored in DiskANN's L2 form: every vector scaled by ``1 / max_base_norm`` with one extra
  coordinate ``sqrt(1 - |x|^2 / max_base_norm^2)`` (dim = d + 1); ``<p>_disk.index_max_base_norm.bin`` keeps the constant.

NOT supported, and said so loudly: ``<p>_disk_graph.index`` / ``<p>_partition.bin`` -- the graph of a recompute-mode
(``is_recompute=True``) stock bundle -- are the fork's private format; such a bundle has no public graph file left
(``_disk.index`` is deleted, ``diskann_backend.py:144-200``), so only its PQ table, codes and medoid can be reused here.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass
from pathlib import Path
from typing import Optional

import numpy as np

SECTOR = 4096
N_CENTROIDS = 256


class DiskannFormatError(ValueError):
    pass


# ---- the generic bin container ------------------------------------------------------------------
def read_bin(path, dtype, offset: int = 0) -> np.ndarray:
    """``i32 npts, i32 ndims, data`` at byte ``offset`` -> array [npts, ndims]."""
    dt = np.dtype(dtype)
    with open(path, "rb") as f:
        f.seek(offset)
        hdr = f.read(8)
        if len(hdr) != 8:
            raise DiskannFormatError(f"{path}: truncated bin header at offset {offset}")
        npts, ndims = struct.unpack("<ii", hdr)
        if npts < 0 or ndims < 0:
            raise DiskannFormatError(f"{path}: negative shape ({npts}, {ndims})")
        a = np.fromfile(f, dtype=dt, count=npts * ndims)
    if a.size != npts * ndims:
        raise DiskannFormatError(f"{path}: {a.size} of {npts * ndims} values present")
    return a.reshape(npts, ndims)


def write_bin(path, arr: np.ndarray, offset: int = 0, mode: str = "wb") -> int:
    """Writes ``arr`` [npts, ndims] as a bin at ``offset``; returns the bytes written (header included)."""
    a = np.ascontiguousarray(arr)
    if a.ndim == 1:
        a = a.reshape(-1, 1)
    with open(path, mode) as f:
        f.seek(offset)
        f.write(struct.pack("<ii", a.shape[0], a.shape[1]))
        f.write(a.tobytes())
    return 8 + a.nbytes


# ---- small auxiliary files (pinned by tests/test_diskann_partition.py:258-281 of the reference) -----
def read_medoids(path) -> np.ndarray:
    m = read_bin(path, np.uint32)
    if m.shape[1] != 1 or m.shape[0] < 1:
        raise DiskannFormatError(f"{path}: expected an (nshards, 1) u32 bin, got {m.shape}")
    return m[:, 0].astype(np.int64)


def write_medoids(path, medoids) -> None:
    write_bin(path, np.asarray(medoids, np.uint32).reshape(-1, 1))


def read_max_base_norm(path) -> float:
    v = read_bin(path, np.float32)
    if v.shape != (1, 1) or not np.isfinite(v[0, 0]) or v[0, 0] <= 0:
        raise DiskannFormatError(f"{path}: expected one positive finite f32, got {v}")
    return float(v[0, 0])


def write_max_base_norm(path, norm: float) -> None:
    write_bin(path, np.array([[norm]], np.float32))


# ---- product quantiser --------------------------------------------------------------------------
def read_pq_pivots(path):
    """-> (pivots f32 [256, dim], centroid f32 [dim], chunk_offsets i32 [nchunks + 1])."""
    meta = read_bin(path, np.uint64)
    if meta.shape[1] != 1 or meta.shape[0] not in (4, 5):
        raise DiskannFormatError(f"{path}: metadata bin has shape {meta.shape}; expected 4 (or 5, older writers) section offsets")
    offs = [int(v) for v in meta[:, 0]]
    pivots = read_bin(path, np.float32, offs[0])
    if pivots.shape[0] != N_CENTROIDS:
        raise DiskannFormatError(f"{path}: {pivots.shape[0]} pivots, expected {N_CENTROIDS}")
    dim = pivots.shape[1]
    centroid = read_bin(path, np.float32, offs[1])
    if centroid.size != dim:
        raise DiskannFormatError(f"{path}: centroid has {centroid.size} values for dimension {dim}")
    # five-section files carry a dimension rearrangement between the centroid and the chunk offsets (identity in every writer known)
    chunk = read_bin(path, np.uint32, offs[3] if len(offs) == 5 else offs[2]).reshape(-1).astype(np.int64)
    if len(offs) == 5:
        perm = read_bin(path, np.uint32, offs[2]).reshape(-1)
        if not np.array_equal(perm, np.arange(dim)):
            raise DiskannFormatError(f"{path}: non-identity dimension rearrangement is not supported")
    if chunk.size < 2 or chunk[0] != 0 or chunk[-1] != dim or np.any(np.diff(chunk) < 0):
        raise DiskannFormatError(f"{path}: bad chunk offsets {chunk[:8]}... for dimension {dim}")
    return pivots, centroid.reshape(-1), chunk.astype(np.int32)


def write_pq_pivots(path, pivots: np.ndarray, centroid: np.ndarray, chunk_offsets: np.ndarray) -> None:
    """The four-section layout current DiskANN writers produce (fixtures; our own builder keeps <stem>_pq.npz)."""
    pivots = np.ascontiguousarray(pivots, np.float32)
    assert pivots.shape[0] == N_CENTROIDS
    offs = [SECTOR]
    with open(path, "wb") as f:
        f.write(b"\0" * SECTOR)
    offs.append(offs[-1] + write_bin(path, pivots, offs[-1], "r+b"))
    offs.append(offs[-1] + write_bin(path, np.asarray(centroid, np.float32).reshape(-1, 1), offs[-1], "r+b"))
    offs.append(offs[-1] + write_bin(path, np.asarray(chunk_offsets, np.uint32).reshape(-1, 1), offs[-1], "r+b"))
    write_bin(path, np.asarray(offs, np.uint64).reshape(-1, 1), 0, "r+b")


def read_pq_compressed(path) -> np.ndarray:
    return read_bin(path, np.uint8)


# ---- the disk index (graph + full-precision vectors) ----------------------------------------------
def read_disk_index(path, dtype=np.float32):
    """-> (vectors [npts, dim] in the file's (possibly MIPS-transformed) space, degrees i32 [npts], neighbors i32 [sum deg], medoid)."""
    meta = read_bin(path, np.uint64)
    if meta.shape[1] != 1 or meta.shape[0] < 5:
        raise DiskannFormatError(f"{path}: metadata bin has shape {meta.shape}")
    npts, dim, medoid, max_node_len, nps = (int(v) for v in meta[:5, 0])
    esz = np.dtype(dtype).itemsize
    if max_node_len < dim * esz + 4 or (max_node_len - dim * esz - 4) % 4:
        raise DiskannFormatError(f"{path}: max_node_len {max_node_len} does not fit dimension {dim}")
    max_deg = (max_node_len - dim * esz - 4) // 4
    raw = np.fromfile(path, dtype=np.uint8, offset=SECTOR)
    if nps > 0:
        nsec = (npts + nps - 1) // nps
        if raw.size < nsec * SECTOR:
            raise DiskannFormatError(f"{path}: {raw.size} bytes of node sectors, {nsec * SECTOR} expected")
        rec = raw[: nsec * SECTOR].reshape(nsec, SECTOR)[:, : nps * max_node_len].reshape(nsec * nps, max_node_len)[:npts]
    else:
        spn = (max_node_len + SECTOR - 1) // SECTOR
        if raw.size < npts * spn * SECTOR:
            raise DiskannFormatError(f"{path}: {raw.size} bytes of node sectors, {npts * spn * SECTOR} expected")
        rec = raw[: npts * spn * SECTOR].reshape(npts, spn * SECTOR)[:, :max_node_len]
    rec = np.ascontiguousarray(rec)
    vec = rec[:, : dim * esz].copy().view(dtype).reshape(npts, dim)
    deg = rec[:, dim * esz : dim * esz + 4].copy().view(np.uint32).reshape(npts).astype(np.int64)
    if npts and int(deg.max()) > max_deg:
        raise DiskannFormatError(f"{path}: a node claims {int(deg.max())} neighbours, records hold {max_deg}")
    nb = rec[:, dim * esz + 4 :].copy().view(np.uint32).reshape(npts, max_deg)
    mask = np.arange(max_deg)[None, :] < deg[:, None]
    neighbors = nb[mask].astype(np.int64)
    if neighbors.size and (int(neighbors.max()) >= npts):
        raise DiskannFormatError(f"{path}: neighbour id {int(neighbors.max())} out of range")
    if npts and not 0 <= medoid < npts:
        raise DiskannFormatError(f"{path}: medoid {medoid} out of range")
    return vec, deg.astype(np.int32), neighbors.astype(np.int32), medoid


def write_disk_index(path, vectors: np.ndarray, adjacency: list, medoid: int, max_degree: Optional[int] = None) -> None:
    """Fixture writer in the same layout (vectors already in the stored space)."""
    v = np.ascontiguousarray(vectors, np.float32)
    npts, dim = v.shape
    max_deg = max_degree if max_degree is not None else max((len(a) for a in adjacency), default=0)
    max_node_len = dim * 4 + 4 + 4 * max_deg
    nps = SECTOR // max_node_len
    rec = np.zeros((npts, max_node_len), np.uint8)
    rec[:, : dim * 4] = v.view(np.uint8).reshape(npts, dim * 4)
    for i, a in enumerate(adjacency):
        a = np.asarray(a, np.uint32)
        rec[i, dim * 4 : dim * 4 + 4] = np.array([a.size], np.uint32).view(np.uint8)
        rec[i, dim * 4 + 4 : dim * 4 + 4 + 4 * a.size] = a.view(np.uint8)
    if nps > 0:
        nsec = (npts + nps - 1) // nps
        body = np.zeros((nsec, SECTOR), np.uint8)
        pad = np.zeros((nsec * nps, max_node_len), np.uint8)
        pad[:npts] = rec
        body[:, : nps * max_node_len] = pad.reshape(nsec, nps * max_node_len)
    else:
        spn = (max_node_len + SECTOR - 1) // SECTOR
        body = np.zeros((npts, spn * SECTOR), np.uint8)
        body[:, :max_node_len] = rec
    size = SECTOR + body.size
    meta = np.array([npts, dim, medoid, max_node_len, nps, 0, 0, 0, size], np.uint64)
    with open(path, "wb") as f:
        f.write(b"\0" * SECTOR)
    write_bin(path, meta.reshape(-1, 1), 0, "r+b")
    with open(path, "r+b") as f:
        f.seek(SECTOR)
        f.write(body.tobytes())


# ---- the bundle, mapped onto the library's inputs ----------------------------------------------------
@dataclass
class StockBundle:
    d: int                        # the caller's dimension (without DiskANN's MIPS augmentation coordinate)
    metric: str
    chunk_offsets: np.ndarray     # i32 [m + 1], m % 4 == 0 (padded with empty chunks), offsets <= d
    codebooks: np.ndarray         # f32 flat: chunk j = 256 x len_j at 256 * chunk_offsets[j]
    codes: np.ndarray             # u8 [N, m]
    medoid: int
    max_base_norm: Optional[float]
    vectors: Optional[np.ndarray] = None    # [N, d] in the CALLER's space (when <p>_disk.index is present)
    degrees: Optional[np.ndarray] = None
    neighbors: Optional[np.ndarray] = None

    def graph(self):
        """The stock graph as a one-level compact-CSR graph entered at the medoid (None without <p>_disk.index)."""
        if self.degrees is None:
            return None
        from ._lib import METRIC_INNER_PRODUCT, METRIC_L2
        from .csr_format import HnswCsr

        METRIC_TYPE = {"mips": METRIC_INNER_PRODUCT, "cosine": METRIC_INNER_PRODUCT, "l2": METRIC_L2}
        n = self.degrees.shape[0]
        cs = np.cumsum(self.degrees.astype(np.int64))
        level_ptr = np.zeros(2 * n, np.uint64)
        level_ptr[0::2] = cs - self.degrees
        level_ptr[1::2] = cs
        md = int(self.degrees.max()) if n else 0
        return HnswCsr(d=self.d, ntotal=n, metric_type=METRIC_TYPE[self.metric], levels=np.ones(n, np.int32), level_ptr=level_ptr,
                       node_offsets=np.arange(n + 1, dtype=np.uint64) * 2, neighbors=self.neighbors.astype(np.int32), entry_point=int(self.medoid),
                       max_level=0 if n else -1, ef_construction=0, cum_nneighbor_per_level=np.array([0, md, md], np.int32))


def load_stock_bundle(prefix, d: int, metric: str) -> StockBundle:
    """``prefix`` = ``<dir>/<stem>`` of a bundle written by the stock DiskANN backend.  PQ pivots / codes / medoid are required;
    ``<prefix>_disk.index`` (graph + vectors) is read when present.  Inner-product bundles (dimension d + 1 on disk) are mapped back
    to the caller's space: codebooks = (pivot + centroid)[:d] * max_base_norm, so that -q . c ranks like the stored L2 form does
    (the augmentation coordinate meets a zero in the query and drops out)."""
    prefix = str(prefix)
    metric = metric.lower()
    for suffix in ("_disk_graph.index", "_partition.bin"):
        if Path(prefix + suffix).exists() and not Path(prefix + "_disk.index").exists():
            import logging

            logging.getLogger(__name__).warning(
                f"{prefix}{suffix}: the recompute-mode graph files of the stock DiskANN backend are the fork's private format and are not "
                "read; without <prefix>_disk.index this bundle contributes its PQ table, codes and medoid only")
            break
    pivots, centroid, chunk = read_pq_pivots(prefix + "_pq_pivots.bin")
    codes = read_pq_compressed(prefix + "_pq_compressed.bin")
    dim = pivots.shape[1]
    nchunks = chunk.shape[0] - 1
    if codes.shape[1] != nchunks:
        raise DiskannFormatError(f"{prefix}: {codes.shape[1]} code bytes per vector but {nchunks} chunks")
    norm_file = Path(prefix + "_disk.index_max_base_norm.bin")
    norm = read_max_base_norm(norm_file) if norm_file.exists() else None
    augmented = dim == d + 1
    if dim != d and not augmented:
        raise DiskannFormatError(f"{prefix}: PQ pivots have dimension {dim}, the index {d}")
    if augmented and norm is None:
        raise DiskannFormatError(f"{prefix}: dimension {dim} = d + 1 (inner-product form) but no _disk.index_max_base_norm.bin")
    scale = norm if augmented else 1.0
    full = (pivots + centroid[None, :]) * np.float32(scale)  # reconstruction table in the caller's units
    chunk = np.minimum(chunk, d).astype(np.int32)            # the augmentation coordinate carries no query mass
    m = (nchunks + 3) // 4 * 4
    chunk_p = np.concatenate([chunk, np.full(m - nchunks, chunk[-1], np.int32)])
    cb = np.concatenate([np.ascontiguousarray(full[:, chunk_p[j] : chunk_p[j + 1]]).reshape(-1) for j in range(m)]) if m else np.zeros(0, np.float32)
    codes_p = np.zeros((codes.shape[0], m), np.uint8)
    codes_p[:, :nchunks] = codes
    med_file = Path(prefix + "_disk.index_medoids.bin")
    medoid = int(read_medoids(med_file)[0]) if med_file.exists() else -1
    b = StockBundle(d=d, metric=metric, chunk_offsets=chunk_p, codebooks=cb.astype(np.float32), codes=codes_p, medoid=medoid, max_base_norm=norm)
    disk = Path(prefix + "_disk.index")
    if disk.exists():
        vec, deg, nbr, med = read_disk_index(disk)
        if vec.shape[1] != dim or vec.shape[0] != codes.shape[0]:
            raise DiskannFormatError(f"{disk}: {vec.shape} vectors for {codes.shape[0]} codes of dimension {dim}")
        b.vectors = np.ascontiguousarray(vec[:, :d] * np.float32(scale))
        b.degrees, b.neighbors = deg, nbr
        if b.medoid < 0:
            b.medoid = med
    if b.medoid < 0:
        raise DiskannFormatError(f"{prefix}: no medoid (neither _disk.index_medoids.bin nor _disk.index)")
    return b
