#!/usr/bin/env python
"""Writes tests/golden/stock_diskann/: a small bundle in the PUBLIC microsoft/DiskANN file layout, the way the stock
leann-backend-diskann writes one in its default (non-recompute) mode (file names: diskann_backend.py:151-162; the reference's
native writer is an absent submodule, and its tests pin only the two smallest files, tests/test_diskann_partition.py:258-281).
Every byte is packed HERE with struct -- independently of leann_amd/diskann_files.py, whose readers the tests run over these files.

    python tests/golden/make_golden_diskann.py          # deterministic; rewrites the fixture in place

Content: 200 vectors, d = 24, inner-product metric -> stored in DiskANN's L2 form (scaled by 1 / max_norm, one extra coordinate:
dim 25), PQ with 6 chunks over those 25 dimensions (5,4,4,4,4,4), graph degree <= 12, medoid = the node nearest the mean."""
import struct
from pathlib import Path

import numpy as np

OUT = Path(__file__).resolve().parent / "stock_diskann"
SECTOR = 4096


def bin_bytes(arr, fmt):
    a = np.ascontiguousarray(arr)
    if a.ndim == 1:
        a = a.reshape(-1, 1)
    return struct.pack("<ii", a.shape[0], a.shape[1]) + a.astype(fmt).tobytes()


def main():
    OUT.mkdir(exist_ok=True)
    rng = np.random.default_rng(20260921)
    n, d, nchunks, deg = 200, 24, 6, 12
    centres = rng.standard_normal((8, d)).astype(np.float32)
    x = (centres[rng.integers(0, 8, n)] + 0.35 * rng.standard_normal((n, d))).astype(np.float32)
    x *= rng.uniform(0.5, 1.5, (n, 1)).astype(np.float32)  # unequal norms: the MIPS transform matters
    norms = np.linalg.norm(x, axis=1)
    max_norm = np.float32(norms.max())
    xs = x / max_norm
    aug = np.sqrt(np.maximum(0.0, 1.0 - (xs * xs).sum(1))).astype(np.float32)
    y = np.concatenate([xs, aug[:, None]], 1).astype(np.float32)  # [n, 25], unit norm: L2 here ranks like IP there
    dim = d + 1
    # graph: 8 nearest neighbours in the stored space + 4 random long-range edges per node, medoid = nearest the mean
    d2 = ((y[:, None, :] - y[None, :, :]) ** 2).sum(-1)
    np.fill_diagonal(d2, np.inf)
    nbrs = np.argsort(d2, axis=1, kind="stable")[:, :deg]
    nbrs[:, :4] = rng.integers(0, n, (n, 4))  # long-range edges (what Vamana's alpha pruning keeps): a kNN graph of clustered data is disconnected
    nbrs[:, 0] = np.where(nbrs[:, 0] == np.arange(n), (np.arange(n) + 1) % n, nbrs[:, 0])
    degs = rng.integers(deg - 4, deg + 1, n)  # ragged lists
    medoid = int(np.argmin(((y - y.mean(0)) ** 2).sum(1)))
    # PQ: centre, chunk, 256 pivots per chunk (a few Lloyd steps), codes
    centroid = y.mean(0).astype(np.float32)
    yc = y - centroid
    chunk = np.array([0, 5, 9, 13, 17, 21, 25], np.uint32)
    pivots = np.zeros((256, dim), np.float32)
    codes = np.zeros((n, nchunks), np.uint8)
    for c in range(nchunks):
        lo, hi = int(chunk[c]), int(chunk[c + 1])
        sub = yc[:, lo:hi]
        cen = sub[rng.integers(0, n, 256)].copy() + 1e-3 * rng.standard_normal((256, hi - lo)).astype(np.float32)
        for _ in range(4):
            a = ((sub[:, None, :] - cen[None, :, :]) ** 2).sum(-1).argmin(1)
            for k in range(256):
                if np.any(a == k):
                    cen[k] = sub[a == k].mean(0)
        codes[:, c] = ((sub[:, None, :] - cen[None, :, :]) ** 2).sum(-1).argmin(1).astype(np.uint8)
        pivots[:, lo:hi] = cen
    # ---- <p>_pq_pivots.bin: 4096-byte metadata block (bin of 4 u64 offsets), pivots, centroid, chunk offsets ----
    body_p, body_c, body_o = bin_bytes(pivots, "<f4"), bin_bytes(centroid, "<f4"), bin_bytes(chunk, "<u4")
    offs = [SECTOR, SECTOR + len(body_p), SECTOR + len(body_p) + len(body_c), SECTOR + len(body_p) + len(body_c) + len(body_o)]
    meta = bin_bytes(np.array(offs, np.uint64), "<u8")
    (OUT / "fx_pq_pivots.bin").write_bytes(meta + b"\0" * (SECTOR - len(meta)) + body_p + body_c + body_o)
    (OUT / "fx_pq_compressed.bin").write_bytes(bin_bytes(codes, "u1"))
    (OUT / "fx_disk.index_medoids.bin").write_bytes(struct.pack("<III", 1, 1, medoid))
    (OUT / "fx_disk.index_max_base_norm.bin").write_bytes(struct.pack("<IIf", 1, 1, float(max_norm)))
    # ---- <p>_disk.index: sector 0 = bin of u64 metadata, then sectors of node records [25 f32 | u32 degree | 12 u32 ids] ----
    max_node_len = dim * 4 + 4 + 4 * deg
    nps = SECTOR // max_node_len
    nsec = (n + nps - 1) // nps
    size = SECTOR * (1 + nsec)
    meta = bin_bytes(np.array([n, dim, medoid, max_node_len, nps, 0, 0, 0, size], np.uint64), "<u8")
    out = bytearray(meta + b"\0" * (SECTOR - len(meta)))
    for s in range(nsec):
        sec = bytearray()
        for i in range(s * nps, min(n, (s + 1) * nps)):
            k = int(degs[i])
            ids = list(map(int, nbrs[i, :k])) + [0] * (deg - k)
            sec += y[i].astype("<f4").tobytes() + struct.pack("<I", k) + struct.pack(f"<{deg}I", *ids)
        out += sec + b"\0" * (SECTOR - len(sec))
    (OUT / "fx_disk.index").write_bytes(bytes(out))
    # what the readers must reproduce, in the CALLER's space
    np.savez(OUT / "expected.npz", x=x, codes=codes, chunk=chunk.astype(np.int32), medoid=medoid, max_norm=max_norm, degs=degs.astype(np.int32),
             nbrs=nbrs.astype(np.int32), recon=((pivots + centroid)[:, :d] * max_norm).astype(np.float32))
    print("wrote", sorted(p.name for p in OUT.iterdir()))


if __name__ == "__main__":
    main()
