"""Product quantiser for the DiskANN-style path (index build time) + flat-graph helper.

DiskANN keeps PQ-compressed vectors in memory and traverses on PQ distances
(diskann_backend.py:444-449; compression budget `search_memory_maximum` ~ N*D*4/10 bytes,
:105-111 -> m ~ D*4/10 bytes per vector).  Training (k-means per sub-space) and encoding are
torch ops (any device); the traversal kernel lives in csrc/lm_pq_impl.h.
"""

from __future__ import annotations

import numpy as np
import torch

from .csr_format import HnswCsr


@torch.no_grad()
def train_pq(x: torch.Tensor, m: int, iters: int = 12, sample: int = 131072, seed: int = 0) -> torch.Tensor:
    """k-means (256 centroids) in each of the m sub-spaces.  Returns codebooks [m, 256, d/m] float32."""
    n, d = x.shape
    if d % m:
        raise ValueError("m must divide d")
    g = torch.Generator(device="cpu").manual_seed(seed)
    idx = torch.randperm(n, generator=g)[: min(n, sample)].to(x.device)
    xs = x[idx].float().view(-1, m, d // m).transpose(0, 1).contiguous()  # [m, s, dsub]
    s = xs.shape[1]
    init = torch.randperm(s, generator=g)[:256].to(x.device)
    if s < 256:
        init = torch.arange(256, device=x.device) % s
    cb = xs[:, init].clone()  # [m, 256, dsub]
    for _ in range(iters):
        d2 = (xs * xs).sum(-1, keepdim=True) - 2 * xs @ cb.transpose(1, 2) + (cb * cb).sum(-1)[:, None, :]
        a = d2.argmin(-1)  # [m, s]
        onehot = torch.zeros((m, s, 256), device=x.device, dtype=xs.dtype)
        onehot.scatter_(2, a.unsqueeze(-1), 1.0)
        cnt = onehot.sum(1)  # [m, 256]
        new = onehot.transpose(1, 2) @ xs  # [m, 256, dsub]
        cb = torch.where(cnt.unsqueeze(-1) > 0, new / cnt.clamp(min=1).unsqueeze(-1), cb)
    return cb.contiguous()


@torch.no_grad()
def encode_pq(x: torch.Tensor, codebooks: torch.Tensor, block: int = 65536) -> torch.Tensor:
    """Nearest centroid per sub-space -> codes [N, m] uint8."""
    n, d = x.shape
    m = codebooks.shape[0]
    out = torch.empty((n, m), dtype=torch.uint8, device=x.device)
    cbn = (codebooks * codebooks).sum(-1)  # [m, 256]
    for b0 in range(0, n, block):
        xs = x[b0 : b0 + block].float().view(-1, m, d // m).transpose(0, 1)  # [m, b, dsub]
        d2 = -2 * xs @ codebooks.transpose(1, 2) + cbn[:, None, :]
        out[b0 : b0 + block] = d2.argmin(-1).transpose(0, 1).to(torch.uint8)
    return out


def flat_graph(g: HnswCsr, x) -> HnswCsr:
    """Single-level (Vamana-style) graph from the level-0 lists of ``g``, entered at the medoid
    (the node closest to the mean; DiskANN's `<prefix>_disk.index_medoids.bin`).  ``x``: [N, D] numpy array or torch tensor
    (any device; the 10M-chunk configuration passes the HBM-resident table)."""
    n = g.ntotal
    p0 = g.node_offsets[:-1].astype(np.int64)
    beg = g.level_ptr[p0].astype(np.int64)
    end = g.level_ptr[p0 + 1].astype(np.int64)
    deg = end - beg
    level_ptr = np.zeros(2 * n, np.uint64)
    cs = np.cumsum(deg)
    level_ptr[0::2] = cs - deg
    level_ptr[1::2] = cs
    total = int(cs[-1]) if n else 0
    # position of every level-0 entry inside g.neighbors: beg[i] + (k - start[i]) for k in [start[i], start[i] + deg[i])
    idx = np.repeat(beg - (cs - deg), deg) + np.arange(total, dtype=np.int64)
    neighbors = g.neighbors[idx].astype(np.int32)
    medoid = -1
    if n:
        xt = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
        mean = torch.zeros(xt.shape[1], dtype=torch.float64, device=xt.device)
        for b0 in range(0, n, 1 << 20):
            mean += xt[b0 : b0 + (1 << 20)].double().sum(0)
        mean = (mean / n).float()
        best, medoid = float("inf"), 0
        for b0 in range(0, n, 1 << 20):
            d2 = ((xt[b0 : b0 + (1 << 20)].float() - mean) ** 2).sum(1)
            v, i = torch.min(d2, 0)
            if float(v) < best:
                best, medoid = float(v), b0 + int(i)
    return HnswCsr(d=g.d, ntotal=n, metric_type=g.metric_type, levels=np.ones(n, np.int32), level_ptr=level_ptr,
                   node_offsets=np.arange(n + 1, dtype=np.uint64) * 2, neighbors=neighbors, entry_point=medoid,
                   max_level=0 if n else -1, ef_construction=g.ef_construction,
                   cum_nneighbor_per_level=g.cum_nneighbor_per_level)
