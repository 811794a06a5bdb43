"""Bulk HNSW-format graph construction on the GPU (index build time; not the query hot path).

The reference inserts points one at a time with faiss (hnsw_backend.py:83-90: M=32,
efConstruction=200) -- minutes to hours at 1M-60M chunks on CPU.  On an MI355X the whole
embedding table fits in HBM, so the graph is built in bulk instead:
  * per level, EXACT k-nearest-neighbour candidates by tiled matmul + top-k (MFMA GEMMs via torch);
  * the HNSW neighbour-selection heuristic (Malkov & Yashunin Alg. 4: keep a candidate only if it
    is closer to the base point than to every neighbour already kept) vectorised over nodes;
  * reverse links, capped at 2M (level 0) / M (upper levels) by similarity;
  * levels drawn from the HNSW geometric distribution, entry point = a top-level node.
The output is the same compact-CSR structure (convert_to_csr.py:494-548) the search path reads.
Works on any torch device (CPU for the small tests).
"""

from __future__ import annotations

import math

import numpy as np
import torch

from .csr_format import METRIC_INNER_PRODUCT, METRIC_L2, HnswCsr


@torch.no_grad()
def _knn(x: torch.Tensor, sub: torch.Tensor, k: int, metric: int, row_block: int, col_block: int):
    """Exact kNN among rows ``sub`` of x (similarity = ip or -l2).  Returns (ids [n,k] local indices, sim [n,k])."""
    xs = x[sub]
    n = xs.shape[0]
    k = min(k, n - 1)
    cd = torch.float16 if x.is_cuda else torch.float32
    xh = xs.to(cd)
    sq = (xs.float() ** 2).sum(1) if metric == METRIC_L2 else None
    out_i = torch.empty((n, k), dtype=torch.int64, device=x.device)
    out_s = torch.empty((n, k), dtype=torch.float32, device=x.device)
    for r0 in range(0, n, row_block):
        r1 = min(n, r0 + row_block)
        best_s = torch.full((r1 - r0, k), -float("inf"), device=x.device)
        best_i = torch.zeros((r1 - r0, k), dtype=torch.int64, device=x.device)
        for c0 in range(0, n, col_block):
            c1 = min(n, c0 + col_block)
            s = (xh[r0:r1] @ xh[c0:c1].T).float()
            if metric == METRIC_L2:
                s = 2 * s - sq[r0:r1, None] - sq[None, c0:c1]
            # mask self
            lo, hi = max(r0, c0), min(r1, c1)
            if lo < hi:
                idx = torch.arange(lo, hi, device=x.device)
                s[idx - r0, idx - c0] = -float("inf")
            kk = min(k, c1 - c0)
            ts, ti = torch.topk(s, kk, dim=1)
            cs = torch.cat([best_s, ts], 1)
            ci = torch.cat([best_i, ti + c0], 1)
            best_s, sel = torch.topk(cs, k, dim=1)
            best_i = torch.gather(ci, 1, sel)
        out_i[r0:r1] = best_i
        out_s[r0:r1] = best_s
    return out_i, out_s


@torch.no_grad()
def _select_heuristic(xs: torch.Tensor, cand: torch.Tensor, sim: torch.Tensor, m: int, metric: int, block: int):
    """Vectorised HNSW select-neighbours heuristic.  cand/sim: [n,K] sorted best first.
    Returns keep mask [n,K] with at most m True per row."""
    n, K = cand.shape
    keep = torch.zeros((n, K), dtype=torch.bool, device=xs.device)
    cd = torch.float16 if xs.is_cuda else torch.float32
    for b0 in range(0, n, block):
        b1 = min(n, b0 + block)
        cv = xs[cand[b0:b1]].to(cd)  # [b,K,D]
        cc = torch.bmm(cv, cv.transpose(1, 2)).float()  # ip among candidates
        if metric == METRIC_L2:
            sq = (cv.float() ** 2).sum(-1)
            cc = 2 * cc - sq[:, :, None] - sq[:, None, :]
        kb = torch.zeros((b1 - b0, K), dtype=torch.bool, device=xs.device)
        cnt = torch.zeros((b1 - b0,), dtype=torch.int32, device=xs.device)
        sb = sim[b0:b1]
        valid = torch.isfinite(sb)
        for j in range(K):
            # candidate j is dropped if some kept s is at least as close to it as the base point is
            conflict = ((cc[:, j, :] >= sb[:, j : j + 1]) & kb).any(1)
            ok = (~conflict) & (cnt < m) & valid[:, j]
            kb[:, j] = ok
            cnt += ok.int()
        keep[b0:b1] = kb
    return keep


@torch.no_grad()
def _level_graph(x: torch.Tensor, sub: torch.Tensor, m_out: int, cap: int, k_cand: int, metric: int,
                 row_block: int, col_block: int):
    """Adjacency among ``sub`` (global ids): returns (src_sorted global ids, dst global ids, counts per sub node)."""
    n = sub.shape[0]
    if n <= 1:
        return torch.zeros(0, dtype=torch.int64, device=x.device), torch.zeros(n, dtype=torch.int64, device=x.device)
    ci, cs = _knn(x, sub, k_cand, metric, row_block, col_block)
    keep = _select_heuristic(x[sub], ci, cs, m_out, metric, block=max(256, row_block // 4))
    src = torch.arange(n, device=x.device)[:, None].expand_as(ci)[keep]
    dst = ci[keep]
    w = cs[keep]
    # add reverse links, dedupe, keep the `cap` most similar per source
    s2 = torch.cat([src, dst])
    d2 = torch.cat([dst, src])
    w2 = torch.cat([w, w])
    key = s2 * n + d2
    key, perm = torch.sort(key, stable=True)
    first = torch.ones_like(key, dtype=torch.bool)
    first[1:] = key[1:] != key[:-1]
    s2, d2, w2 = s2[perm][first], d2[perm][first], w2[perm][first]
    # order by (src, -w): sort by w descending first, then stable sort by src
    o1 = torch.argsort(w2, descending=True, stable=True)
    s2, d2 = s2[o1], d2[o1]
    o2 = torch.argsort(s2, stable=True)
    s2, d2 = s2[o2], d2[o2]
    counts = torch.bincount(s2, minlength=n)
    starts = torch.cumsum(counts, 0) - counts
    rank = torch.arange(s2.shape[0], device=x.device) - starts[s2]
    ok = rank < cap
    s2, d2 = s2[ok], d2[ok]
    counts = torch.bincount(s2, minlength=n)
    return sub[d2], counts


@torch.no_grad()
def build_graph_gpu(x: torch.Tensor, metric: str = "mips", M: int = 32, k_cand: int = 96, seed: int = 12345,
                    row_block: int = 8192, col_block: int = 131072) -> HnswCsr:
    """x: [N, D] float tensor (any device).  Returns the compact-CSR graph on the host."""
    metric = metric.lower()
    if metric not in ("mips", "cosine", "l2"):
        raise ValueError(f"Unsupported distance_metric '{metric}'.")
    mt = METRIC_L2 if metric == "l2" else METRIC_INNER_PRODUCT
    n, d = x.shape
    dev = x.device
    g = torch.Generator(device="cpu").manual_seed(seed)
    u = torch.rand(n, generator=g, dtype=torch.float64).clamp_min(1e-300)
    lv = (-(u.log()) / math.log(M)).floor().clamp(max=30).to(torch.int64)  # top level of each node
    if n > 0 and int(lv.max()) > 0:
        # keep the hierarchy meaningful for tiny inputs: at least one node on every level below the top
        pass
    levels = (lv + 1).to(torch.int32)
    max_level = int(lv.max()) if n else -1
    lv_dev = lv.to(dev)
    per_level = []
    for l in range(max_level + 1):
        sub = torch.nonzero(lv_dev >= l, as_tuple=False).flatten()
        cap = 2 * M if l == 0 else M
        mo = M if l == 0 else max(M // 2, 2)
        dst, counts = _level_graph(x, sub, mo, cap, k_cand if l == 0 else min(k_cand, 64), mt, row_block, col_block)
        per_level.append((sub.cpu().numpy(), dst.cpu().numpy().astype(np.int32), counts.cpu().numpy().astype(np.int64)))
    # assemble CSR: node-major, level-minor (convert_to_csr.py:507-548)
    levels_np = levels.numpy()
    node_offsets = np.zeros(n + 1, np.uint64)
    node_offsets[1:] = np.cumsum(levels_np.astype(np.int64) + 1)
    nptr = int(node_offsets[-1]) if n else 0
    deg = np.zeros(nptr, np.int64)  # degree at pointer slot (closing slots stay 0)
    for l, (sub, dst, counts) in enumerate(per_level):
        deg[node_offsets[sub].astype(np.int64) + l] = counts
    level_ptr = np.zeros(nptr, np.uint64)
    if nptr:
        level_ptr[1:] = np.cumsum(deg)[:-1]
    total = int(deg.sum())
    neighbors = np.empty(total, np.int32)
    for l, (sub, dst, counts) in enumerate(per_level):
        if dst.shape[0] == 0:
            continue
        begin = level_ptr[node_offsets[sub].astype(np.int64) + l].astype(np.int64)
        # dst is grouped by source in `sub` order
        src_rep = np.repeat(np.arange(sub.shape[0]), counts)
        within = np.arange(dst.shape[0]) - np.repeat(np.cumsum(counts) - counts, counts)
        neighbors[begin[src_rep] + within] = dst
    top = np.nonzero(levels_np == max_level + 1)[0]
    entry = int(top[0]) if n else -1
    cum = np.array([0, 2 * M] + [2 * M + M * (i + 1) for i in range(max(max_level, 0) + 1)], dtype=np.int32)
    return HnswCsr(d=d, ntotal=n, metric_type=mt, levels=levels_np, level_ptr=level_ptr, node_offsets=node_offsets,
                   neighbors=neighbors, entry_point=entry, max_level=max_level, ef_construction=k_cand,
                   cum_nneighbor_per_level=cum)
