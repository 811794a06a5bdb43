"""Bulk HNSW graph construction driven by the GPU search kernel (index build time).

The reference inserts points one at a time with faiss (hnsw_backend.py:83-90: M=32,
efConstruction=200) -- far too slow for 1M-60M chunks without a tuned CPU library.  Here the same
insertion algorithm (Malkov & Yashunin Alg. 1: search the current graph with ef=efConstruction,
pick <= M diverse neighbours with the Alg. 4 heuristic, add reverse links, re-prune overflowing
lists) is run in BATCHES: every batch of new points searches the graph built so far with
``lm_index_search_device`` (stored-embedding mode: the embedding table is HBM resident at build
time), and link selection / reverse-link pruning are vectorised torch ops.  Levels are built top
down; the level-l graph is seeded with the finished level-(l+1) graph (its nodes exist on level l
too), which hands the lower level its long-range links -- the role early insertions play in
sequential HNSW.  Output: the compact-CSR arrays of convert_to_csr.py:494-548.

``search_fn`` is injectable so that the CPU tests can drive the builder with the oracle; the
default is the HIP path (no CPU fallback).
"""

from __future__ import annotations

import math
from typing import Callable, Optional

import numpy as np
import torch

from .csr_format import METRIC_INNER_PRODUCT, METRIC_L2, HnswCsr

SearchFn = Callable[[HnswCsr, torch.Tensor, torch.Tensor, int, int], "tuple[torch.Tensor, torch.Tensor]"]


# ---------------------------------------------------------------------------------------------
# similarity helpers (similarity = ip, or -squared-l2: larger is closer)
# ---------------------------------------------------------------------------------------------
def _cdtype(x: torch.Tensor):
    return torch.float16 if x.is_cuda else torch.float32


@torch.no_grad()
def _bruteforce_knn(xs: torch.Tensor, k: int, metric: int):
    n = xs.shape[0]
    k = min(k, n - 1)
    if k <= 0:
        return torch.zeros((n, 0), dtype=torch.int64, device=xs.device), torch.zeros((n, 0), device=xs.device)
    s = xs.float() @ xs.float().T
    if metric == METRIC_L2:
        sq = (xs.float() ** 2).sum(1)
        s = 2 * s - sq[:, None] - sq[None, :]
    s.fill_diagonal_(-float("inf"))
    ts, ti = torch.topk(s, k, dim=1)
    return ti, ts


def _domination_threshold(sim: torch.Tensor, metric: int, alpha: float) -> torch.Tensor:
    """thr[t]: a kept neighbour i rules candidate t out when cc[t, i] >= thr[t].  alpha = 1 is the HNSW rule (i at least as close to t as
    the base node is: thr = sim, bit for bit the historic behaviour).  alpha > 1 is Vamana's relaxation (DiskANN, Subramanya et al. 2019,
    RobustPrune: alpha * d(i, t) <= d(t, base)), which keeps more and longer edges: on squared L2 (similarity = -d^2)
    thr = sim / alpha^2; on inner products of UNIT vectors (d^2 = 2 - 2 ip) thr = 1 - (1 - sim) / alpha^2."""
    if alpha == 1.0:
        return sim
    a2 = float(alpha) * float(alpha)
    return sim / a2 if metric == METRIC_L2 else 1.0 - (1.0 - sim) / a2


@torch.no_grad()
def _select_heuristic_scan(xs: torch.Tensor, cand: torch.Tensor, sim: torch.Tensor, m: int, metric: int, block: int = 2048, alpha: float = 1.0):
    """The heuristic as a scan over the K candidates (the form every graph measured up to GPU session r3-15 was built with: ~10 small
    launches per candidate).  Kept as the reference the selection form below is tested against, and as its stand-in should a torch build
    reject one of the selection form's indexing ops."""
    n, K = cand.shape
    keep = torch.zeros((n, K), dtype=torch.bool, device=xs.device)
    cd = _cdtype(xs)
    for b0 in range(0, n, block):
        b1 = min(n, b0 + block)
        cb = cand[b0:b1]
        valid = cb >= 0
        cv = xs[cb.clamp(min=0)].to(cd)
        cc = torch.bmm(cv, cv.transpose(1, 2)).float()
        if metric == METRIC_L2:
            sq = (cv.float() ** 2).sum(-1)
            cc = 2 * cc - sq[:, :, None] - sq[:, None, :]
        kb = torch.zeros_like(valid)
        cnt = torch.zeros((b1 - b0,), dtype=torch.int32, device=xs.device)
        sb = sim[b0:b1]
        for j in range(K):
            conflict = ((cc[:, j, :] >= sb[:, j : j + 1]) & kb).any(1)
            ok = (~conflict) & (cnt < m) & valid[:, j]
            kb[:, j] = ok
            cnt += ok.int()
        if alpha != 1.0:  # second pass (DiskANN occlude_list: alpha 1, then the relaxed rule over what is left, while slots remain)
            sa = _domination_threshold(sb, metric, alpha)
            for j in range(K):
                conflict = ((cc[:, j, :] >= sa[:, j : j + 1]) & kb).any(1)
                ok = (~conflict) & (cnt < m) & valid[:, j] & ~kb[:, j]
                kb[:, j] |= ok
                cnt += ok.int()
        keep[b0:b1] = kb
    return keep


def _select_heuristic(xs: torch.Tensor, cand: torch.Tensor, sim: torch.Tensor, m: int, metric: int, block: int = 2048, alpha: float = 1.0):
    try:
        return _select_heuristic_selection(xs, cand, sim, m, metric, block, alpha)
    except (RuntimeError, NotImplementedError, IndexError) as ex:  # build-time torch code: same result through the scan form
        import logging

        logging.getLogger(__name__).warning(f"select heuristic: selection form failed ({type(ex).__name__}: {ex}); using the candidate scan")
        return _select_heuristic_scan(xs, cand, sim, m, metric, block, alpha)


@torch.no_grad()
def _select_heuristic_selection(xs: torch.Tensor, cand: torch.Tensor, sim: torch.Tensor, m: int, metric: int, block: int = 2048, alpha: float = 1.0):
    """HNSW select-neighbours heuristic, vectorised over rows.  cand [n,K] (-1 = empty), sim [n,K]
    sorted best first.  Returns keep mask [n,K] with <= m True per row.

    The rule (faiss shrink_neighbor_list / HNSW Alg. 4): scan the candidates best first; keep candidate j unless an already kept i is
    at least as close to j as the base node is (cc[j, i] >= sim[j]), until m are kept.  Written as a loop over SELECTIONS rather than
    over candidates: take the first candidate still alive, keep it, strike every candidate it dominates -- <= m steps of ~8 small
    launches instead of K steps of ~10 (K = 2 m at level 0), and the same keep mask bit for bit (everything before the first alive
    candidate is already decided; tests/test_host_helpers.py compares with the candidate-by-candidate scan above).  A row keeps ~10-20
    neighbours in practice, so the loop ends after that many steps (one host check per step) instead of K = 128 scan steps."""
    n, K = cand.shape
    keep = torch.zeros((n, K), dtype=torch.bool, device=xs.device)
    cd = _cdtype(xs)
    for b0 in range(0, n, block):
        b1 = min(n, b0 + block)
        cb = cand[b0:b1]
        B = b1 - b0
        cv = xs[cb.clamp(min=0)].to(cd)
        cc = torch.bmm(cv, cv.transpose(1, 2)).float()
        if metric == METRIC_L2:
            sq = (cv.float() ** 2).sum(-1)
            cc = 2 * cc - sq[:, :, None] - sq[:, None, :]
        alive = cb >= 0
        kb = torch.zeros_like(alive)
        rows = torch.arange(B, device=xs.device)
        for a in ((1.0,) if alpha == 1.0 else (1.0, alpha)):  # DiskANN occlude_list: the strict rule first, then the relaxed one over what is left
            dom = cc >= _domination_threshold(sim[b0:b1], metric, a)[:, :, None]  # dom[b, t, i]: kept i rules out candidate t
            relaxed = a != 1.0
            if relaxed:  # not kept, not ruled out (relaxed rule) by anything kept so far, rows with slots left only
                alive = (cb >= 0) & ~kb & ~(dom & kb[:, None, :]).any(2) & (kb.sum(1) < m)[:, None]
            for _ in range(min(m, K)):  # (strict pass: every step keeps one candidate per live row, so m steps bound every row by m)
                has = alive.any(1)
                if not bool(has.any()):
                    break
                first = alive.int().argmax(1)  # first alive candidate of every row (0 where none: masked by `has`)
                kb[rows, first] |= has
                struck = dom[rows, :, first]  # [B, K]: candidates the new neighbour dominates (itself included or cleared below)
                alive &= ~(struck & has[:, None])
                alive[rows, first] = False
                if relaxed:  # rows enter this pass with different counts
                    alive &= (kb.sum(1) < m)[:, None]
        keep[b0:b1] = kb
    return keep


class _LevelGraph:
    """Fixed-capacity adjacency of one level over the node subset ``sub`` (sorted global ids)."""

    def __init__(self, sub: torch.Tensor, cap: int):
        n = sub.shape[0]
        self.sub = sub
        self.cap = cap
        self.adj = torch.full((n, cap), -1, dtype=torch.int64, device=sub.device)
        self.sim = torch.full((n, cap), -float("inf"), dtype=torch.float32, device=sub.device)
        self.deg = torch.zeros((n,), dtype=torch.int64, device=sub.device)
        self.alpha = 1.0  # neighbour-selection relaxation (_domination_threshold); build_graph_gpu(alpha=...) sets it

    @torch.no_grad()
    def add_links(self, xs: torch.Tensor, src: torch.Tensor, dst: torch.Tensor, w: torch.Tensor, metric: int):
        """Add directed edges src->dst (weights w = similarity); lists that overflow ``cap`` are
        re-pruned with the heuristic (faiss shrink_neighbor_list)."""
        if src.numel() == 0:
            return
        n, cap = self.adj.shape
        aff = torch.unique(src)
        # existing edges of the affected rows
        ea = self.adj[aff]
        em = ea >= 0
        es = aff[:, None].expand_as(ea)[em]
        s_all = torch.cat([es, src])
        d_all = torch.cat([ea[em], dst])
        w_all = torch.cat([self.sim[aff][em], w])
        # dedupe (src,dst)
        key = s_all * n + d_all
        key, perm = torch.sort(key, stable=True)
        first = torch.ones_like(key, dtype=torch.bool)
        first[1:] = key[1:] != key[:-1]
        s_all, d_all, w_all = s_all[perm][first], d_all[perm][first], w_all[perm][first]
        # order by (src, -w)
        o1 = torch.argsort(w_all, descending=True, stable=True)
        s_all, d_all, w_all = s_all[o1], d_all[o1], w_all[o1]
        o2 = torch.argsort(s_all, stable=True)
        s_all, d_all, w_all = s_all[o2], d_all[o2], w_all[o2]
        cnt = torch.bincount(s_all, minlength=n)
        start = torch.cumsum(cnt, 0) - cnt
        rank = torch.arange(s_all.shape[0], device=s_all.device) - start[s_all]
        Kc = 2 * cap
        okc = rank < Kc
        s_all, d_all, w_all, rank = s_all[okc], d_all[okc], w_all[okc], rank[okc]
        # dense candidate rows for the affected nodes
        row_of = torch.full((n,), -1, dtype=torch.int64, device=s_all.device)
        row_of[aff] = torch.arange(aff.shape[0], device=s_all.device)
        cand = torch.full((aff.shape[0], Kc), -1, dtype=torch.int64, device=s_all.device)
        csim = torch.full((aff.shape[0], Kc), -float("inf"), dtype=torch.float32, device=s_all.device)
        cand[row_of[s_all], rank] = d_all
        csim[row_of[s_all], rank] = w_all
        ccount = (cand >= 0).sum(1)
        over = ccount > cap
        new_adj = cand[:, :cap].clone()
        new_sim = csim[:, :cap].clone()
        if bool(over.any()):
            oi = torch.nonzero(over).flatten()
            keep = _select_heuristic(xs, cand[oi], csim[oi], cap, metric, alpha=self.alpha)
            # compact kept entries to the front (stable)
            order = torch.argsort((~keep).int(), dim=1, stable=True)
            kc = torch.gather(cand[oi], 1, order)[:, :cap]
            ks = torch.gather(csim[oi], 1, order)[:, :cap]
            kk = torch.gather(keep, 1, order)[:, :cap]
            kc[~kk] = -1
            ks[~kk] = -float("inf")
            new_adj[oi] = kc
            new_sim[oi] = ks
        self.adj[aff] = new_adj
        self.sim[aff] = new_sim
        self.deg[aff] = (new_adj >= 0).sum(1)


def _assemble_csr(levels_top: np.ndarray, graphs: "list[tuple[np.ndarray, np.ndarray]]", base_sub: np.ndarray,
                  d: int, mt: int, entry_global: int, M: int, efc: int) -> HnswCsr:
    """CSR over the nodes ``base_sub`` (sorted global ids) from per-level adjacencies.
    graphs[j] = (sub_global_ids, adj [n_j, cap] LOCAL to that sub, -1 padded) for level base+j.
    levels_top: top level (relative to base) of every node of base_sub."""
    n = base_sub.shape[0]
    nlev = levels_top.astype(np.int64) + 1
    node_offsets = np.zeros(n + 1, np.uint64)
    node_offsets[1:] = np.cumsum(nlev + 1)
    nptr = int(node_offsets[-1])
    deg = np.zeros(nptr, np.int64)
    per = []
    for j, (sub, adj) in enumerate(graphs):
        loc = np.searchsorted(base_sub, sub)  # sub-local -> base-local
        m = adj >= 0
        dj = m.sum(1)
        deg[node_offsets[loc].astype(np.int64) + j] = dj
        per.append((loc, adj, m, dj))
    level_ptr = np.zeros(nptr, np.uint64)
    if nptr:
        level_ptr[1:] = np.cumsum(deg)[:-1]
    neighbors = np.empty(int(deg.sum()), np.int32)
    for j, (loc, adj, m, dj) in enumerate(per):
        if not m.any():
            continue
        begin = level_ptr[node_offsets[loc].astype(np.int64) + j].astype(np.int64)
        # entries are left-packed per row (add_links keeps them compact); positions by cumulative count
        pos = np.cumsum(m, axis=1) - 1
        rows = np.nonzero(m)
        neighbors[begin[rows[0]] + pos[rows]] = loc[adj[rows]]
    max_level = int(levels_top.max()) if n else -1
    entry = int(np.searchsorted(base_sub, entry_global)) if n else -1
    cum = np.array([0, 2 * M] + [2 * M + M * (i + 1) for i in range(max(max_level, 0) + 1)], dtype=np.int32)
    return HnswCsr(d=d, ntotal=n, metric_type=mt, levels=nlev.astype(np.int32), level_ptr=level_ptr,
                   node_offsets=node_offsets, neighbors=neighbors, entry_point=entry, max_level=max_level,
                   ef_construction=efc, cum_nneighbor_per_level=cum)


def hip_search_fn(device_index: int = 0, beam: int = 2) -> SearchFn:
    """Default candidate search: the HIP stored-embedding search (lm_index_search_device)."""

    def fn(g: HnswCsr, table: torch.Tensor, queries: torch.Tensor, ef: int, k: int):
        from .index import Mi355xIndex

        idx = Mi355xIndex.from_csr(g, device=device_index)
        try:
            dp = idx.info.d_padded
            if table.shape[1] != dp:
                t = torch.zeros((table.shape[0], dp), dtype=torch.float32, device=table.device)
                t[:, : table.shape[1]] = table
                table = t
            idx.set_stream(torch.cuda.current_stream().cuda_stream)
            idx.attach_table(table.float().contiguous())
            prm = idx.make_params(ef=ef, beam=beam, recompute=False, max_batch=16384)
            dist, ids = idx.search_device(queries.float().contiguous(), k, prm)
            torch.cuda.synchronize()
            sim = dist if g.metric_type == METRIC_INNER_PRODUCT else -dist
            return ids, sim
        finally:
            idx.close()

    return fn


@torch.no_grad()
def build_graph_gpu(x: torch.Tensor, metric: str = "mips", M: int = 32, ef_construction: int = 200, seed: int = 12345,
                    search_fn: Optional[SearchFn] = None, growth: float = 1.5, k_cand: int = 0,
                    seed_nodes: int = 2048, refine: bool = True, verbose: bool = False, alpha: float = 1.0) -> HnswCsr:
    """x: [N, D] float tensor on the build device.  Returns the compact-CSR HNSW graph (host).  ``alpha`` > 1 relaxes the neighbour
    selection the way Vamana does (denser lists with longer edges: what a PQ-guided walk over a flat graph needs at 10M nodes -- DESIGN 8;
    inner-product metrics then assume unit vectors); 1.0 = the HNSW rule, the graphs every measurement so far was taken on."""
    metric = metric.lower()
    if metric not in ("mips", "cosine", "l2"):
        raise ValueError(f"Unsupported distance_metric '{metric}'.")
    mt = METRIC_L2 if metric == "l2" else METRIC_INNER_PRODUCT
    n, d = x.shape
    dev = x.device
    if n == 0:
        return HnswCsr(d=d, ntotal=0, metric_type=mt, levels=np.zeros(0, np.int32), level_ptr=np.zeros(0, np.uint64),
                       node_offsets=np.zeros(1, np.uint64), neighbors=np.zeros(0, np.int32), entry_point=-1, max_level=-1)
    if search_fn is None:
        search_fn = hip_search_fn(dev.index or 0)
    k_cand = k_cand or min(ef_construction, 128)
    gen = torch.Generator(device="cpu").manual_seed(seed)
    u = torch.rand(n, generator=gen, dtype=torch.float64).clamp_min(1e-300)
    lv = (-(u.log()) / math.log(M)).floor().clamp(max=30).to(torch.int64)
    max_level = int(lv.max())
    lv_np = lv.numpy()
    entry_global = int(np.nonzero(lv_np == max_level)[0][0])
    lv_dev = lv.to(dev)
    finished: "list[_LevelGraph]" = []  # finished[0] is the level just above the one being built

    for l in range(max_level, -1, -1):
        sub = torch.nonzero(lv_dev >= l).flatten()
        nl = sub.shape[0]
        xs = x[sub]
        cap = 2 * M if l == 0 else M
        G = _LevelGraph(sub, cap)
        G.alpha = alpha
        inserted = torch.zeros(nl, dtype=torch.bool, device=dev)
        if finished and finished[0].sub.shape[0] >= 2:
            # seed with the level above (its nodes are a subset of this level)
            up = finished[0]
            loc = torch.searchsorted(sub, up.sub)
            m = up.adj >= 0
            a = torch.full((up.adj.shape[0], cap), -1, dtype=torch.int64, device=dev)
            s = torch.full((up.adj.shape[0], cap), -float("inf"), device=dev)
            a[:, : up.cap][m] = loc[up.adj[m]]
            s[:, : up.cap][m] = up.sim[m]
            G.adj[loc], G.sim[loc] = a, s
            G.deg[loc] = m.sum(1)
            inserted[loc] = True
        else:
            ns = min(nl, seed_nodes)
            perm0 = torch.randperm(nl, generator=gen)[:ns].to(dev)
            if finished:  # keep the upper node(s) inside the seed set
                perm0 = torch.unique(torch.cat([torch.searchsorted(sub, finished[0].sub), perm0]))
            ci, cs = _bruteforce_knn(xs[perm0], min(k_cand, perm0.shape[0] - 1), mt)
            if ci.shape[1] > 0:
                keep = _select_heuristic(xs[perm0], ci, cs, M, mt, alpha=alpha)
                src = perm0[torch.arange(perm0.shape[0], device=dev)[:, None].expand_as(ci)[keep]]
                dst = perm0[ci[keep]]
                w = cs[keep]
                G.add_links(xs, torch.cat([src, dst]), torch.cat([dst, src]), torch.cat([w, w]), mt)
            inserted[perm0] = True
        rest = torch.nonzero(~inserted).flatten()
        rest = rest[torch.randperm(rest.shape[0], generator=gen).to(dev)]
        # levels (relative to l) of the nodes of this subset, for the temporary search structure
        rel_top = (lv_np[sub.cpu().numpy()] - l).astype(np.int64)
        sub_np = sub.cpu().numpy()

        def temp_csr():
            graphs = [(sub_np, G.adj.cpu().numpy())] + [(f.sub.cpu().numpy(), f.adj.cpu().numpy()) for f in finished]
            return _assemble_csr(rel_top, graphs, sub_np, d, mt, entry_global, M, ef_construction)

        def insert(batch: torch.Tensor, replace: bool):
            g = temp_csr()
            kk = min(k_cand + (1 if replace else 0), max(int(inserted.sum()) - 1, 1))
            ids, sim = search_fn(g, xs, xs[batch], max(ef_construction, kk), kk)
            ids = ids.to(dev)
            sim = sim.to(dev).float()
            if replace:  # refinement: drop self, merge with current links
                selfm = ids == batch[:, None]
                ids = ids.masked_fill(selfm, -1)
                sim = sim.masked_fill(selfm, -float("inf"))
                ids = torch.cat([ids, G.adj[batch]], 1)
                sim = torch.cat([sim, G.sim[batch]], 1)
                # dedupe within rows: sort by id, blank repeated
                so = torch.argsort(ids, dim=1, stable=True)
                ids_s = torch.gather(ids, 1, so)
                dup = torch.zeros_like(ids_s, dtype=torch.bool)
                dup[:, 1:] = (ids_s[:, 1:] == ids_s[:, :-1]) & (ids_s[:, 1:] >= 0)
                dupo = torch.zeros_like(dup)
                dupo.scatter_(1, so, dup)
                ids = ids.masked_fill(dupo, -1)
                sim = sim.masked_fill(dupo, -float("inf"))
                o = torch.argsort(sim, dim=1, descending=True, stable=True)
                ids, sim = torch.gather(ids, 1, o), torch.gather(sim, 1, o)
            ids = ids.masked_fill(~torch.isfinite(sim), -1)
            keep = _select_heuristic(xs, ids, sim, M, mt, alpha=alpha)
            src = batch[:, None].expand_as(ids)[keep]
            dst = ids[keep]
            w = sim[keep]
            if replace:
                G.adj[batch] = -1
                G.sim[batch] = -float("inf")
                G.deg[batch] = 0
            G.add_links(xs, torch.cat([src, dst]), torch.cat([dst, src]), torch.cat([w, w]), mt)
            inserted[batch] = True

        pos = 0
        while pos < rest.shape[0]:
            have = int(inserted.sum())
            bs = max(256, int(have * (growth - 1.0)))
            batch = rest[pos : pos + bs]
            insert(batch, replace=False)
            pos += batch.shape[0]
            if verbose:
                print(f"[build] level {l}: {int(inserted.sum())}/{nl} nodes, mean degree {float(G.deg[inserted].float().mean()):.1f}")
        if refine and nl > seed_nodes:
            allb = torch.arange(nl, device=dev)
            step = max(4096, nl // 8)
            for b0 in range(0, nl, step):
                insert(allb[b0 : b0 + step], replace=True)
            if verbose:
                print(f"[build] level {l}: refined, mean degree {float(G.deg.float().mean()):.1f}")
        finished.insert(0, G)

    # final assembly over all nodes (level 0 subset == everything)
    graphs = [(f.sub.cpu().numpy(), f.adj.cpu().numpy()) for f in finished]
    all_ids = np.arange(n, dtype=np.int64)
    return _assemble_csr(lv_np, graphs, all_ids, d, mt, entry_global, M, ef_construction)


@torch.no_grad()
def prune_preserving_hubs(g: HnswCsr, x: torch.Tensor, M: int, m_low: int, hub_fraction: float = 0.02) -> HnswCsr:
    """High-degree-preserving pruning of the level-0 graph (LEANN paper, Algorithm 3, p. 6): storage drops from ~2M links per node
    to ~m_low while the few hub nodes that most searches pass through keep their full lists.

      * V* = the ``hub_fraction`` (paper: 2 %) nodes of highest degree -- in-degree here: how many lists a node appears in;
      * every node re-selects its out-links from its candidate list W(v) in the original (heuristic) order: up to 2M (the
        level-0 cap of the unpruned graph) for v in V*, up to ``m_low`` < M otherwise.  W(v) is v's list in the input graph:
        the builder produced it with the select-neighbours heuristic from an ef_construction-sized candidate set, so this
        is the paper's re-selection without repeating the N searches;
      * for every kept link v -> u the reverse link u -> v is offered as well and every node may hold up to 2M links in
        total, overflowing lists being shrunk with the same heuristic (paper: "all nodes establish bidirectional edges up
        to the maximum threshold M; only the number of outgoing selections of low-degree nodes is restricted").
    Upper levels are untouched (they hold ~N/M nodes).  Returns a new graph; ``x`` = the [N, D] embeddings on any device."""
    n = g.ntotal
    if n == 0 or m_low >= 2 * M:
        return g
    dev = x.device
    mt = g.metric_type
    p0 = g.node_offsets[:-1].astype(np.int64)
    beg = g.level_ptr[p0].astype(np.int64)
    deg = (g.level_ptr[p0 + 1].astype(np.int64) - beg)
    cap = 2 * M
    # dense level-0 adjacency [n, cap] in stored order (-1 padded)
    adj = np.full((n, cap), -1, np.int64)
    col = np.arange(cap)[None, :]
    m = col < np.minimum(deg, cap)[:, None]
    adj[m] = g.neighbors[(beg[:, None] + col)[m]]
    indeg = np.bincount(adj[m], minlength=n)  # in how many level-0 lists a node appears
    nh = max(1, int(round(hub_fraction * n)))
    hubs = np.argpartition(-indeg, min(nh, n - 1))[:nh]
    quota = np.full(n, m_low, np.int64)
    quota[hubs] = cap
    keep = (col < quota[:, None]) & (adj >= 0)
    src = torch.from_numpy(np.broadcast_to(np.arange(n)[:, None], adj.shape)[keep].copy()).to(dev)
    dst = torch.from_numpy(adj[keep]).to(dev)
    # similarities of the kept links (larger = closer), in blocks
    w = torch.empty(src.shape[0], dtype=torch.float32, device=dev)
    for b0 in range(0, src.shape[0], 1 << 20):
        a, b = x[src[b0 : b0 + (1 << 20)]].float(), x[dst[b0 : b0 + (1 << 20)]].float()
        w[b0 : b0 + (1 << 20)] = (a * b).sum(1) if mt == METRIC_INNER_PRODUCT else -((a - b) ** 2).sum(1)
    G = _LevelGraph(torch.arange(n, device=dev), cap)
    step = 1 << 21  # bound the temporaries of add_links
    for b0 in range(0, src.shape[0], step):
        s_, d_, w_ = src[b0 : b0 + step], dst[b0 : b0 + step], w[b0 : b0 + step]
        G.add_links(x, torch.cat([s_, d_]), torch.cat([d_, s_]), torch.cat([w_, w_]), mt)
    new0 = G.adj.cpu().numpy()
    # reassemble: level 0 replaced, upper levels copied
    nlev = g.levels.astype(np.int64)
    nptr = int(g.node_offsets[-1])
    old_deg = np.diff(np.concatenate([g.level_ptr.astype(np.int64), [g.neighbors.shape[0]]]))[:nptr] if nptr else np.zeros(0, np.int64)
    new_deg = old_deg.copy()
    m0 = new0 >= 0
    new_deg[p0] = m0.sum(1)
    # the sentinel slot (last pointer of every node) holds no list
    new_deg[(g.node_offsets[1:].astype(np.int64) - 1)] = 0
    level_ptr = np.zeros(nptr, np.uint64)
    if nptr:
        level_ptr[1:] = np.cumsum(new_deg)[:-1]
    neighbors = np.empty(int(new_deg.sum()), np.int32)
    # level 0
    pos = np.cumsum(m0, axis=1) - 1
    rows = np.nonzero(m0)
    neighbors[level_ptr[p0].astype(np.int64)[rows[0]] + pos[rows]] = new0[rows]
    # upper levels: copy list by list (few nodes)
    up = np.nonzero(nlev > 1)[0]
    for i in up:
        for l in range(1, int(nlev[i])):
            p = int(g.node_offsets[i]) + l
            b, e = int(g.level_ptr[p]), int(g.level_ptr[p + 1])
            nb = int(level_ptr[p])
            neighbors[nb : nb + (e - b)] = g.neighbors[b:e]
    out = HnswCsr(d=g.d, ntotal=n, metric_type=mt, levels=g.levels.copy(), level_ptr=level_ptr, node_offsets=g.node_offsets.copy(),
                  neighbors=neighbors, entry_point=g.entry_point, max_level=g.max_level, ef_construction=g.ef_construction,
                  cum_nneighbor_per_level=g.cum_nneighbor_per_level)
    out.validate()
    return out
