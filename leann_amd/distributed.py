"""Multi-GPU host logic: one process per GPU, `torch.distributed` (backend "nccl" == RCCL over xGMI).

Two modes; the reference has no distributed code at all, so this is new design:

* **replicated graph, partitioned queries** (1M / 10M-chunk configs): queries are independent units,
  every rank searches its slice; no collective on the data path.  `gather_results` optionally
  concatenates the slices on every rank with one all_gather.
* **sharded graph** (60M-chunk config): each rank holds a disjoint shard (local node ids +
  ``id_base``), every rank searches ALL queries on its shard, the per-shard top-k lists
  ``(B,k) x {f32 dist, i64 id}`` are packed into one buffer and exchanged with ONE all_gather (30 KB/rank at B=256, k=10 --
  latency bound, so a single one-shot collective, not a ring pipeline) and merged per query by
  ``lm_topk_merge`` (order: internal distance, then id).

The local search and the merge are injectable (``search_fn`` / ``merge_fn``) so the world_size-2
``gloo`` tests can run the host logic on CPU with the oracle as the compute stand-in; the defaults
are the HIP paths and raise without a GPU.
"""

from __future__ import annotations

import ctypes as C
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist

from . import _lib


def partition(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced slice [lo, hi) of n items for `rank` (first n % world ranks get one more)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def hip_merge_fn(ids: torch.Tensor, dist_: torch.Tensor, metric: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """(S,B,k) device tensors -> (B,k) via the lm_topk_merge kernel."""
    _lib.require_gpu()
    lib = _lib.load()
    S, B, k = ids.shape
    ids = ids.contiguous()
    dist_ = dist_.contiguous()
    oi = torch.empty((B, k), dtype=torch.int64, device=ids.device)
    od = torch.empty((B, k), dtype=torch.float32, device=ids.device)
    _lib.check(lib.lm_topk_merge(C.c_void_p(ids.data_ptr()), C.c_void_p(dist_.data_ptr()), S, B, k, metric,
                                 C.c_void_p(oi.data_ptr()), C.c_void_p(od.data_ptr()),
                                 C.c_void_p(torch.cuda.current_stream().cuda_stream)), "lm_topk_merge")
    return oi, od


def pack_results(d: torch.Tensor, i: torch.Tensor) -> torch.Tensor:
    """(m,k) f32 distances + (m,k) i64 ids -> ONE (m, 3k) int32 buffer (ids as two words each, then the distance bits): what a rank
    contributes to the single all_gather of a step (12 bytes per result: 30 KB at B = 256, k = 10)."""
    return torch.cat([i.contiguous().view(torch.int32), d.contiguous().view(torch.int32)], dim=1)


def unpack_results(buf: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Inverse of pack_results on (..., 3k) int32 -> (dist (..., k) f32, ids (..., k) i64)."""
    ids = buf[..., : 2 * k].contiguous().view(torch.int64)
    d = buf[..., 2 * k :].contiguous().view(torch.float32)
    return d, ids


def all_gather_results(d: torch.Tensor, i: torch.Tensor, world: int, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """ONE collective for a step's results: every rank's (m,k) distances and ids -> (world, m, k) of each on every rank."""
    m, k = d.shape
    buf = pack_results(d, i)
    out = torch.empty((world * m, 3 * k), dtype=torch.int32, device=buf.device)
    dist.all_gather_into_tensor(out, buf, group=group)
    gd, gi = unpack_results(out, k)
    return gd.view(world, m, k), gi.view(world, m, k)


class PartitionedSearch:
    """Replicated index, queries partitioned across ranks."""

    def __init__(self, search_fn: Callable[[torch.Tensor, int], Tuple[torch.Tensor, torch.Tensor]],
                 group: Optional[dist.ProcessGroup] = None):
        self.search_fn = search_fn
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def search(self, queries: torch.Tensor, k: int, gather_results: bool = True):
        """queries: (B,D) identical on every rank.  Returns (dist (B,k), ids (B,k)) for all B when
        gather_results, else this rank's slice and its (lo, hi)."""
        B = queries.shape[0]
        lo, hi = partition(B, self.world, self.rank)
        d, i = self.search_fn(queries[lo:hi].contiguous(), k)
        if not gather_results or self.world == 1:
            return (d, i) if self.world == 1 else (d, i, (lo, hi))
        # equal-size all_gather: pad the slices to the largest one
        m = (B + self.world - 1) // self.world
        pd = torch.zeros((m, k), dtype=torch.float32, device=d.device)
        pi = torch.full((m, k), -1, dtype=torch.int64, device=i.device)
        pd[: hi - lo], pi[: hi - lo] = d, i
        gd, gi = all_gather_results(pd, pi, self.world, self.group)  # one packed exchange (distances + ids together)
        od = torch.cat([gd[r][: partition(B, self.world, r)[1] - partition(B, self.world, r)[0]] for r in range(self.world)])
        oi = torch.cat([gi[r][: partition(B, self.world, r)[1] - partition(B, self.world, r)[0]] for r in range(self.world)])
        return od, oi


class ShardedSearch:
    """Graph sharded across ranks: local top-k on every shard, one all_gather, per-query merge."""

    def __init__(self, search_fn: Callable[[torch.Tensor, int], Tuple[torch.Tensor, torch.Tensor]], id_base: int,
                 metric: int, merge_fn: Callable = hip_merge_fn, group: Optional[dist.ProcessGroup] = None):
        self.search_fn = search_fn
        self.id_base = int(id_base)
        self.metric = int(metric)
        self.merge_fn = merge_fn
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def search(self, queries: torch.Tensor, k: int):
        """queries: (B,D) identical on every rank -> global (dist (B,k), ids (B,k)), identical on every rank."""
        d, i = self.search_fn(queries, k)
        i = torch.where(i >= 0, i + self.id_base, i)
        if self.world == 1:
            return self.merge_fn(i[None].contiguous(), d[None].contiguous(), self.metric)[::-1]
        gd, gi = all_gather_results(d, i, self.world, self.group)  # one packed exchange: (world, B, k) x {f32, i64}
        oi, od = self.merge_fn(gi.contiguous(), gd.contiguous(), self.metric)
        return od, oi


def shard_bounds(n: int, world: int):
    """Node ranges of the shards: [(lo, hi)] * world."""
    return [partition(n, world, r) for r in range(world)]


def build_fingerprint() -> int:
    """63-bit fingerprint of what a rank runs: package version, the library's own version string, its ABI revision (lm_abi_revision) and the
    exported-symbol list.  Ranks of one job must agree (same checkout, same interface): a rank whose library differs would otherwise
    interpret broadcast bytes or packed results differently -- or hang in a collective the others never enter.  NOT the byte size of the
    shared object (round 5 hashed it): nodes that build the same checkout with different compiler paths get different sizes and the
    job would fail for nothing."""
    import hashlib

    from . import __version__

    h = hashlib.sha256(__version__.encode())
    try:
        lib = _lib.load()
        h.update(bytes(lib.lm_version() or b""))
        h.update(",".join(_lib.EXPORTED_SYMBOLS).encode())
        h.update(str(int(lib.lm_abi_revision())).encode())
    except Exception as ex:  # noqa: BLE001 - a rank without the library is a different build by definition
        h.update(("no library: " + type(ex).__name__).encode())
    return int.from_bytes(h.digest()[:8], "little") >> 1


def check_same_build(device: Optional[torch.device] = None, group: Optional[dist.ProcessGroup] = None) -> None:
    """Every rank contributes its build_fingerprint() to ONE all_gather; if they differ, EVERY rank raises the same RuntimeError (they all
    see the same list), so a mixed job fails loudly at start-up instead of hanging in its first data-path collective."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    import os

    if os.environ.get("LEANN_MI355X_SKIP_BUILD_CHECK") == "1":  # the operator's override (set it on EVERY rank: the check is a collective)
        return
    dev = device if device is not None else torch.device("cpu")
    mine = torch.tensor([build_fingerprint()], dtype=torch.int64, device=dev)
    got = torch.empty((dist.get_world_size(group),), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(got, mine, group=group)
    vals = [int(v) for v in got.cpu().tolist()]
    if len(set(vals)) != 1:
        odd = [r for r, v in enumerate(vals) if v != vals[0]]
        raise RuntimeError(f"leann-backend-mi355x: ranks run different builds of the package / libleann_mi355x.so (fingerprints differ from rank 0's on ranks {odd}); "
                           "every rank of a job must use the same checkout and the same built library")


def broadcast_graph(g, src: int = 0, device: Optional[torch.device] = None, group: Optional[dist.ProcessGroup] = None):
    """Rank `src` built the compact-CSR graph; every other rank passes ``g=None`` and receives a copy (the index is
    built ONCE per job and replicated, not rebuilt per rank).  Arrays travel in their own dtype as raw bytes on `device` (RCCL
    over xGMI when it is a GPU, gloo on the CPU tests); scalars in one header tensor."""
    from .csr_format import HnswCsr

    import numpy as np

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return g
    rank = dist.get_rank(group)
    dev = device if device is not None else torch.device("cpu")
    check_same_build(dev, group)  # (a rank with another build fails here, on every rank at once, not in the middle of the transfer)
    names = ("levels", "level_ptr", "node_offsets", "neighbors", "cum_nneighbor_per_level")
    dtypes = {"levels": np.int32, "level_ptr": np.uint64, "node_offsets": np.uint64, "neighbors": np.int32, "cum_nneighbor_per_level": np.int32}
    if rank == src:
        arrs = [np.ascontiguousarray(getattr(g, n)) for n in names]
        hdr = torch.tensor([g.d, g.ntotal, g.metric_type, g.entry_point, g.max_level, g.ef_construction] + [a.shape[0] for a in arrs],
                           dtype=torch.int64, device=dev)
    else:
        hdr = torch.zeros(6 + len(names), dtype=torch.int64, device=dev)
    dist.broadcast(hdr, src, group=group)
    h = [int(v) for v in hdr.cpu().tolist()]
    # the header every rank now holds must describe a graph (the same check on every rank: a bad header raises everywhere, nobody waits);
    # metric_type is 0 (inner product) or 1 (L2): the only two lm_index_create_from_csr accepts (faiss metrics > 1 carry a metric_arg
    # this library has no kernel for)
    if not (h[0] > 0 and h[1] > 0 and h[2] in (0, 1) and 0 <= h[3] < h[1] and h[4] >= 0 and h[6] == h[1] and h[8] == h[1] + 1 and h[7] >= h[1] and h[9] >= 0):
        raise RuntimeError(f"broadcast_graph: implausible header from rank {src}: {h}")
    out = []
    for i, n in enumerate(names):
        nbytes = h[6 + i] * np.dtype(dtypes[n]).itemsize
        if rank == src:
            t = torch.from_numpy(np.ascontiguousarray(arrs[i].astype(dtypes[n], copy=False)).view(np.uint8).copy()).to(dev)
        else:
            t = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        dist.broadcast(t, src, group=group)
        out.append(t.cpu().numpy().view(dtypes[n]).copy())
    if rank == src:
        return g
    return HnswCsr(d=h[0], ntotal=h[1], metric_type=h[2], levels=out[0], level_ptr=out[1], node_offsets=out[2], neighbors=out[3],
                   entry_point=h[3], max_level=h[4], ef_construction=h[5], cum_nneighbor_per_level=out[4])
