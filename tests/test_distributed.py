"""world_size-2 `gloo` tests of the multi-GPU host logic on CPU.  The local search /
merge are injected with the ORACLE as compute stand-in (test infrastructure); on a GPU box the same
classes run with the HIP defaults."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from leann_amd.distributed import PartitionedSearch, ShardedSearch, partition, shard_bounds
from tests.util import clustered, oracle_graph, queries_near


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_partition_covers_everything():
    for n in (0, 1, 7, 256, 1000):
        for w in (1, 2, 3, 8):
            sl = [partition(n, w, r) for r in range(w)]
            assert sl[0][0] == 0 and sl[-1][1] == n
            assert all(sl[i][1] == sl[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in sl) - min(h - l for l, h in sl) <= 1


def _oracle_merge(ids, dist_, metric):
    from oracle import oracle as orc

    oi, od = orc.merge_topk(ids.numpy(), dist_.numpy(), metric)
    return torch.from_numpy(oi), torch.from_numpy(od)


def _worker(rank, world, port, mode, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from leann_amd.hnsw_builder import build_hnsw
    from oracle import oracle as orc

    torch.set_num_threads(1)
    x = clustered(3000, 64, 5)
    q = torch.from_numpy(queries_near(x, 33, 6))
    if mode in ("partitioned", "broadcast"):
        if mode == "broadcast":  # the index is built once, on rank 0, and replicated (bench.py's N > 1 path)
            from leann_amd.distributed import broadcast_graph

            g0 = build_hnsw(x, "mips", M=8, ef_construction=40, num_threads=1) if rank == 0 else None
            g = broadcast_graph(g0, 0)
            g.validate()
            ref = build_hnsw(x, "mips", M=8, ef_construction=40, num_threads=1)
            assert all(np.array_equal(getattr(g, n), getattr(ref, n)) for n in ("levels", "level_ptr", "node_offsets", "neighbors"))
            assert (g.entry_point, g.max_level, g.ntotal, g.d, g.metric_type) == (ref.entry_point, ref.max_level, ref.ntotal, ref.d, ref.metric_type)
        else:
            g = build_hnsw(x, "mips", M=8, ef_construction=40, num_threads=1)
        og = oracle_graph(g, 64)

        def search_fn(qq, k):
            i, d, _ = orc.search(og, qq.numpy(), k, ef=32, table=x)
            return torch.from_numpy(d), torch.from_numpy(i)

        d, i = PartitionedSearch(search_fn).search(q, 5)
        ei, ed, _ = orc.search(og, q.numpy(), 5, ef=32, table=x)
        ok = np.array_equal(i.numpy(), ei) and np.array_equal(d.numpy(), ed)
    else:
        lo, hi = shard_bounds(3000, world)[rank]
        xs = x[lo:hi]
        g = build_hnsw(xs, "mips", M=8, ef_construction=40, num_threads=1)
        og = oracle_graph(g, 64)

        def search_fn(qq, k):
            i, d, _ = orc.search(og, qq.numpy(), k, ef=64, table=xs)
            return torch.from_numpy(d), torch.from_numpy(i)

        d, i = ShardedSearch(search_fn, id_base=lo, metric=0, merge_fn=_oracle_merge).search(q, 5)
        # every rank holds the same merged answer; it must be close to the exact global top-5
        gt, _ = orc.bruteforce_topk(x, q.numpy(), 5, 0)
        rec = np.mean([len(set(i[r].tolist()) & set(gt[r].tolist())) / 5 for r in range(q.shape[0])])
        ok = rec > 0.95 and bool(np.all(np.diff(d.numpy(), axis=1) <= 0)) and int(i.max()) < 3000
        gi = [torch.empty_like(i) for _ in range(world)]
        dist.all_gather(gi, i)
        ok = ok and all(torch.equal(gi[0], t) for t in gi)
    out[rank] = ok
    dist.destroy_process_group()


def _mixed_build_worker(rank, world, port, out):
    """Rank 1 pretends to run another build: broadcast_graph must raise ON EVERY RANK (nobody is left waiting in a collective)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from leann_amd import distributed as D
    from leann_amd.hnsw_builder import build_hnsw

    torch.set_num_threads(1)
    real = D.build_fingerprint()
    assert real == D.build_fingerprint() and real >= 0
    if rank == 1:
        D.build_fingerprint = lambda: real ^ 1
    g0 = build_hnsw(clustered(300, 16, 5), "mips", M=4, ef_construction=20, num_threads=1) if rank == 0 else None
    try:
        D.broadcast_graph(g0, 0)
        out[rank] = "no error"
    except RuntimeError as ex:
        out[rank] = "different builds" in str(ex) and "[1]" in str(ex)
    dist.destroy_process_group()


def test_two_rank_gloo_mixed_builds_fail_loudly(built_libs):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_mixed_build_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


@pytest.mark.parametrize("mode", ["partitioned", "broadcast", "sharded"])
def test_two_rank_gloo(mode, built_libs):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), mode, out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}


def test_pack_and_unpack_results_round_trip():
    from leann_amd.distributed import pack_results, unpack_results

    d = torch.tensor([[1.5, -2.25, float("inf")], [0.0, 3.0, -0.0]], dtype=torch.float32)
    i = torch.tensor([[7, -1, 2**40 + 3], [0, 5, -1]], dtype=torch.int64)
    buf = pack_results(d, i)
    assert buf.dtype == torch.int32 and tuple(buf.shape) == (2, 9)
    d2, i2 = unpack_results(buf, 3)
    assert torch.equal(d2.view(torch.int32), d.view(torch.int32)) and torch.equal(i2, i)


def _nccl_worker(rank, world, port, out):
    """Both multi-GPU host classes on RCCL: one rank per GPU, HIP search + HIP merge, every rank must hold the single-GPU answer."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from leann_amd.distributed import broadcast_graph
    from leann_amd.hnsw_builder import build_hnsw
    from leann_amd.index import Mi355xIndex
    from oracle import oracle as orc

    x = clustered(3000, 64, 5)
    qn = queries_near(x, 33, 6)
    q = torch.from_numpy(qn).to(dev)
    g = broadcast_graph(build_hnsw(x, "mips", M=8, ef_construction=40, num_threads=1) if rank == 0 else None, 0, device=dev)
    idx = Mi355xIndex.from_csr(g, device=rank)
    idx.attach_table(x)
    prm = idx.make_params(ef=32, recompute=False)
    d, i = PartitionedSearch(lambda qq, k: idx.search_device(qq, k, prm)).search(q, 5)
    ei, ed, _ = orc.search(oracle_graph(g, 64), qn, 5, ef=32, table=x)
    ok = np.array_equal(i.cpu().numpy(), ei) and np.array_equal(d.cpu().numpy(), ed)
    lo, hi = shard_bounds(3000, world)[rank]
    gs = build_hnsw(x[lo:hi], "mips", M=8, ef_construction=40, num_threads=1)
    ids = Mi355xIndex.from_csr(gs, device=rank)
    ids.attach_table(x[lo:hi])
    prs = ids.make_params(ef=64, recompute=False)
    d2, i2 = ShardedSearch(lambda qq, k: ids.search_device(qq, k, prs), id_base=lo, metric=0).search(q, 5)
    gt, _ = orc.bruteforce_topk(x, qn, 5, 0)
    rec = np.mean([len(set(i2[r].tolist()) & set(gt[r].tolist())) / 5 for r in range(q.shape[0])])
    ok = ok and rec > 0.95 and bool(torch.all(torch.diff(d2, dim=1) <= 0)) and int(i2.max()) < 3000
    gi = [torch.empty_like(i2) for _ in range(world)]
    dist.all_gather(gi, i2)
    ok = ok and all(torch.equal(gi[0], t) for t in gi)
    out[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_rccl(built_libs):
    """The first RCCL execution of PartitionedSearch / ShardedSearch / broadcast_graph should not be the driver's 8-GPU run: runs
    wherever two GPUs are visible (the 1-GPU test box skips it)."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs (one rank per GPU over RCCL)")
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_nccl_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}
