"""SURVEY 8(e) on the CPU box with the product's OWN kernels: two ranks over `gloo`, each holding the emulated product library
(tests/hip_emul) instead of an oracle stand-in -- broadcast_graph, PartitionedSearch (query partition + one packed all_gather) and
ShardedSearch (per-shard search + one packed all_gather + the lm_topk_merge kernel) exactly as tests/test_distributed.py:_nccl_worker
runs them on two GPUs over RCCL.  Spawned by tests/test_emulated_search.py; only this file pretends host tensors are device tensors."""
import os
from unittest import mock

import numpy as np
import torch
import torch.distributed as dist


class _Stream:
    cuda_stream = 0


def worker(rank: int, world: int, port: int, lib_path: str, out) -> None:
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from tests.emulated_search_cases import _load

    _load(lib_path)
    from leann_amd.distributed import PartitionedSearch, ShardedSearch, broadcast_graph, shard_bounds
    from leann_amd.hnsw_builder import build_hnsw
    from leann_amd.index import Mi355xIndex
    from oracle import oracle as orc
    from tests.util import clustered, oracle_graph, queries_near

    orc.set_num_threads(1)
    n = 900
    x = clustered(n, 64, 5)
    qn = queries_near(x, 11, 6)
    q = torch.from_numpy(qn)
    with mock.patch.object(torch.Tensor, "is_cuda", new=property(lambda self: True)), \
            mock.patch("torch.cuda.current_stream", new=lambda *a, **k: _Stream()):
        # replicated graph (built on rank 0, broadcast), partitioned queries, one packed all_gather of the results
        g = broadcast_graph(build_hnsw(x, "mips", M=8, ef_construction=40, num_threads=1) if rank == 0 else None, 0)
        idx = Mi355xIndex.from_csr(g)
        idx.attach_table(x)
        prm = idx.make_params(ef=32, recompute=False)
        d, i = PartitionedSearch(lambda qq, k: idx.search_device(qq.contiguous(), k, prm)).search(q, 5)
        ei, ed, _ = orc.search(oracle_graph(g, 64), qn, 5, ef=32, table=x)
        ok = np.array_equal(i.numpy(), ei) and np.array_equal(d.numpy(), ed)
        # sharded graph: every rank searches all queries on its shard; one packed all_gather + the lm_topk_merge kernel
        lo, hi = shard_bounds(n, world)[rank]
        gs = build_hnsw(x[lo:hi], "mips", M=8, ef_construction=40, num_threads=1)
        ids = Mi355xIndex.from_csr(gs)
        ids.attach_table(x[lo:hi])
        prs = ids.make_params(ef=64, recompute=False)
        d2, i2 = ShardedSearch(lambda qq, k: ids.search_device(qq.contiguous(), k, prs), id_base=lo, metric=0).search(q, 5)
        # the merged answer = the oracle's merge of the oracle's per-shard answers (same tie-break), on every rank
        parts_i, parts_d = [], []
        for r in range(world):
            l2, h2 = shard_bounds(n, world)[r]
            gr = gs if r == rank else build_hnsw(x[l2:h2], "mips", M=8, ef_construction=40, num_threads=1)
            oi, od, _ = orc.search(oracle_graph(gr, 64), qn, 5, ef=64, table=x[l2:h2])
            parts_i.append(np.where(oi >= 0, oi + l2, -1))
            parts_d.append(od)
        mi, md = orc.merge_topk(np.stack(parts_i), np.stack(parts_d), 0)
        ok = ok and np.array_equal(i2.numpy(), mi) and np.array_equal(d2.numpy(), md)
        gt, _ = orc.bruteforce_topk(x, qn, 5, 0)
        rec = np.mean([len(set(i2[r].tolist()) & set(gt[r].tolist())) / 5 for r in range(q.shape[0])])
        ok = ok and rec > 0.95 and bool(torch.all(torch.diff(d2, dim=1) <= 0)) and int(i2.max()) < n
        idx.close()
        ids.close()
    out[rank] = bool(ok)
    dist.destroy_process_group()
