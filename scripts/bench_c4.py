#!/usr/bin/env python
"""BASELINE.json configs[3] (C4): 60M chunks (DPR-Wikipedia scale), HNSW graph sharded 8-way, query batch 256, per-shard top-k
merged over RCCL/xGMI.  One process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 scripts/bench_c4.py
    python scripts/bench_c4.py --chunks 7500000          # one rank = one shard of the 8 (what a single GPU can measure)

Every rank builds ITS shard (chunks [rank * n/W, (rank + 1) * n/W): own token store, own embeddings, own HNSW graph with local
ids) and holds the same 256 queries; a step = leann_amd.distributed.ShardedSearch.search: recompute search of all queries on
the local shard, ONE all_gather of the (B, k) x {f32, i64} lists (30 KB per rank at B = 256), per-query merge by the
lm_topk_merge kernel -- all inside the timed region.  Prints ONE JSON line on rank 0 (bench.py's keys + "rccl_ranks")."""
import argparse
import json
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
import torch.distributed as dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=60_000_000, help="TOTAL chunks over all ranks")
    ap.add_argument("--model", default="sentence-transformers/all-MiniLM-L6-v2")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--ef", type=int, default=64)
    ap.add_argument("--M", type=int, default=32)
    ap.add_argument("--efc", type=int, default=200)
    ap.add_argument("--cpu-baseline-seconds", type=float, default=20.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-check", action="store_true", help="skip the untimed GPU-vs-oracle parity check on the shard's own index")
    ap.add_argument("--dry-run-emulated", default=None, metavar="LIB", help="TEST ONLY (tests/test_bench_dry_run.py): as bench.py's switch of this name")
    args = ap.parse_args()
    dry = bool(args.dry_run_emulated)
    if dry:
        import contextlib

        import bench as _bench

        with contextlib.ExitStack() as stack:
            _bench._enter_dry_run(args, stack)
            return _main(args, dry)
    return _main(args, dry)


def _main(args, dry):
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))

    from leann_amd import _lib
    from leann_amd.distributed import ShardedSearch, all_gather_results, shard_bounds
    from leann_amd.encoder import BertEncoder, config_for
    from leann_amd.gpu_graph_build import build_graph_gpu
    from leann_amd.index import Mi355xIndex
    from leann_amd.recompute import RecomputeProvider
    from leann_amd.synth import CorpusSpec, SyntheticCorpus
    from leann_amd.token_store import TokenStore

    _lib.require_gpu()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local) if not dry else torch.device("cpu")
    lib_dev = local if not dry else 0
    if world > 1:
        if dry:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    def log(*a):
        if rank == 0:
            print("[c4]", *a, file=sys.stderr, flush=True)

    B, K, W = args.batch, args.steps, args.warmup
    cfg0 = config_for(args.model)
    kt_dominant = _lib.KT_LAYER_TAIL if cfg0.hidden == 384 and cfg0.ffn % 192 == 0 else _lib.KT_GEMM_F16
    _lib.kernel_timing_enable(1 << kt_dominant)  # event pairs around the dominant encoder kernel, recorded by the library (as bench.py)
    lo, hi = shard_bounds(args.chunks, world)[rank]
    ns = hi - lo
    t_all = time.time()
    # the shard's chunks: an independent corpus per shard (seed = 1234 + rank), ids local to the shard + id_base = lo
    corpus = SyntheticCorpus(CorpusSpec(n_chunks=ns, seed=1234 + rank) if not dry else
                             CorpusSpec(n_chunks=ns, seed=1234 + rank, vocab_size=2000, n_topics=8, len_mean=10.0, len_std=3.0, len_min=4, len_max=20))
    tok, off = corpus.chunks_torch(dev)
    tokens = TokenStore(tok, off, device=lib_dev)
    cfg = config_for(args.model)
    enc = BertEncoder.load(args.model, allow_random=True).to(dev, dtype=torch.float16).eval()
    D = cfg.hidden
    provider = RecomputeProvider(enc, tokens, (D + 63) // 64 * 64, dev)
    X = torch.empty((ns, D), dtype=torch.float32, device=dev)  # build time only: dropped below (60M x 384 x 4 B would be 92 GB in total)
    for b0 in range(0, ns, 32768):
        ids = torch.arange(b0, min(ns, b0 + 32768), dtype=torch.int32, device=dev)
        X[b0 : b0 + ids.shape[0]] = provider.embed_ids(ids)
    g = build_graph_gpu(X, "mips", M=args.M, ef_construction=args.efc)
    log(f"shard of {ns} chunks ready in {time.time() - t_all:.0f}s (mean level-0 degree {g.level0_degrees().mean():.1f})")
    idx = Mi355xIndex.from_csr(g, device=lib_dev)
    idx.set_stream(torch.cuda.current_stream().cuda_stream)
    idx.set_provider(provider)
    # queries: identical on every rank (drawn from shard 0's documents; broadcast from rank 0)
    nq = B * (K + W)
    if rank == 0:
        qt, qo, _ = corpus.queries(nq, seed=4321)
        Q = RecomputeProvider(enc, TokenStore(qt, qo, device=lib_dev), provider.dp, dev).embed_ids(torch.arange(nq, dtype=torch.int32, device=dev)).contiguous()
    else:
        Q = torch.empty((nq, D), dtype=torch.float32, device=dev)
    if world > 1:
        dist.broadcast(Q, 0)
    # exact ground truth over ALL shards: local exact top-10 (query blocks of 64: a (B, shard) score matrix at once would be
    # 256 x 7.5M x 4 B = 7.7 GB, and more at larger batches), gathered and merged
    ld = torch.empty((nq, 10), dtype=torch.float32, device=dev)
    li = torch.empty((nq, 10), dtype=torch.int64, device=dev)
    from leann_amd.exact import exact_topk_ip

    ld[:], li[:] = exact_topk_ip(Q, X, 10, q_block=64)
    li = li + lo
    if world > 1:
        gd = [torch.empty_like(ld) for _ in range(world)]
        gi = [torch.empty_like(li) for _ in range(world)]
        dist.all_gather(gd, ld)
        dist.all_gather(gi, li)
        ad, ai = torch.cat(gd, 1), torch.cat(gi, 1)
        top = torch.topk(ad, 10, dim=1).indices
        gt = torch.gather(ai, 1, top).cpu().numpy()
    else:
        gt = li.cpu().numpy()
    prm = idx.make_params(ef=args.ef, beam=1, recompute=True, max_batch=B)
    ss = ShardedSearch(lambda qq, k: idx.search_device(qq, k, prm), id_base=lo, metric=g.metric_type)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for w in range(W):
        ss.search(Q[w * B : (w + 1) * B].contiguous(), 10)
    labels = []
    agg = {"ndis": 0, "nunique": 0, "nrounds": 0}
    barrier()
    _lib.kernel_timing_read(reset=True)
    t0 = time.perf_counter()
    for st in range(K):
        _, l = ss.search(Q[(W + st) * B : (W + st + 1) * B].contiguous(), 10)
        labels.append(l)
        st_ = idx.stats()
        for k_ in agg:
            agg[k_] += st_[k_]
    barrier()
    elapsed = time.perf_counter() - t0
    kt_timed = _lib.kernel_timing_read(reset=True)
    _lib.kernel_timing_enable(0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    lab = torch.cat(labels).cpu().numpy()
    rec = float(np.mean([len(set(lab[i]) & set(gt[W * B + i])) / 10 for i in range(K * B)]))
    # the collective step alone (all_gather + merge of B x k lists), timed separately
    d0, i0 = idx.search_device(Q[:B].contiguous(), 10, prm)
    barrier()
    t0 = time.perf_counter()
    for _ in range(20):
        if world > 1:
            gd, gi = all_gather_results(d0, i0, world)  # the packed exchange ShardedSearch.search does
            ss.merge_fn(gi.contiguous(), gd.contiguous(), g.metric_type)
        else:
            ss.merge_fn(i0[None].contiguous(), d0[None].contiguous(), g.metric_type)
    barrier()
    coll_us = (time.perf_counter() - t0) / 20 * 1e6
    # ---- parity on THIS shard's own index and queries (untimed; rank 0): GPU vs the CPU oracle exactly as bench.py's parity_check does it
    #      for C2 -- stored-embedding mode (ids, distances, evaluation counts; the faiss transcription as well) and recompute mode
    #      (the oracle replays the GPU encoder's own per-round outputs) -- then the shard's embeddings are dropped ----
    parity = None
    if rank == 0 and not args.no_parity_check:
        try:
            import bench as _b

            t1 = time.time()
            idx.attach_table(X)
            parity = _b.parity_check(idx, g, X, Q, provider, args.ef, 1, D, n_table=min(64, B), n_recompute=min(16, max(1, nq - 64)))
            parity["what"] = (f"shard-level: the searches of rank 0's shard ({ns} chunks) on the run's own queries against oracle/lm_oracle.c; the cross-shard merge "
                              "(lm_topk_merge) has its own oracle test (tests/test_distributed.py, tests/emulated_two_rank.py)")
            parity["seconds"] = round(time.time() - t1, 1)
        except Exception as ex:  # noqa: BLE001 - an untimed check may never cost the line
            parity = {"error": repr(ex)[:300]}
    del X
    torch.cuda.empty_cache()
    # roofline of the dominant kernel of the timed region (the fused layer tail, MFMA bound), as in bench.py
    kname = "lm::k_layer_tail_h384" if kt_dominant == _lib.KT_LAYER_TAIL else "lm::k_gemm_f16"
    kt = kt_timed.get(kname)
    roofline = None
    if kt and kt["ms"] > 0:
        tf = kt["work"] / (kt["ms"] * 1e-3) / 1e12
        fpt = 4 * cfg.ffn * cfg.hidden + 2 * cfg.hidden * cfg.hidden if kt_dominant == _lib.KT_LAYER_TAIL else None
        roofline = {"bound": "mfma", "kernel": kname + (" (attention output projection + LayerNorm + feed-forward block + LayerNorm, generation 4)" if fpt else " (general MFMA GEMM)"),
                    "achieved": round(tf, 2), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tf / 2500.0, 5), "traffic": None,
                    "flops_per_token": fpt, "launches": kt["launches"], "avg_launch_us": round(1e3 * kt["ms"] / kt["launches"], 1),
                    "tokens_per_launch": round(kt["work"] / fpt / kt["launches"]) if fpt else None, "share_of_timed_region": round(kt["ms"] / (elapsed * 1e3), 4),
                    "timing": "library-side HIP event pairs around every launch of the timed region, rank 0 (csrc/lm_timing.cpp); library-side recompute provider"}
    cpu_base = None
    if rank == 0 and not args.no_cpu_baseline:  # the oracle traversal + fp32 CPU encoder on shard 0, a bounded sample of the same queries
        try:
            sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
            import bench as _b

            class _A:
                model, cpu_baseline_seconds = args.model, args.cpu_baseline_seconds

            tok_np = tok.cpu().numpy() if isinstance(tok, torch.Tensor) else tok
            off_np = off.cpu().numpy() if isinstance(off, torch.Tensor) else off
            cpu_base = _b.cpu_baseline(_A, g, Q, tok_np, off_np, cfg, args.ef, 1)
            cpu_base["sample"] += f"; ONE shard of {ns} chunks on this box's host cores (the {world}-shard job needs every shard searched: divide by {world} for one host)"
        except Exception as ex:  # noqa: BLE001
            cpu_base = {"value": None, "unit": "queries/s", "cores": 0, "kind": "port", "sample": "failed: " + repr(ex)[:200]}
    if rank == 0:
        print(json.dumps({
            "metric": f"queries/sec, {args.chunks}-chunk HNSW sharded {world}-way, query batch {B}, RCCL all_gather of per-shard top-k + merge",
            "value": round(K * B / elapsed, 3), "unit": "queries/s", "n_gpus": world, "rccl_ranks": world, "steps": K, "warmup": W,
            "ms_per_step": round(1e3 * elapsed / K, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp16",
            "dtype_detail": "encoder fp16 MFMA with fp32 accumulation; distances / merge f32",
            "data": "synthetic" if not dry else "dry-run (emulated library on the CPU: control flow only, nothing here is a measurement)", "dry_run": dry,
            "roofline": roofline, "cpu_baseline": cpu_base, "parity_check": parity,
            "what_this_line_is": (f"the {world}-rank job" if world > 1 or args.chunks >= 60_000_000 // 2 else
                                  f"ONE rank's work of C4 -- one shard of {ns} chunks, all {B} queries, local top-10, exchange + lm_topk_merge in the timed region "
                                  "at world size 1 -- on one MI355X; no 8-GPU run exists (no multi-GPU node was available to this repo in any round)"),
            "per_query": {"distance_evals": round(agg["ndis"] / max(K * B, 1), 1), "recomputed_chunks": round(agg["nunique"] / max(K * B, 1), 1),
                          "rounds_per_step": round(agg["nrounds"] / max(K, 1), 1)},
            "config": {"workload": f"{args.chunks} synthetic chunks in {world} shard(s) of {ns}, HNSW M={args.M} per shard (GPU-built), {args.model} shape, "
                                   f"ef_search={args.ef}, beam=1, top-10, {B} queries per step searched on EVERY shard, all_gather + lm_topk_merge in the timed region",
                       "baseline_config": "c4", "n_chunks": args.chunks, "shard_chunks": ns, "queries_per_step": B,
                       "multi_gpu_path": "leann_amd.distributed.ShardedSearch"},
            "recall_at_10": round(rec, 4), "allgather_plus_merge_us": round(coll_us, 1),
            "exchange_bytes_per_rank": B * 10 * 12, "setup_s": round(time.time() - t_all)}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
