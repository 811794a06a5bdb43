#!/usr/bin/env python
"""bench.py -- queries/sec of the selective-recompute beam search at recall@10 >= 0.9.

Workload (BASELINE.json configs[1]): 1M synthetic text chunks, HNSW graph (M=32), all-MiniLM-L6-v2
shaped encoder (384-d, seeded random weights -- no checkpoints offline), embeddings recomputed at
query time, 1 x MI355X.  A "step" is one pass of the hot path over one batch of B queries:
lm_index_search_device(recompute=1) -> per round: CSR expand / visited / dedup kernels, HBM token
gather, BERT forward (hand-written fp16 MFMA kernels), fused distance + beam-update kernel.  Queries, graph,
token store and results are HBM resident when the timed region starts.

`value` is the library's DEFAULT search call: lm_search_params_default() sets recompute_memo = 1, i.e. within ONE call (= one step)
an embedding that was recomputed for one query / round is kept in HBM until the call returns and never recomputed for another
query / round of the same call (the reference's own switch of that kind: dedup_node_dis, diskann_backend.py:394,413).  Nothing
survives a call: every step starts from an empty memo, ids and distances are bit-identical to the memo-less search (checked in
this run on the timed queries themselves).  The memo-less rate (dedup per lock-step round only -- what rounds 1 and 2 reported
as `value`) is timed in the same run on the same queries and printed as `without_call_memo`.

The timed steps run the PRODUCT DEFAULT path end to end: the library-side recompute provider (csrc/lm_recompute.hip: ids -> token
store -> packed forward inside lm_index_search_device, no interpreter in the search loop) over the default kernel set (the one
`pytest -m gpu` tests).  Instrumentation inside the timed region: ONE HIP event pair around each launch of the dominant kernel,
recorded BY THE LIBRARY on the launch stream (csrc/lm_timing.cpp, lm_kernel_timing_enable; ~1k pairs per step, ~2 us each
against ~1.7 ms launches -- stated in the JSON line as roofline.instrumentation); the search library's own per-launch profiling
(event pairs + device span stamps around the distance kernels) and the event pairs of the other encoder kernels are OFF in the
timed steps and ON only in one extra profiled step, which is where roofline_distance_kernel / roofline_encoder /
encoder_kernels_profiled_step come from.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
For N > 1 the driver launches one rank per GPU through torch.distributed.run; the graph and the
token store are replicated, the query batch is partitioned across ranks, there is no collective on
the data path (weak scaling: B queries per rank per step).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--chunks", type=int, default=1_000_000)
    ap.add_argument("--batch", type=int, default=2048, help="queries per rank per step")
    ap.add_argument("--ef", type=int, default=64, help="efSearch of the timed steps (BASELINE.json configs[1]: 64); 0 = smallest of the sweep with recall@10 >= 0.9")
    ap.add_argument("--no-min-ef-step", action="store_true", help="skip the extra step at the smallest ef reaching recall 0.9")
    ap.add_argument("--beam", type=int, default=1)
    ap.add_argument("--model", default="sentence-transformers/all-MiniLM-L6-v2")
    ap.add_argument("--M", type=int, default=32)
    ap.add_argument("--efc", type=int, default=200)
    ap.add_argument("--cpu-baseline-seconds", type=float, default=20.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--table-roofline", action="store_true", default=True, help="also time the stored-embedding (HBM gather) mode (default on)")
    ap.add_argument("--no-table-roofline", dest="table_roofline", action="store_false")
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4", "c5"],
                    help="c2 (default, BASELINE.json configs[1]): 1M chunks, MiniLM-L6; c5 (configs[4]): 10M chunks, bge-base-en-v1.5 768-d fp16, "
                         "query batch 1024 split over the ranks; c3 (configs[2], 10M DiskANN-style) and c4 (configs[3], 60M sharded over the "
                         "ranks, all_gather + merge timed) run scripts/bench_c3.py / scripts/bench_c4.py with this command's --steps/--warmup")
    ap.add_argument("--no-parity-check", action="store_true", help="skip the untimed GPU-vs-oracle parity check on the benchmark's own index")
    ap.add_argument("--no-latency-rows", action="store_true", help="skip the small-batch (B = 1, 16, 64, 256) latency rows and value_by_batch")
    ap.add_argument("--no-provider-ab", action="store_true", help="skip the extra full-size step over the Python form of the provider")
    ap.add_argument("--extra-batches", default="", help="comma-separated batch sizes: one extra search call each on fresh queries after the timed steps (library defaults), "
                    "reported as extra_batch_rows -- e.g. 128 = the per-rank batch of C5 (query batch 1024) on 8 GPUs")
    ap.add_argument("--fixed-len", type=int, default=0, help="SURVEY 8(d) variant: every chunk exactly this many tokens (256: 6.06 GFLOP per chunk), instead of len ~ N(180, 50)")
    ap.add_argument("--box-probe-only", action="store_true",
                    help="run only the box probe's kernel launches (the 262,107-token layer tail) and print its line: what the probe's rocprofv3 --pmc child runs")
    ap.add_argument("--box-survey", action="store_true",
                    help="the whole box probe alone, L2 / fabric counter pass included (~1 GPU-minute): one JSON line per box, for a table of boxes (DESIGN 6.1)")
    ap.add_argument("--no-box-probe", action="store_true", help="skip the box probe (clock / power sampling, reference launches of the dominant kernel, copy rate)")
    ap.add_argument("--dry-run-emulated", default=None, metavar="LIB",
                    help="TEST ONLY (tests/test_bench_dry_run.py): run this script's control flow -- incl. every world > 1 branch, over gloo -- on the CPU against "
                         "the thread-per-lane build of the library (tests/hip_emul), with a tiny model and corpus; the line says data = dry-run and measures nothing")
    args = ap.parse_args()
    if args.box_probe_only:
        return _box_probe_only()
    if args.box_survey:
        return _box_survey()
    if args.dry_run_emulated:
        import contextlib

        with contextlib.ExitStack() as stack:
            _enter_dry_run(args, stack)
            return _main(args, ap)
    return _main(args, ap)


def _enter_dry_run(args, stack):
    """Everything the CPU dry run needs: the emulated library instead of the product's, host tensors pretending to be device tensors
    (as tests/emulated_two_rank.py does for the library's own multi-GPU classes), a tiny general-width model preset and corpus."""
    from unittest import mock

    import torch

    from leann_amd import _lib, encoder

    _lib.LIB_PATH = Path(args.dry_run_emulated)
    _lib._lib = None

    class _Stream:
        cuda_stream = 0

    stack.enter_context(mock.patch.object(torch.Tensor, "is_cuda", new=property(lambda self: True)))
    stack.enter_context(mock.patch("torch.cuda.current_stream", new=lambda *a, **k: _Stream()))
    stack.enter_context(mock.patch("torch.cuda.synchronize", new=lambda *a, **k: None))
    stack.enter_context(mock.patch("torch.cuda.set_device", new=lambda *a, **k: None))
    encoder.PRESETS["dry-run-tiny"] = encoder.EncoderConfig(vocab_size=2000, hidden=128, layers=1, heads=4, ffn=128, max_pos=32, pooling="mean",
                                                            max_seq_length=24)
    args.model = "dry-run-tiny"
    torch.set_num_threads(1)


def _main(args, ap):
    dry = bool(args.dry_run_emulated)
    if args.config in ("c3", "c4"):  # own entry points (different index type / sharded index); same JSON contract
        import runpy

        defaults = {a.dest: a.default for a in ap._actions}
        argv = ["--steps", str(args.steps), "--warmup", str(args.warmup)]
        if args.chunks != defaults["chunks"]:
            argv += ["--chunks", str(args.chunks)]
        if args.batch != defaults["batch"]:
            argv += ["--batch", str(args.batch)]
        script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scripts", f"bench_{args.config}.py")
        sys.argv = [script] + argv
        runpy.run_path(script, run_name="__main__")
        return
    if args.config == "c5":  # BASELINE.json configs[4]; explicit --chunks / --model / --batch still win
        defaults = {a.dest: a.default for a in ap._actions}
        if args.chunks == defaults["chunks"]:
            args.chunks = 10_000_000
        if args.model == defaults["model"]:
            args.model = "BAAI/bge-base-en-v1.5"
        if args.batch == defaults["batch"]:
            args.batch = max(1, 1024 // int(os.environ.get("WORLD_SIZE", "1")))

    import torch
    import torch.distributed as dist

    from leann_amd import _lib
    from leann_amd.encoder import BertEncoder, config_for
    from leann_amd.gpu_graph_build import build_graph_gpu
    from leann_amd.index import Mi355xIndex
    from leann_amd.recompute import RecomputeProvider
    from leann_amd.synth import CorpusSpec, SyntheticCorpus
    from leann_amd.token_store import TokenStore

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    _lib.require_gpu()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank) if not dry else torch.device("cpu")
    lib_dev = local_rank if not dry else 0  # (the emulated library is one "device")
    if world > 1:
        if dry:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    K, W, B = args.steps, args.warmup, args.batch
    # HIP-event pairs around every launch of the dominant kernel for the WHOLE process, recorded by the library itself (csrc/lm_timing.cpp)
    # and read out phase by phase: the roofline line uses the launches of the timed region; the all-launch average is what a rocprofv3
    # --kernel-trace --stats table of this same command reports for the kernel.
    kt_dominant = _lib.KT_LAYER_TAIL if config_for(args.model).hidden == 384 and config_for(args.model).ffn % 192 == 0 else _lib.KT_GEMM_F16
    KT_NAMES = {_lib.KT_LAYER_TAIL: "lm::k_layer_tail_h384", _lib.KT_GEMM_WS: "lm::k_gemm_ws_h384", _lib.KT_ATTN: "lm::k_attn_varlen", _lib.KT_GEMM_F16: "lm::k_gemm_f16",
                _lib.KT_QKV: "lm::k_qkv_h384", _lib.KT_QKV_ATTN: "lm::k_qkv_attn_h384"}
    _lib.kernel_timing_enable(1 << kt_dominant)
    kt_phase = {}  # phase -> {kernel: {"launches", "ms", "work"}}

    def kt_close(phase):  # fold everything recorded since the previous call into `phase`
        cur = _lib.kernel_timing_read(reset=True)
        acc = kt_phase.setdefault(phase, {})
        for k_, v_ in cur.items():
            a_ = acc.setdefault(k_, {"launches": 0, "ms": 0.0, "work": 0.0})
            for f_ in a_:
                a_[f_] += v_[f_]
        return acc

    EXTRA_ROWS = 8192 if not dry else 32  # small-batch latency rows (library-side provider, then the Python provider) + parity-check queries (fresh, after every step's block)
    n_q = B * (K + W + 5) + EXTRA_ROWS  # +5: the profiled step and the extra steps (smallest ef reaching recall 0.9; hub cache; two-level search; one spare)
    t_setup = time.time()

    # ---- corpus -> HBM token store ------------------------------------------------------------
    spec = (CorpusSpec(n_chunks=args.chunks, seed=1234, vocab_size=2000, n_topics=8, len_mean=10.0, len_std=3.0, len_min=4, len_max=20) if dry else
            CorpusSpec(n_chunks=args.chunks, seed=1234) if not args.fixed_len else
            CorpusSpec(n_chunks=args.chunks, seed=1234, len_mean=float(args.fixed_len), len_std=0.0, len_min=args.fixed_len, len_max=args.fixed_len))
    corpus = SyntheticCorpus(spec)
    # 2M chunks and more (C5: 10M): the same generative model evaluated on the GPU (3 s instead of 195 s per 10M chunks; deterministic, but
    # not the numpy path's bits -- the 1M-chunk default keeps the numpy corpus every round so far has been measured on)
    tok, off = corpus.chunks_torch(dev) if (args.chunks >= 2_000_000 and not dry) else corpus.chunks()
    tokens = TokenStore(tok, off, device=lib_dev)
    log(f"corpus: {args.chunks} chunks, {int(off[-1])} tokens ({time.time() - t_setup:.1f}s)")

    # ---- encoder ------------------------------------------------------------------------------
    cfg = config_for(args.model)
    enc = BertEncoder.load(args.model, allow_random=True).to(dev, dtype=torch.float16).eval()
    D = cfg.hidden
    provider = RecomputeProvider(enc, tokens, (D + 63) // 64 * 64, dev)

    # ---- embeddings of every chunk + graph (index build time only): built ONCE, on rank 0, then replicated over RCCL ----
    t0 = time.time()
    X = torch.empty((args.chunks, D), dtype=torch.float32, device=dev)
    g = None
    t_embed = t_graph = 0.0
    if rank == 0:
        step = 32768
        for b0 in range(0, args.chunks, step):
            ids = torch.arange(b0, min(args.chunks, b0 + step), dtype=torch.int32, device=dev)
            X[b0 : b0 + ids.shape[0]] = provider.embed_ids(ids)
        torch.cuda.synchronize()
        t_embed = time.time() - t0
        log(f"embedded corpus in {t_embed:.1f}s ({args.chunks / t_embed:.0f} chunks/s)")
        t0 = time.time()
        g = build_graph_gpu(X, "mips", M=args.M, ef_construction=args.efc, verbose=bool(os.environ.get("BENCH_VERBOSE")))
        t_graph = time.time() - t0
    if world > 1:
        from leann_amd.distributed import broadcast_graph

        t0 = time.time()
        dist.broadcast(X, 0)
        g = broadcast_graph(g, 0, device=dev)
        torch.cuda.synchronize()
        log(f"index replicated to {world} ranks in {time.time() - t0:.1f}s")
    deg0 = g.level0_degrees()
    log(f"graph built in {t_graph:.1f}s: max_level={g.max_level} mean level-0 degree={deg0.mean():.1f} edges={g.neighbors.shape[0]}")
    idx = Mi355xIndex.from_csr(g, device=lib_dev)
    idx.set_stream(torch.cuda.current_stream().cuda_stream)

    # ---- queries + exact ground truth ------------------------------------------------------------
    qt, qo, _ = corpus.queries(n_q * world, seed=4321)
    qstore = TokenStore(qt, qo, device=lib_dev)
    Q_all = RecomputeProvider(enc, qstore, provider.dp, dev).embed_ids(torch.arange(n_q * world, dtype=torch.int32, device=dev))
    Q = Q_all[rank * n_q : (rank + 1) * n_q].contiguous()
    from leann_amd.exact import exact_topk_ip

    gt_np = exact_topk_ip(Q, X, 10)[1].cpu().numpy()  # blocked: no GEMM call with more than 2^28 output elements (leann_amd/exact.py says why)

    def recall(labels_np, rows):
        hit = 0
        for i, r in enumerate(rows):
            hit += len(set(labels_np[i].tolist()) & set(gt_np[r].tolist()))
        return hit / (10 * len(rows))

    # ---- ef selection in stored-embedding mode (same traversal, no encoder) -----------------------
    idx.attach_table(X)
    sweep = {}
    nsel = min(n_q, 512)
    for ef in (16, 32, 64, 128, 256):
        _, l = idx.search_device(Q[:nsel], 10, idx.make_params(ef=ef, beam=args.beam, recompute=False))
        st = idx.stats()
        sweep[ef] = {"recall": recall(l.cpu().numpy(), range(nsel)), "ndis_per_query": st["ndis"] / nsel}
    log("ef sweep (stored-embedding mode):", json.dumps(sweep))
    ef_min = next((e for e in sorted(sweep) if sweep[e]["recall"] >= 0.9), 256)
    ef = args.ef or ef_min

    # optional: HBM-gather roofline of the fused kernel in stored-embedding mode, big batch
    table_roof = None
    extras_errors = {}
    if args.table_roofline and world == 1:
        try:
            idx.set_profiling(True)
            # 32768 DISTINCT queries of their own (held-out chunks, another seed than the timed queries'): no row gather of this measurement is a
            # guaranteed re-read of the same query's twin.  32768 in flight, not 8192 as until round 5: a stored-embedding search is a chain of
            # dependent gathers per query, so the kernel's rate is set by the queries resident per CU -- 8192 / 16384 / 32768 in flight: 5.1 / 6.0 /
            # 6.5 TB/s at beam 1 (profiles/r6_table_mode_queries_in_flight_and_occupancy_sweep.json); the 8192-query rows stay for continuity
            nbig = 32768 if not dry else 16
            bt, bo, _ = corpus.queries(nbig, seed=97531)
            Qbig = RecomputeProvider(enc, TokenStore(bt, bo, device=lib_dev), provider.dp, dev).embed_ids(torch.arange(nbig, dtype=torch.int32, device=dev)).contiguous()
            bytes_eval = D * 4 + 4
            table_roof = {"queries_in_flight": nbig, "distinct_queries": int(torch.unique(Qbig, dim=0).shape[0])}
            try:  # L2-miss-side bytes of this kernel from its own FETCH_SIZE pass (scripts/pmc_table_mode.sh), as a ratio to the algorithmic bytes
                pm = json.loads((ROOT / "profiles" / "r6_pmc_table_mode_32768_queries.json").read_text())
                table_roof["traffic"] = {f"beam{c['beam']}": {"l2_miss_side_over_algorithmic": c["l2_miss_side_over_algorithmic"], "l2_miss_side_bytes_per_launch": c["l2_miss_side_bytes"],
                                                              "algorithmic_bytes_per_launch": c["algorithmic_bytes"]} for c in pm["run"]["calls"][-2:]}
                table_roof["traffic_source"] = ("profiles/r6_pmc_table_mode_32768_queries.json: rocprofv3 --pmc FETCH_SIZE over scripts/pmc_table_mode.py (same index, 32768 distinct queries), calibrated "
                                                "in-pass on a read of every table row once in the kernel's own access pattern; FETCH_SIZE counts L2 misses (Infinity-Cache hits included)")
            except Exception:  # noqa: BLE001
                pass
            # interleaved A/B: persistent one-launch kernel (wave / workgroup per query; -1 = the library's own choice) vs lock-step rounds
            for persistent, wave, nq_t in ((1, -1, nbig), (1, 1, nbig), (1, 0, nbig), (1, -1, min(8192, nbig)), (0, 0, min(8192, nbig))) * 2:
                for beam_t, ef_t in ((4, ef), (1, ef)):
                    idx.set_option("persistent_table", persistent)
                    idx.set_option("persistent_wave", wave)
                    prm = idx.make_params(ef=ef_t, beam=beam_t, recompute=False, max_batch=32768)
                    idx.search_device(Qbig[:nq_t], 10, prm)
                    st = idx.stats()
                    form = {-1: "auto", 1: "wave_per_query", 0: "workgroup_per_query"}[wave]
                    key = f"{'k_search_table_persistent_' + form if persistent else 'lockstep_k_update'}_beam{beam_t}_ef{ef_t}" + ("" if nq_t == nbig else f"_{nq_t}_queries")
                    net_ms = max(st["update_span_ms"], 1e-6)
                    r = {"GBps": round(st["ndis"] * bytes_eval / (net_ms * 1e-3) / 1e9, 1), "launches": st["update_launches"],
                         "us_per_launch": round(1e3 * net_ms / max(st["update_launches"], 1), 2),
                         "us_per_launch_event_pair": round(1e3 * st["update_ms"] / max(st["update_launches"], 1), 2),
                         "expand_us_per_launch": round(1e3 * st["expand_ms"] / max(st["update_launches"], 1), 2)}
                    tr_ = (table_roof.get("traffic") or {}).get(f"beam{beam_t}")
                    if tr_ and persistent and nq_t == nbig:
                        r["GBps_l2_miss_side"] = round(r["GBps"] * tr_["l2_miss_side_over_algorithmic"], 1)
                        r["frac_of_8TBps_algorithmic"] = round(r["GBps"] / 8000.0, 4)
                        r["frac_of_8TBps_l2_miss_side"] = round(r["GBps_l2_miss_side"] / 8000.0, 4)
                    table_roof.setdefault(key, []).append(r)
            idx.set_option("persistent_table", 1)
            idx.set_option("persistent_wave", -1)
            idx.set_profiling(False)
        except Exception as ex:  # noqa: BLE001 - an optional measurement must never cost the headline line
            extras_errors["roofline_table_mode"] = repr(ex)[:300]
            table_roof = None
            idx.set_option("persistent_table", 1)
            idx.set_option("persistent_wave", -1)
            idx.set_profiling(False)

    # ---- timed region: recompute mode ---------------------------------------------------------------
    idx.set_provider(provider)
    prm = idx.make_params(ef=ef, beam=args.beam, recompute=True, max_batch=B)
    idx.set_profiling(False)  # the timed steps carry no event pairs / span stamps; one extra profiled step follows them
    setup_s = time.time() - t_setup
    log(f"setup done in {setup_s:.1f}s; ef={ef}; timing {K} steps x {B} queries (+{W} warmup)")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The step goes through the library's own multi-GPU host class (leann_amd/distributed.py: PartitionedSearch): the step's
    # GLOBAL query batch (B per rank, identical tensor on every rank, HBM resident before the clock starts) is partitioned
    # across the ranks, every rank searches its slice on its replica, and the (B x world, k) results are gathered on
    # every rank with one RCCL all_gather -- SURVEY 8(e).  With one rank it is the plain device search.
    from leann_amd.distributed import PartitionedSearch

    ps = PartitionedSearch(lambda qq, k: idx.search_device(qq, k, prm))

    def global_batch(step):
        lo_ = step * B
        if world == 1:
            return Q[lo_ : lo_ + B]
        return Q_all.view(world, n_q, D)[:, lo_ : lo_ + B].reshape(world * B, D).contiguous()

    batches = [global_batch(w) for w in range(W + K)]
    out_labels = []
    # what box is this?  (reference launches of the dominant kernel + a copy rate on the idle chip, before anything is timed)
    probe = None
    if cfg.hidden == 384 and not dry and not args.no_box_probe:
        probe = box_probe(enc, dev, local_rank, counter_pass=(rank == 0 and world == 1))  # (the counter pass runs a child process on device 0: single-GPU runs only)
        log("box probe:", json.dumps(probe)[:600])
    kt_close("setup")
    for w in range(W):
        ps.search(batches[w], 10)
    agg = {"ndis": 0, "nunique": 0, "nrounds": 0, "update_launches": 0}
    provider.chunks = 0
    barrier()
    kt_close("warmup")
    sampler = BoxSampler(local_rank, period_s=0.25).start() if (probe is not None and rank == 0) else None  # a thread reading sysfs four times a second (rocm-smi every 5 s where sysfs has no clocks)
    t0 = time.perf_counter()
    for s in range(K):
        _, l = ps.search(batches[W + s], 10)
        out_labels.append(l[rank * B : (rank + 1) * B])
        st = idx.stats()
        for k_ in agg:
            agg[k_] += st[k_]
    barrier()
    elapsed = time.perf_counter() - t0
    if sampler is not None:
        probe["clocks_during_the_timed_steps"] = sampler.stop()
    ktimes = kt_close("timed")  # (reading waits for the last pairs: after the clock has stopped)
    # ---- the same steps WITHOUT the per-call memo (dedup per lock-step round only: rounds 1 / 2's `value`), on the same queries:
    #      K2 = min(K, 3) steps (one, when a step takes more than 20 s) after one warm-up step, same bracketing; labels must be identical to
    #      the memo steps' ----------
    K2 = min(K, 3) if elapsed / max(K, 1) < 20.0 else min(K, 1)  # (a 58-second step -- C5 at 10M chunks -- gets one memo-off step, not three)
    prm_nomemo = idx.make_params(ef=ef, beam=args.beam, recompute=True, max_batch=B, recompute_memo=False)
    ps_nm = PartitionedSearch(lambda qq, k: idx.search_device(qq, k, prm_nomemo))
    if K2:
        ps_nm.search(batches[0], 10)
    agg_nm = {"ndis": 0, "nunique": 0, "nrounds": 0}
    same_labels = True
    barrier()
    kt_close("warmup_no_memo")
    t0 = time.perf_counter()
    nm_labels = []
    for s in range(K2):
        _, l = ps_nm.search(batches[W + s], 10)
        nm_labels.append(l[rank * B : (rank + 1) * B])
        st = idx.stats()
        for k_ in agg_nm:
            agg_nm[k_] += st[k_]
    barrier()
    elapsed_nm = time.perf_counter() - t0
    for s in range(K2):
        same_labels &= bool(torch.equal(nm_labels[s], out_labels[s]))
    del nm_labels
    kt_close("timed_no_memo")
    del batches
    if world > 1:
        t = torch.tensor([elapsed, elapsed_nm, 0.0 if same_labels else 1.0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, elapsed_nm, same_labels = float(t[0].item()), float(t[1].item()), float(t[2].item()) == 0.0
    # ---- one more step of the same workload WITH profiling (HIP event pairs + device span stamps per launch): the
    #      source of the roofline figures; not part of `value` ----------------------------------------------------
    idx.set_profiling(True)
    _lib.kernel_timing_enable((1 << _lib.KT_COUNT) - 1)  # every instrumented encoder kernel, this one step only
    lo = (W + K) * B
    torch.cuda.synchronize()
    t1p = time.perf_counter()
    idx.search_device(Q[lo : lo + B], 10, prm)
    torch.cuda.synchronize()
    prof_step_s = time.perf_counter() - t1p
    prof = idx.stats()
    kprof = kt_close("profiled")
    _lib.kernel_timing_enable(1 << kt_dominant)
    idx.set_profiling(False)
    # ---- extras (NOT `value`; single-GPU runs only -- they contain no collectives and may never cost the headline
    #      line): one extra step each, on fresh queries ----------------------------------------------------------
    min_ef = with_hub = two_level = None
    do_extras = world == 1 and not args.no_min_ef_step

    def extra_step(params, slot):
        lo_ = (W + K + 1 + slot) * B
        torch.cuda.synchronize()
        t1_ = time.perf_counter()
        _, lx = idx.search_device(Q[lo_ : lo_ + B], 10, params)
        torch.cuda.synchronize()
        ex_ = time.perf_counter() - t1_
        return ex_, round(recall(lx.cpu().numpy(), range(lo_, lo_ + B)), 4), idx.stats()

    if do_extras and ef_min != ef:  # the smallest ef of the sweep with recall@10 >= 0.9
        try:
            e2, r2, _ = extra_step(idx.make_params(ef=ef_min, beam=args.beam, recompute=True, max_batch=B), 0)
            min_ef = {"ef_search": ef_min, "queries_per_s": round(B / e2, 3), "recall_at_10": r2, "steps": 1}
        except Exception as ex:  # noqa: BLE001
            extras_errors["at_min_ef"] = repr(ex)[:300]
    if do_extras:  # hub-embedding cache, 10 % highest in-degree nodes (LEANN paper section 5)
        try:
            from leann_amd.backend import hub_nodes

            hubs = hub_nodes(g, 0.10)
            idx.set_hub_cache(hubs, X[torch.from_numpy(hubs).long().to(dev)].contiguous())
            e5, r5, st5 = extra_step(prm, 3)
            with_hub = {"ef_search": ef, "hub_cache_ratio": 0.10, "cached_embeddings_MB": round(hubs.shape[0] * D * 4 / 1e6, 1),
                        "queries_per_s": round(B / e5, 3), "recall_at_10": r5,
                        "recomputed_chunks_per_query": round(st5["nunique"] / B, 1), "steps": 1}
        except Exception as ex:  # noqa: BLE001
            extras_errors["with_hub_cache"] = repr(ex)[:300]
        finally:
            try:
                idx.set_hub_cache(None)
            except Exception:  # noqa: BLE001
                pass
    if do_extras:  # two-level search (paper Alg. 2): prune_ratio 0.5, global strategy, PQ 48 B/vector
        try:
            from leann_amd.pq import encode_pq, train_pq

            t1 = time.time()
            cb = train_pq(X, 48, iters=8, seed=0)
            codes = encode_pq(X, cb)
            idx.attach_pq(cb.cpu().numpy(), codes.cpu().numpy())
            t_pq = time.time() - t1
            e4, r4, st4 = extra_step(idx.make_params(ef=ef, beam=args.beam, recompute=True, max_batch=B, prune_ratio=0.5), 2)
            two_level = {"ef_search": ef, "prune_ratio": 0.5, "pruning_strategy": "global", "pq_bytes": 48,
                         "queries_per_s": round(B / e4, 3), "recall_at_10": r4,
                         "recomputed_chunks_per_query": round(st4["nunique"] / B, 1), "adc_evals_per_query": round(st4["nadc"] / B, 1),
                         "pq_train_encode_s": round(t_pq, 1), "steps": 1}
        except Exception as ex:  # noqa: BLE001
            extras_errors["with_two_level_search"] = repr(ex)[:300]
    # ---- small-batch latency (B = 1, 16, 64, 256) and the parity check on this very index: untimed extras, rank 0 / N = 1 ----
    latency_rows = parity = None
    next_row = B * (K + W + 5)
    # The timed steps above already ran over the product default: the LIBRARY-side provider (csrc/lm_recompute.hip -- no interpreter
    # in the search loop, one host synchronisation per round).  The Python form of the provider is the A/B here.
    latency_python_provider = provider_ab = latency_speculate = None
    if world == 1 and not args.no_latency_rows:
        try:
            latency_rows, next_row = small_batch_latency(
                idx, Q, lambda b: idx.make_params(ef=ef, beam=args.beam, recompute=True, max_batch=b), recall, next_row)
            # option "speculate" (k_speculate: a one-query round also embeds the neighbours of its S best unexpanded candidates; same labels,
            # fewer forwards, more chunks): off by default, reported next to the default row
            latency_speculate = {}
            for S in (4, 16):
                idx.set_option("speculate", S)
                try:
                    rows_s, next_row = small_batch_latency(
                        idx, Q, lambda b: idx.make_params(ef=ef, beam=args.beam, recompute=True, max_batch=b), recall, next_row, batches=(1,), budget_s=4.0)
                    latency_speculate[str(S)] = rows_s[0] if rows_s else None
                finally:
                    idx.set_option("speculate", 0)
            if idx.native_provider:  # A/B in the same run: the same batch sizes over the Python provider
                os.environ["LEANN_MI355X_NATIVE_PROVIDER"] = "0"
                try:
                    idx.set_provider(provider)
                    assert not idx.native_provider
                    latency_python_provider, next_row = small_batch_latency(
                        idx, Q, lambda b: idx.make_params(ef=ef, beam=args.beam, recompute=True, max_batch=b), recall, next_row, batches=(1, 16, 256))
                finally:
                    os.environ.pop("LEANN_MI355X_NATIVE_PROVIDER", None)
                    idx.set_provider(provider)
        except Exception as ex:  # noqa: BLE001
            extras_errors["small_batch_latency"] = repr(ex)[:300]
    frontier = None
    if world == 1 and not args.no_latency_rows:
        try:
            frontier = latency_frontier(idx, Q, recall, next_row)
        except Exception as ex:  # noqa: BLE001
            extras_errors["small_batch_latency_frontier"] = repr(ex)[:300]
    if world == 1 and idx.native_provider and K and not args.no_provider_ab:  # one full-size step over the PYTHON provider, on a timed step's own queries: same labels
        os.environ["LEANN_MI355X_NATIVE_PROVIDER"] = "0"
        try:
            idx.set_provider(provider)
            lo_ = W * B
            idx.search_device(Q[:B], 10, prm)  # warm-up of that path
            torch.cuda.synchronize()
            t1_ = time.perf_counter()
            _, lx = idx.search_device(Q[lo_ : lo_ + B], 10, prm)
            torch.cuda.synchronize()
            provider_ab = {"queries_per_s_python_provider": round(B / (time.perf_counter() - t1_), 3), "steps": 1,
                           "labels_identical_to_the_timed_step": bool(torch.equal(lx, out_labels[0])),
                           "library_side_provider_stats_of_the_run": provider.native_stats()}
        except Exception as ex:  # noqa: BLE001
            extras_errors["provider_ab"] = repr(ex)[:300]
        finally:
            os.environ.pop("LEANN_MI355X_NATIVE_PROVIDER", None)
            idx.set_provider(provider)
    # ---- value by batch size, with and without the per-call memo (one step each, fresh queries): `value` is a B = 2048 / N = 1M figure --
    #      the memo's gain is a property of B / N (it is what cross-query dedup buys when a call re-embeds a large part of the corpus) ----
    value_by_batch = None
    chunks_per_s = (args.chunks / t_embed) if t_embed > 0 else None  # encoder throughput of this run (corpus embedding at index-build time)
    if world == 1 and not args.no_latency_rows:
        try:
            value_by_batch = []
            for b in (64, 256, 8192):
                row = {"batch": b}
                for memo in (True, False):
                    if next_row + b > Q.shape[0]:
                        next_row = B * (K + W + 5)  # re-use the latency rows' queries (values only; recall is not reported here)
                    pb = idx.make_params(ef=ef, beam=args.beam, recompute=True, max_batch=b, recompute_memo=memo)
                    qb = Q[next_row : next_row + b].contiguous() if next_row + b <= Q.shape[0] else Q_all[:b].contiguous()
                    next_row += b
                    torch.cuda.synchronize()
                    t1_ = time.perf_counter()
                    idx.search_device(qb, 10, pb)
                    torch.cuda.synchronize()
                    e_ = time.perf_counter() - t1_
                    st_ = idx.stats()
                    row["memo" if memo else "no_memo"] = {
                        "queries_per_s": round(b / e_, 2), "recomputed_chunks_per_query": round(st_["nunique"] / b, 1),
                        "fraction_of_corpus_reembedded": round(st_["nunique"] / args.chunks, 4)}
                if chunks_per_s:
                    row["brute_force_bound_queries_per_s"] = round(b * chunks_per_s / args.chunks, 2)
                value_by_batch.append(row)
        except Exception as ex:  # noqa: BLE001
            extras_errors["value_by_batch"] = repr(ex)[:300]
    extra_batch_rows = None
    if world == 1 and args.extra_batches:
        try:
            extra_batch_rows = []
            for b in (int(v) for v in args.extra_batches.split(",") if v):
                lo_ = B * (K + W + 5)  # the latency rows' block of fresh queries
                if lo_ + 2 * b > Q.shape[0]:
                    continue
                pb = idx.make_params(ef=ef, beam=args.beam, recompute=True, max_batch=b)
                idx.search_device(Q[lo_ + b : lo_ + 2 * b].contiguous(), 10, pb)  # warm-up of this batch size (workspace sizing), other queries
                qb = Q[lo_ : lo_ + b].contiguous()
                torch.cuda.synchronize()
                t1_ = time.perf_counter()
                _, lx = idx.search_device(qb, 10, pb)
                torch.cuda.synchronize()
                e_ = time.perf_counter() - t1_
                st_ = idx.stats()
                extra_batch_rows.append({"batch": b, "queries_per_s": round(b / e_, 3), "ms": round(1e3 * e_, 1), "recall_at_10": round(recall(lx.cpu().numpy(), range(lo_, lo_ + b)), 4),
                                         "recomputed_chunks_per_query": round(st_["nunique"] / b, 1), "rounds": int(st_["nrounds"]), "steps": 1,
                                         "note": "one call after one warm-up call of the same batch size on other queries"})
        except Exception as ex:  # noqa: BLE001
            extras_errors["extra_batch_rows"] = repr(ex)[:300]
    if world == 1 and not args.no_parity_check:
        try:
            t1 = time.time()
            parity = parity_check(idx, g, X, Q[min(next_row, n_q - 272):], provider, ef, args.beam, D)
            parity["seconds"] = round(time.time() - t1, 1)
        except Exception as ex:  # noqa: BLE001
            extras_errors["parity_check"] = repr(ex)[:300]
    labels_np = torch.cat(out_labels).cpu().numpy() if out_labels else np.zeros((0, 10), np.int64)
    rec = recall(labels_np, range(W * B, (W + K) * B)) if K else 0.0
    if world > 1:
        t = torch.tensor([rec], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        rec = float(t.item()) / world
    qps = world * K * B / elapsed if elapsed > 0 else 0.0

    # ---- roofline of the hand-written distance/beam-update kernel (HIP events, timed region) -------
    bytes_eval = D * 4 + 4  # D*s_e + 4 (id); the distance never goes back to HBM (fused)
    # Per-launch duration = the kernel's execution span measured ON the device (per-workgroup wall-clock stamps,
    # max end - min start): the same quantity rocprofv3 --kernel-trace --stats reports.  The HIP-event pair around
    # each launch (which also contains dispatch latency) is reported next to it.
    ev_over = idx.event_overhead_us()
    raw_us = 1e3 * prof["update_ms"] / max(prof["update_launches"], 1)
    net_us = 1e3 * prof["update_span_ms"] / max(prof["update_span_launches"], 1)
    upd_s = prof["update_span_ms"] * 1e-3
    achieved = prof["ndis"] * bytes_eval / upd_s / 1e9 if upd_s > 0 else 0.0
    traffic = None
    traffic_src = None
    try:  # HBM bytes per launch from the separate PMC pass (profiles/r1_pmc_k_update.json), scaled to this run's launch size
        pmc_file = next(f for f in ("r4_pmc_k_update.json", "r1_pmc_k_update.json") if (ROOT / "profiles" / f).exists())
        pmc = json.loads((ROOT / "profiles" / pmc_file).read_text())
        per_eval = pmc.get("hbm_bytes_per_eval") or pmc.get("hbm_bytes_per_eval_corrected_x1.36") or pmc.get("hbm_bytes_per_eval_corrected_x1.08")
        if per_eval:
            traffic = round(per_eval * prof["ndis"] / max(prof["update_launches"], 1))
            traffic_src = (f"rocprofv3 --pmc FETCH_SIZE pass on scripts/kernel_bench.py --provider (profiles/{pmc_file}), calibrated on a known byte count measured "
                           "in the same pass, scaled by evals/launch")
        else:
            traffic_src = f"profiles/{pmc_file} carries no hbm_bytes_per_eval key: traffic not reported"
    except Exception as ex:  # noqa: BLE001
        traffic_src = "no PMC file of the distance kernel could be read: " + repr(ex)[:120]
    roofline_dist = {"bound": "hbm", "kernel": "lm::k_update<6,false,false,1,256> (fused gather + distance + beam update, recompute mode)", "achieved": round(achieved, 2),
                "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 5), "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": round(bytes_eval * prof["ndis"] / max(prof["update_launches"], 1)),
                "bytes_per_eval": bytes_eval, "evals_per_launch": round(prof["ndis"] / max(prof["update_launches"], 1), 1),
                "us_per_launch": round(net_us, 2), "us_per_launch_event_pair": round(raw_us, 2), "empty_kernel_event_pair_us": round(ev_over, 2),
                "timing": "device wall-clock span per launch (HIP events alongside), from one profiled step after the timed ones",
                "profiled_step_ms": round(1e3 * prof_step_s, 1)}
    # encoder (MFMA bound): flops of the chunks actually recomputed / HIP-event time of the provider
    lens = np.diff(off.astype(np.int64))
    mean_flops = float(np.mean([cfg.flops_per_chunk(int(t)) for t in np.random.default_rng(0).choice(lens, 4096)]))
    enc_tf = prof["nunique"] * mean_flops / (prof["provider_ms"] * 1e-3) / 1e12 if prof["provider_ms"] > 0 else 0.0
    roofline_encoder = {"bound": "mfma", "achieved": round(enc_tf, 2), "peak": 2500.0, "unit": "TFLOP/s",
                        "frac": round(enc_tf / 2500.0, 5), "chunks_per_query": round(agg["nunique"] / max(K * B, 1), 1),
                        "mean_gflop_per_chunk": round(mean_flops / 1e9, 3), "provider_ms_share": round(prof["provider_ms"] / (prof_step_s * 1e3), 4),
                        "whole_step_TFLOPs": round(agg["nunique"] * mean_flops / max(elapsed, 1e-9) / 1e12, 2)}

    # `roofline` = the DOMINANT kernel of the timed region: the fused second half of an encoder layer (attention output projection,
    # LayerNorm, feed-forward block, LayerNorm: 70 % of the encoder's flops, the largest share of the step's time), MFMA bound;
    # duration = HIP event pairs around every one of its launches in the timed region (torch's current stream = the stream it is
    # launched on).  Algorithmic flops per token: 4 * ffn * hidden + 2 * hidden^2.
    kt_extras = kt_close("extras")
    kname = KT_NAMES[kt_dominant]
    if kt_dominant == _lib.KT_LAYER_TAIL:
        kdesc = ("lm::k_layer_tail_h384<0,2,4,1,0> (attention output projection + residual + LayerNorm + fc1 + GELU + fc2 + residual + LayerNorm in one kernel, "
                 "generation 4)")
        fpt = 4 * cfg.ffn * cfg.hidden + 2 * cfg.hidden * cfg.hidden
    else:  # hidden != 384: the general GEMM is the dominant kernel
        fpt = None
        kdesc = ("lm::k_gemm_f16<GemmShape<2,4,4,2>, *> (general 256 x 256-tile MFMA GEMM with bias / GELU / residual epilogues: the QKV, "
                 "attention-output and both feed-forward projections of every layer)")
    mlp = ktimes.get(kname)
    mlp_all = {"launches": 0, "ms": 0.0, "work": 0.0}
    for ph_ in kt_phase.values():
        for f_ in mlp_all:
            mlp_all[f_] += ph_.get(kname, {}).get(f_, 0)
    if mlp and mlp["ms"] > 0:
        mlp_tf = mlp["work"] / (mlp["ms"] * 1e-3) / 1e12
        tpl = mlp["work"] / fpt / max(mlp["launches"], 1) if fpt else None  # tokens per launch
        ktraffic = ktraffic_src = None
        try:  # HBM-side bytes per launch: the separate PMC passes of this kernel (profiles/r6_pmc_layer_tail.json: re-taken on round 6's final tree; round 4's as a fallback), per token x this run's launch size
            pmc_file_t = next(f for f in ("r6_pmc_layer_tail.json", "r4_pmc_layer_tail.json") if (ROOT / "profiles" / f).exists())
            pmc_t = json.loads((ROOT / "profiles" / pmc_file_t).read_text())
            if kt_dominant == _lib.KT_LAYER_TAIL:
                ktraffic = round(pmc_t["k_layer_tail_h384"]["hbm_bytes_per_token"] * tpl)
                ktraffic_src = pmc_t["k_layer_tail_h384"].get("how", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over scripts/kbench.cpp (profiles/r4_pmc_layer_tail.json), scaled by tokens/launch")
        except Exception:  # noqa: BLE001
            pass
        roofline = {"bound": "mfma", "kernel": kdesc,
                    "achieved": round(mlp_tf, 2), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(mlp_tf / 2500.0, 5), "traffic": ktraffic,
                    "traffic_source": ktraffic_src,
                    "flops_per_token": fpt, "launches": mlp["launches"],
                    "avg_launch_us": round(1e3 * mlp["ms"] / max(mlp["launches"], 1), 1),
                    "tokens_per_launch": round(tpl) if tpl else None,
                    "gflop_per_launch": round(mlp["work"] / max(mlp["launches"], 1) / 1e9, 2),
                    "share_of_timed_region": round(mlp["ms"] / (elapsed * 1e3), 4),
                    "all_launches_of_the_process": {"launches": mlp_all["launches"], "avg_launch_us": round(1e3 * mlp_all["ms"] / max(mlp_all["launches"], 1), 1),
                                                    "TFLOPs": round(mlp_all["work"] / max(mlp_all["ms"] * 1e-3, 1e-12) / 1e12, 2),
                                                    "note": "corpus embedding, warm-up, timed, profiled and extra steps together: the population a rocprofv3 --kernel-trace --stats table of this command averages"},
                    "timing": "library-side: HIP event pairs recorded by the library around every launch of this kernel in the timed region, on the launch stream "
                              "(csrc/lm_timing.cpp); the timed region is the product default path (library-side recompute provider, one-call forwards)",
                    "instrumentation": f"the timed region contains these {mlp['launches']} event pairs (2 records per launch of this kernel, nothing else)"}
        try:  # context, not the roofline's `peak`: what a register-only MFMA loop sustains on this part under its power cap (scripts/mfma_sustained.cpp, measured once)
            sus = json.loads((ROOT / "profiles" / "r6_mfma_sustained_register_only_ceiling_under_the_power_cap.jsonl").read_text().splitlines()[0])
            roofline["sustained_mfma_ceiling"] = {
                "TFLOPs": sus["TFLOPs"], "achieved_over_it": round(mlp_tf / sus["TFLOPs"], 4), "power_w": (sus.get("power_w") or {}).get("median"),
                "what": "v_mfma_f32_32x32x16_f16 back to back on every SIMD, operands (random fp16 bits) and accumulators in registers, nothing else, for seconds: the rate the "
                        "socket's power cap leaves of the 2.5 PFLOP/s figure `peak` quotes (profiles/r6_mfma_sustained_register_only_ceiling_under_the_power_cap.jsonl; another "
                        "box than this run's).  `frac` stays achieved / peak"}
        except Exception:  # noqa: BLE001
            pass
    else:  # encoder without the fused block (hidden != 384): fall back to the distance kernel's line
        roofline = roofline_dist
    result = {
        "metric": ("queries/sec at recall@10>=0.9, 1M-chunk HNSW, MiniLM-L6 recompute" if args.config == "c2" else
                   f"queries/sec at recall@10>=0.9, {args.chunks}-chunk HNSW, {args.model} fp16 recompute (BASELINE.json configs[4])"),
        "value": round(qps, 3), "unit": "queries/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": round(1e3 * elapsed / max(K, 1), 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "fp16", "dry_run": dry, "dtype_detail": "encoder (99.9 % of the arithmetic): fp16 MFMA with fp32 accumulation = the reference's CUDA precision (embedding_compute.py:157-162); distances, beam update and top-k: f32", "data": "synthetic" if not dry else "dry-run (emulated library on the CPU: control flow only, nothing here is a measurement)",
        "config": {"workload": f"{args.chunks} synthetic chunks (topic model, {'len~N(180,50)' if not args.fixed_len else 'every chunk ' + str(args.fixed_len) + ' tokens'}), HNSW M={args.M} GPU-built, "
                               f"{args.model} shape (random init), ef_search={ef}, beam={args.beam}, top-10, "
                               f"{B} queries/step/GPU, queries partitioned over {world} GPU(s), graph replicated; library-default search parameters "
                               f"(per-call recompute memo on: within one step no chunk is encoded twice; nothing is kept between steps)",
                   "baseline_config": args.config, "n_chunks": args.chunks, "ef_search": ef, "beam_width": args.beam, "queries_per_step": B * world,
                   "parallelism": f"queries-dp{world}", "rccl_ranks": world, "recompute_memo": 1,
                   "multi_gpu_path": "leann_amd.distributed.PartitionedSearch (index built on rank 0 and broadcast; per-step all_gather of the results)",
                   "workload_note": "topic-model corpus (1000 topics, 16 chunks per document), queries = held-out chunks of existing documents: recall@10 >= 0.9 is already "
                                    "reached at a smaller ef than BASELINE's 64 (ef_sweep, at_min_ef) -- ef=64 is over-provisioned for this corpus and `value` is the rate AT ef=64",
                   "timed_path": "library-side recompute provider (csrc/lm_recompute.hip), one-call forwards: the product default"},
        "recall_at_10": round(rec, 4),
        "without_call_memo": {
            "value": round(world * K2 * B / elapsed_nm, 3) if K2 and elapsed_nm > 0 else None, "unit": "queries/s", "steps": K2, "warmup": 1 if K2 else 0,
            "ms_per_step": round(1e3 * elapsed_nm / max(K2, 1), 3),
            "per_query": {"distance_evals": round(agg_nm["ndis"] / max(K2 * B, 1), 1), "recomputed_chunks": round(agg_nm["nunique"] / max(K2 * B, 1), 1),
                          "rounds_per_step": round(agg_nm["nrounds"] / max(K2, 1), 1)},
            "labels_identical_to_the_memo_steps": bool(same_labels),
            "what": "lm_search_params.recompute_memo = 0 on the first steps' own queries, same bracketing (barrier + synchronize, max over ranks): recomputed "
                    "embeddings are deduplicated within a lock-step round only and dropped after it -- the configuration rounds 1 and 2 reported as `value`"},
        "encoder_switches": {k: v for k, v in sorted(os.environ.items()) if k.startswith("LEANN_MI355X_")},  # kernels in effect
        "roofline": roofline, "roofline_distance_kernel": roofline_dist, "roofline_encoder": roofline_encoder,
        "ef_sweep": sweep,
        "per_query": {"distance_evals": round(agg["ndis"] / max(K * B, 1), 1), "recomputed_chunks": round(agg["nunique"] / max(K * B, 1), 1),
                      "rounds_per_step": round(agg["nrounds"] / max(K, 1), 1)},
        "setup_s": {"total": round(setup_s, 1), "embed_corpus": round(t_embed, 1), "build_graph": round(t_graph, 1)},
    }
    # the driver's record keeps `config` and `roofline` whole but truncates the tail of the line: the kernel-tracking figure (no per-call memo) goes there too
    result["config"]["without_call_memo_queries_per_s"] = result["without_call_memo"]["value"]
    if isinstance(result["roofline"], dict):
        result["roofline"]["value_without_call_memo_queries_per_s"] = result["without_call_memo"]["value"]
        result["roofline"]["whole_encoder_TFLOPs"] = roofline_encoder["achieved"]
        if probe is not None:
            try:  # the three kernels of a hidden-384 layer in the profiled step, scaled to kbench's 262,144-token reference size (every layer's
                # launches process the same tokens: the tail's flop count gives tokens x layers)
                tl = kprof.get("lm::k_layer_tail_h384", {})
                tok_layers = tl.get("work", 0.0) / fpt if fpt else 0.0
                if tok_layers > 0:
                    probe["profiled_step_us_per_262144_tokens"] = {
                        k_.replace("lm::", ""): round(1e3 * v_["ms"] / (tok_layers / 262144.0), 1)
                        for k_, v_ in kprof.items() if v_["launches"] and k_ in ("lm::k_layer_tail_h384", "lm::k_qkv_h384", "lm::k_attn_varlen", "lm::k_qkv_attn_h384")}
                    probe["profiled_step_tokens_x_layers"] = round(tok_layers)
            except Exception as ex:  # noqa: BLE001
                probe["profiled_step_error"] = repr(ex)[:200]
            result["roofline"]["box_probe"] = probe
    if min_ef:
        result["at_min_ef"] = min_ef
    if with_hub:
        result["with_hub_cache"] = with_hub
    if two_level:
        result["with_two_level_search"] = two_level
    if latency_rows:
        result["small_batch_latency"] = latency_rows
        result["small_batch_latency_provider"] = ("library-side provider (csrc/lm_recompute.hip): no interpreter in the search loop, one host "
                                                  "synchronisation per round" if latency_python_provider is not None else "Python provider")
    if frontier:
        result["small_batch_latency_frontier"] = frontier
        if isinstance(result["roofline"], dict) and frontier.get("best_at_recall_0.9"):  # (the driver's record keeps `roofline` whole)
            result["roofline"]["b1_best_at_recall_0.9"] = frontier["best_at_recall_0.9"]
    if latency_python_provider:
        result["small_batch_latency_python_provider"] = latency_python_provider
    if latency_speculate and any(latency_speculate.values()):
        result["small_batch_latency_b1_with_speculative_prefetch"] = latency_speculate
    if provider_ab:
        result["full_step_over_the_python_provider"] = provider_ab
    if value_by_batch:
        result["value_by_batch"] = {"rows": value_by_batch, "timed_batch": B, "encoder_chunks_per_s": round(chunks_per_s, 1) if chunks_per_s else None,
                                    "what": "one search call per row on fresh queries, library defaults except the memo switch; brute_force_bound = B x (encoder chunks/s) / N: "
                                            "the rate of embedding the WHOLE corpus once per call and scanning it (no graph) -- the per-call memo's gain over `no_memo` is a "
                                            "property of B / N (fraction_of_corpus_reembedded), ~0 at B = 1"}
    if kprof:
        tot_ms = sum(v["ms"] for v in kprof.values())
        result["encoder_kernels_profiled_step"] = {
            k_: {"launches": v["launches"], "ms": round(v["ms"], 1), "share_of_step": round(v["ms"] / (prof_step_s * 1e3), 4),
                 "TFLOPs": round(v["work"] / (v["ms"] * 1e-3) / 1e12, 1) if v["work"] and v["ms"] else None}
            for k_, v in kprof.items() if v["launches"]}
        result["encoder_kernels_profiled_step"]["instrumented_ms_of_step_ms"] = [round(tot_ms, 1), round(prof_step_s * 1e3, 1)]
    if extra_batch_rows:
        result["extra_batch_rows"] = extra_batch_rows
    if parity:
        result["parity_check"] = parity
    if table_roof:
        result["roofline_table_mode"] = table_roof
    if extras_errors:
        result["extras_errors"] = extras_errors

    if rank == 0:  # the line so far, on stderr: a run cut short inside the CPU baseline still leaves its GPU numbers in the log
        log("preliminary (before the CPU baseline): " + json.dumps({k: result[k] for k in ("value", "ms_per_step", "recall_at_10", "roofline", "roofline_encoder")}))
    # ---- CPU baseline (rank 0, N=1 only): the oracle + fp32 CPU encoder on a bounded sample ---------
    if world == 1 and not args.no_cpu_baseline:
        try:
            result["cpu_baseline"] = cpu_baseline(args, g, Q, tok, off, cfg, ef, args.beam, X)
        except Exception as ex:  # noqa: BLE001 - report, never lose the line
            result["cpu_baseline"] = {"value": None, "unit": "queries/s", "cores": 0, "kind": "port", "sample": "failed: " + repr(ex)[:200]}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


# ---- box probe: what box did this line run on? --------------------------------------------------------------------
# Two rounds of driver lines swung by +-8 % with nothing in them to tell a slow box from a slow build.  Three things go into
# roofline.box_probe: (i) sclk / mclk / socket power sampled WHILE the timed steps run (a thread reading sysfs, or rocm-smi where sysfs has
# no such file); (ii) before the timed steps, on an idle chip: 10 launches of the dominant kernel at kbench's reference size (262,107 tokens:
# 675-690 us on the boxes DESIGN 6.1 calls fast, 730+ on the slow ones) and the rate of a 1 GiB device-to-device copy; (iii) when (ii) reads
# slow, an L2 / fabric counter pass (rocprofv3 --pmc, kernel trace only) over the same launches in a child process, next to the fast-box
# reference kept under profiles/.
TAIL_PROBE_TOKENS = 262107
TAIL_PROBE_SLOW_US = 750.0


def _sysfs_cards():
    import glob

    return sorted(d for d in glob.glob("/sys/class/drm/card[0-9]*/device") if os.path.exists(os.path.join(d, "pp_dpm_sclk")))


def _sysfs_card_of_device(index):
    """The sysfs card of HIP device `index`.  A box shows every GPU of the node under /sys/class/drm while the container sees one of them: match the
    PCI address (round 6's third survey line sampled an idle neighbour: 158 MHz, 252 W); where that fails, the card that draws the most power."""
    cards = _sysfs_cards()
    if not cards:
        return None
    try:
        import torch

        pr = torch.cuda.get_device_properties(index)
        want = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
        for c in cards:
            if os.path.basename(os.path.realpath(c)).lower().startswith(want):
                return c
    except Exception:  # noqa: BLE001
        pass
    if len(cards) == 1:
        return cards[0]
    best = max(cards, key=lambda c: (_sysfs_sample(c).get("power_w") or 0.0))
    return best


def _sysfs_sample(card):
    """{"sclk_mhz", "mclk_mhz", "power_w"} of one card from sysfs (None where a file is missing)"""
    import glob
    import re

    def cur(name):
        try:
            for ln in open(os.path.join(card, name)):
                if "*" in ln:
                    m = re.search(r"(\d+)\s*[Mm][Hh]z", ln)
                    return int(m.group(1)) if m else None
        except OSError:
            pass
        return None

    pw = None
    for f in glob.glob(os.path.join(card, "hwmon", "hwmon*", "power1_average")) + glob.glob(os.path.join(card, "hwmon", "hwmon*", "power1_input")):
        try:
            pw = int(open(f).read().strip()) / 1e6
            break
        except (OSError, ValueError):
            continue
    return {"sclk_mhz": cur("pp_dpm_sclk"), "mclk_mhz": cur("pp_dpm_mclk"), "power_w": pw}


def _rocm_smi_sample():
    import re
    import subprocess

    try:
        txt = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
    except Exception:  # noqa: BLE001
        return None
    g = lambda pat: (lambda m: float(m.group(1)) if m else None)(re.search(pat, txt))  # noqa: E731
    return {"sclk_mhz": g(r"GPU\[0\]\s*:\s*sclk clock level:[^(]*\((\d+)Mhz\)"), "mclk_mhz": g(r"GPU\[0\]\s*:\s*mclk clock level:[^(]*\((\d+)Mhz\)"),
            "power_w": g(r"GPU\[0\]\s*:\s*Current Socket Graphics Package Power \(W\):\s*([\d.]+)")}


class BoxSampler:
    """Samples clocks and power of GPU `index` in a thread between start() and stop().  sysfs first (no process per sample); rocm-smi (one
    process per sample, every 5 s) where sysfs gives no clock."""

    def __init__(self, index=0, period_s=1.0):
        import threading

        self.card = _sysfs_card_of_device(index)
        self.source = "sysfs" if self.card and _sysfs_sample(self.card)["sclk_mhz"] is not None else "rocm-smi"
        self.period = period_s if self.source == "sysfs" else 5.0
        self.samples = []
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True)

    def _one(self):
        return _sysfs_sample(self.card) if self.source == "sysfs" else _rocm_smi_sample()

    def _run(self):
        while not self._stop.is_set():
            try:
                v = self._one()
                if v:
                    self.samples.append(v)
            except Exception:  # noqa: BLE001
                pass
            self._stop.wait(self.period)

    def start(self):
        self._th.start()
        return self

    def stop(self):
        self._stop.set()
        self._th.join(timeout=30)
        out = {"source": self.source, "n_samples": len(self.samples), "card": os.path.basename(os.path.realpath(self.card)) if self.card else None}
        for k in ("sclk_mhz", "mclk_mhz", "power_w"):
            v = [x[k] for x in self.samples if x.get(k) is not None]
            if v:
                out[k] = {"min": round(min(v), 1), "median": round(float(np.median(v)), 1), "max": round(max(v), 1)}
        return out


def _tail_probe_launches(enc, dev, tokens=TAIL_PROBE_TOKENS, launches=10, warm=60):
    """`launches` timed launches of the fused layer tail (layer 0 of `enc`, random activations) at kbench's reference size; HIP events on the
    launch stream.  Returns the per-launch times in us.  60 warm-up launches (~45 ms) first: the clock of an idle chip (sclk 95 MHz) takes tens
    of milliseconds to come up -- round 6's first probe warmed up with 3 launches and read 788-950 us on a box where kbench read 700."""
    import torch

    from leann_amd.encoder import fused_attn_out_mlp

    g = torch.Generator(device="cpu").manual_seed(11)
    a = (torch.randn((tokens, 384), generator=g) * 0.5).to(dev, torch.float16)
    r = (torch.randn((tokens, 384), generator=g) * 0.5).to(dev, torch.float16)
    layer = enc.layers[0]
    for _ in range(warm):
        if fused_attn_out_mlp(a, r, layer) is None:
            return None
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
    for e0, e1 in ev:
        e0.record()
        fused_attn_out_mlp(a, r, layer)
        e1.record()
    torch.cuda.synchronize()
    return [1e3 * e0.elapsed_time(e1) for e0, e1 in ev]


def _copy_rate(dev, gib=1, reps=5):
    import torch

    n = gib << 30
    src = torch.empty(n, dtype=torch.uint8, device=dev).fill_(3)
    dst = torch.empty_like(src)
    dst.copy_(src)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * n * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9  # read + written bytes per second, GB/s


def _box_probe_only():
    """What the probe's rocprofv3 child runs: the reference launches of the dominant kernel and nothing else."""
    import torch

    from leann_amd import _lib
    from leann_amd.encoder import BertEncoder

    _lib.require_gpu()
    dev = torch.device("cuda", 0)
    enc = BertEncoder.load("sentence-transformers/all-MiniLM-L6-v2", allow_random=True).to(dev, dtype=torch.float16).eval()
    us = _tail_probe_launches(enc, dev)
    print(json.dumps({"tail_probe_us": [round(x, 1) for x in (us or [])]}), flush=True)


def _box_survey():
    import torch

    from leann_amd import _lib
    from leann_amd.encoder import BertEncoder

    _lib.require_gpu()
    dev = torch.device("cuda", 0)
    enc = BertEncoder.load("sentence-transformers/all-MiniLM-L6-v2", allow_random=True).to(dev, dtype=torch.float16).eval()
    os.environ["BENCH_FORCE_TCC_PASS"] = "1"
    out = box_probe(enc, dev, 0)
    try:
        out["hostname"] = os.uname().nodename
        import subprocess

        out["rocm_smi_serial"] = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showserial", "--showuniqueid"], capture_output=True, text=True, timeout=20).stdout.strip()[-400:]
    except Exception:  # noqa: BLE001
        pass
    print(json.dumps(out), flush=True)


TCC_PASS_COUNTERS = ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_LEVEL_sum", "GRBM_GUI_ACTIVE"]


def _tcc_pass(timeout_s=240):
    """L2 hit / miss, fabric read requests and their summed occupancy (LEVEL / RDREQ = mean read latency in L2 cycles) of the probe's launches:
    rocprofv3 --pmc with the kernel trace only, over `bench.py --box-probe-only` in a child process.  Per-launch means of k_layer_tail_h384."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    out = tempfile.mkdtemp(prefix="lm_tcc_", dir="/tmp")
    try:
        env = dict(os.environ, TMPDIR="/tmp")
        cmd = ["rocprofv3", "--pmc", *TCC_PASS_COUNTERS, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "tcc", "--",
               sys.executable, os.path.abspath(__file__), "--box-probe-only"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd="/tmp", env=env)
        f = glob.glob(out + "/**/*counter_collection.csv", recursive=True)
        if not f:
            return {"error": f"no counter file (rc {r.returncode}): {r.stderr[-300:]}"}
        acc, n = {}, {}
        for row in csv.DictReader(open(f[0])):
            if "k_layer_tail_h384" not in row["Kernel_Name"]:
                continue
            c = row["Counter_Name"]
            acc[c] = acc.get(c, 0.0) + float(row["Counter_Value"])
            n[c] = n.get(c, 0) + 1
        res = {c: round(acc[c] / n[c]) for c in acc}
        res["launches"] = max(n.values()) if n else 0
        if res.get("TCC_HIT_sum") is not None and res.get("TCC_MISS_sum") is not None:
            res["l2_hit_rate"] = round(res["TCC_HIT_sum"] / max(res["TCC_HIT_sum"] + res["TCC_MISS_sum"], 1), 4)
        if res.get("TCC_EA0_RDREQ_sum"):
            res["mean_fabric_read_latency_l2_cycles"] = round(res.get("TCC_EA0_RDREQ_LEVEL_sum", 0) / res["TCC_EA0_RDREQ_sum"], 1)
        m = None
        for ln in r.stdout.splitlines():
            if ln.startswith('{"tail_probe_us"'):
                m = json.loads(ln)["tail_probe_us"]
        res["tail_probe_us_under_the_counters"] = m
        return res
    except Exception as ex:  # noqa: BLE001
        return {"error": repr(ex)[:300]}
    finally:
        shutil.rmtree(out, ignore_errors=True)


def box_probe(enc, dev, local_rank=0, counter_pass=True):
    """(ii) and (iii) above.  Never raises.  counter_pass=False (every rank but 0 of a multi-GPU run): no rocprofv3 child process."""
    out = {"what": f"before the timed steps, idle chip: 10 launches of lm_layer_tail_h384_f16 at {TAIL_PROBE_TOKENS} tokens (kbench tail4's size: 675-690 us on the "
                   f"boxes DESIGN 6.1 calls fast, 730+ on the slow ones) and a 1 GiB device-to-device copy; clocks / power sampled during the timed steps"}
    try:
        _tail_probe_launches(enc, dev, launches=2)  # (allocations, weight images, first clock ramp: outside the sampled window)
        smp = BoxSampler(local_rank, period_s=0.004).start()
        us = _tail_probe_launches(enc, dev)
        out["clocks_during_the_probe_launches"] = smp.stop()
        if us:
            out["layer_tail_262107_tokens_us"] = {"median": round(float(np.median(us)), 1), "min": round(min(us), 1), "max": round(max(us), 1), "launches": len(us)}
            out["layer_tail_TFLOPs"] = round(TAIL_PROBE_TOKENS * (4 * 1536 * 384 + 2 * 384 * 384) / (float(np.median(us)) * 1e-6) / 1e12, 1)
            out["box_class"] = "slow" if float(np.median(us)) > TAIL_PROBE_SLOW_US else "fast"
        out["copy_1GiB_GBps_read_plus_written"] = round(_copy_rate(dev), 1)
        ref = ROOT / "profiles" / "r6_box_probe_tcc_pass_fast_box.json"
        if ref.exists():
            out["tcc_pass_fast_box_reference"] = json.loads(ref.read_text()).get("tcc_pass")
        if counter_pass and (os.environ.get("BENCH_FORCE_TCC_PASS") == "1" or (us and float(np.median(us)) > TAIL_PROBE_SLOW_US and os.environ.get("BENCH_NO_TCC_PASS") != "1")):
            out["tcc_pass"] = _tcc_pass()
            out["tcc_pass_counters"] = ("rocprofv3 --pmc " + " ".join(TCC_PASS_COUNTERS) + " --kernel-trace -- python bench.py --box-probe-only (child process, "
                                        "per-launch means of k_layer_tail_h384; LEVEL / RDREQ = mean fabric read latency in L2 cycles)")
    except Exception as ex:  # noqa: BLE001
        out["error"] = repr(ex)[:300]
    return out


def parity_check(idx, g, X, Q, provider, ef, beam, dim, n_table=256, n_recompute=16):
    """GPU path vs the CPU oracle (the checker, untimed) ON THE BENCHMARK'S OWN INDEX AND QUERIES:
      * stored-embedding mode, n_table queries: labels, distances AND the number of distance evaluations must be identical
        (one-launch persistent kernel and lock-step rounds); for beam 1 the independent heap-based faiss transcription
        (oracle/lm_oracle_faiss.c) must return the same labels / distances as well;
      * recompute mode (library default: per-call memo), n_recompute queries: the oracle (memo restated in oracle/oracle.py) replays the GPU
        encoder's own per-round outputs (its provider must be asked for exactly the same sorted unique ids, round by round)
        -> labels and distances identical."""
    import torch

    from leann_amd.devmem import as_tensor
    from oracle import oracle as orc

    og = orc.OracleGraph(g.node_offsets, g.level_ptr, g.neighbors, g.levels, g.entry_point, g.max_level, g.metric_type, dim)
    Xn = X.cpu().numpy()
    qn = Q[:n_table].cpu().numpy()
    oi, od, ost = orc.search(og, qn, 10, ef=ef, beam=beam, table=Xn)
    out = {"n": int(qn.shape[0]), "config": f"stored embeddings, N={Xn.shape[0]}, ef={ef}, beam={beam}, top-10"}
    ids_exact, max_abs, ndis_ok = True, 0.0, True
    for persistent in (1, 0):
        idx.set_option("persistent_table", persistent)
        d, l = idx.search_device(Q[:n_table].contiguous(), 10, idx.make_params(ef=ef, beam=beam, recompute=False))
        st = idx.stats()
        ids_exact &= bool(np.array_equal(l.cpu().numpy(), oi))
        max_abs = max(max_abs, float(np.abs(d.cpu().numpy() - od).max()))
        ndis_ok &= int(st["ndis"]) == int(ost["ndis"])
    idx.set_option("persistent_table", 1)
    out.update({"ids_exact": ids_exact, "max_abs_dist": max_abs, "ndis_equal": ndis_ok})
    if beam == 1:
        fi, fd, _ = orc.faiss_search(og, qn, 10, ef=ef, table=Xn)
        out["faiss_transcription_agrees"] = bool(np.array_equal(fi, oi) and np.array_equal(fd, od))
    # recompute mode: replay
    rounds = []

    def recording(d_ids, cnt, stream):
        ptr = provider(d_ids, cnt, stream)
        torch.cuda.synchronize()
        rounds.append((as_tensor(d_ids, (cnt,), "int32").cpu().numpy().copy(),
                       as_tensor(ptr, (cnt, provider.dp), "float32").cpu().numpy()[:, :dim].copy()))
        return ptr

    idx.set_provider(recording)
    qr = Q[n_table : n_table + n_recompute].contiguous()
    d, l = idx.search_device(qr, 10, idx.make_params(ef=ef, beam=beam, recompute=True))
    torch.cuda.synchronize()
    idx.set_provider(provider)
    it = iter(rounds)
    asked_same = [True]

    def replay(idv):
        ids, emb = next(it)
        asked_same[0] &= bool(np.array_equal(ids, idv))
        return emb

    try:
        ri, rd, _ = orc.search(og, qr.cpu().numpy(), 10, ef=ef, beam=beam, provider=replay, memo=True)  # the library default (per-call memo)
        out["recompute"] = {"n": int(qr.shape[0]), "rounds": len(rounds), "same_ids_requested_every_round": asked_same[0],
                            "ids_exact": bool(np.array_equal(l.cpu().numpy(), ri)),
                            "max_abs_dist": float(np.abs(d.cpu().numpy() - rd).max())}
    except (StopIteration, AssertionError) as ex:  # the oracle asked for more / different rounds than the GPU ran
        out["recompute"] = {"n": int(qr.shape[0]), "rounds": len(rounds), "same_ids_requested_every_round": False, "ids_exact": False,
                            "error": repr(ex)[:200]}
    if getattr(idx, "native_provider", False):  # the same queries over the library-side provider (the product default): identical output
        d2, l2 = idx.search_device(qr, 10, idx.make_params(ef=ef, beam=beam, recompute=True))
        torch.cuda.synchronize()
        out["recompute"]["library_side_provider_identical"] = bool(torch.equal(l2, l) and torch.equal(d2, d))
    return out


def small_batch_latency(idx, Q, prm_of, recall, first_row, batches=(1, 16, 64, 256), budget_s=25.0):
    """LEANN's real call is one query at a time (leann/api.py:644-796 consumes labels[0]; the reference reports
    0.818 s per query for this configuration, docs/configuration-guide.md:357-364).  Per batch size: repeated searches on
    fresh queries, synchronised after each; p50 / mean latency and the resulting queries per second."""
    import torch

    rows, lo = [], first_row
    for b in batches:
        prm = prm_of(b)
        idx.search_device(Q[lo : lo + b].contiguous(), 10, prm)  # warm-up (workspace sizing for this batch size)
        lo += b
        lat, labels, used, chunks, rounds = [], [], [], 0, 0
        t_all = time.perf_counter()
        reps = 0
        while reps < (24 if b == 1 else 8) and time.perf_counter() - t_all < budget_s / len(batches) and lo + b <= Q.shape[0]:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, l = idx.search_device(Q[lo : lo + b].contiguous(), 10, prm)
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t0)
            st_ = idx.stats()  # (of that call)
            chunks += int(st_["nunique"])
            rounds += int(st_["nrounds"])
            labels.append(l)
            used.append(range(lo, lo + b))
            lo += b
            reps += 1
        if not lat:
            continue
        lat_ms = np.array(lat) * 1e3
        rec = float(np.mean([recall(l.cpu().numpy(), r) for l, r in zip(labels, used)]))
        rows.append({"batch": b, "reps": len(lat), "p50_ms": round(float(np.median(lat_ms)), 2), "mean_ms": round(float(lat_ms.mean()), 2),
                     "max_ms": round(float(lat_ms.max()), 2), "queries_per_s": round(b / float(lat_ms.mean()) * 1e3, 2),
                     "recall_at_10": round(rec, 4), "recomputed_chunks_per_query": round(chunks / (len(lat) * b), 1),
                     "rounds_per_call": round(rounds / len(lat), 1)})
    return rows, lo


def latency_frontier(idx, Q, recall, first_row, efs=(16, 32, 64), beams=(1, 4), batch_sizes=(0, 32, 64, 128, 256), reps=16, b256=True, budget_s=75.0):
    """Latency vs recall at B = 1 over the three knobs the reference's search call carries (hnsw_backend.py:203-234): complexity (efSearch),
    beam_width (pops per round) and batch_size (dynamic batching, paper section 4.2: k_expand keeps popping while a round's new-list is
    shorter).  Every cell: `reps` one-query calls on fresh queries (the SAME queries in every cell), p50 latency, recall@10 against the exact
    top-10, rounds and recomputed chunks per call.  `best_at_recall_0.9` = the fastest cell with recall >= 0.9.  B = 256 (one call per
    setting) for the throughput side of the same knob."""
    import torch

    cells, t_all = [], time.perf_counter()
    lo = first_row
    if lo + reps + 1 + 512 > Q.shape[0]:
        lo = max(0, Q.shape[0] - (reps + 1 + 512))
    qs = [Q[lo + 1 + i : lo + 2 + i].contiguous() for i in range(reps)]
    rows = [range(lo + 1 + i, lo + 2 + i) for i in range(reps)]
    for ef in efs:
        for beam in beams:
            for bs in batch_sizes:
                if time.perf_counter() - t_all > budget_s:
                    break
                prm = idx.make_params(ef=ef, beam=beam, recompute=True, max_batch=1, batch_size=bs)
                idx.search_device(Q[lo : lo + 1].contiguous(), 10, prm)  # warm-up of this setting (workspace sizing)
                lat, rec, rounds, chunks, ndis = [], [], 0, 0, 0
                for q1, r1 in zip(qs, rows):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    _, l = idx.search_device(q1, 10, prm)
                    torch.cuda.synchronize()
                    lat.append((time.perf_counter() - t0) * 1e3)
                    st_ = idx.stats()
                    rounds += int(st_["nrounds"])
                    chunks += int(st_["nunique"])
                    ndis += int(st_["ndis"])
                    rec.append(recall(l.cpu().numpy(), r1))
                cells.append({"ef": ef, "beam": beam, "batch_size": bs, "p50_ms": round(float(np.median(lat)), 2), "mean_ms": round(float(np.mean(lat)), 2),
                              "recall_at_10": round(float(np.mean(rec)), 4), "rounds_per_call": round(rounds / reps, 1), "recomputed_chunks_per_call": round(chunks / reps, 1),
                              "distance_evals_per_call": round(ndis / reps, 1)})
    ok = [c for c in cells if c["recall_at_10"] >= 0.9]
    out = {"batch": 1, "reps_per_cell": reps, "cells": cells, "best_at_recall_0.9": min(ok, key=lambda c: c["p50_ms"]) if ok else None,
           "what": "one-query calls (the real caller's batch, api.py:644-796) over efSearch x beam_width x batch_size; the same queries in every cell; recall against the exact top-10"}
    if b256:
        out["batch_256"] = []
        qb = Q[lo + reps + 1 : lo + reps + 1 + 256].contiguous()
        rb = range(lo + reps + 1, lo + reps + 1 + 256)
        for ef, beam, bs in ((64, 1, 0), (64, 1, 64), (64, 4, 0), (32, 1, 64), (16, 1, 0), (16, 1, 64)):
            if time.perf_counter() - t_all > 1.5 * budget_s:
                break
            prm = idx.make_params(ef=ef, beam=beam, recompute=True, max_batch=256, batch_size=bs)
            idx.search_device(Q[lo + reps + 257 : lo + reps + 513].contiguous(), 10, prm)  # warm-up on other queries
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, l = idx.search_device(qb, 10, prm)
            torch.cuda.synchronize()
            e_ = time.perf_counter() - t0
            st_ = idx.stats()
            out["batch_256"].append({"ef": ef, "beam": beam, "batch_size": bs, "ms": round(1e3 * e_, 1), "queries_per_s": round(256 / e_, 1),
                                     "recall_at_10": round(recall(l.cpu().numpy(), rb), 4), "rounds": int(st_["nrounds"]),
                                     "recomputed_chunks_per_query": round(st_["nunique"] / 256, 1)})
    return out


def cpu_traversal_only(orc, og, Q, X, ef, beam, ncores):
    """SURVEY 8(d), first CPU figure: the oracle's traversal + distances alone over STORED embeddings (what faiss does once the
    embeddings exist), 256 queries on the host cores.  Never raises: an optional figure may not cost the line."""
    try:
        nq = int(min(256, Q.shape[0]))
        xt, qt = X.cpu().numpy(), Q[:nq].cpu().numpy()
        orc.search(og, qt[:8], 10, ef=ef, beam=beam, table=xt)  # page the table in
        t0 = time.perf_counter()
        _, _, st = orc.search(og, qt, 10, ef=ef, beam=beam, table=xt)
        el = time.perf_counter() - t0
        return {"value": round(nq / el, 2), "unit": "queries/s", "cores": ncores, "kind": "port",
                "sample": f"{nq} queries, stored embeddings (no encoder): {st['ndis'] / nq:.0f} distance evaluations per query in {el:.2f}s"}
    except Exception as ex:  # noqa: BLE001
        return {"value": None, "unit": "queries/s", "sample": "failed: " + repr(ex)[:200]}


def cpu_baseline(args, g, Q, tok, off, cfg, ef, beam, X=None):
    """The reference path on the host cores: oracle traversal (oracle/lm_oracle.c, kind "port") with
    embeddings recomputed per round by the SAME encoder in fp32 on the CPU
    (embedding_compute.py:148-154 uses fp32 on CPU).  Bounded to ~cpu_baseline_seconds.  With ``X`` (the corpus embeddings) the
    traversal-only figure of SURVEY 8(d) is reported next to it (``traversal_only``)."""
    import torch

    from leann_amd.encoder import BertEncoder
    from oracle import oracle as orc

    ncores = orc.usable_cores()  # affinity / cgroup aware (the GPU box exposes 256 threads, 16 usable)
    orc.set_num_threads(ncores)
    torch.set_num_threads(ncores)
    enc = BertEncoder.load(args.model, allow_random=True).float().eval()
    og = orc.OracleGraph(g.node_offsets, g.level_ptr, g.neighbors, g.levels, g.entry_point, g.max_level, g.metric_type, cfg.hidden)
    lens_all = np.diff(off.astype(np.int64))
    T = int(lens_all.max())
    stat = {"chunks": 0, "enc_s": 0.0}
    trav = cpu_traversal_only(orc, og, Q, X, ef, beam, ncores) if X is not None else None

    def provider(idv):
        t0 = time.perf_counter()
        n = idv.shape[0]
        ln = lens_all[idv].astype(np.int32)
        order = np.argsort(ln, kind="stable")  # length-sorted mini-batches padded to their own maximum, as a careful CPU
        e = np.empty((n, cfg.hidden), np.float32)  # implementation would (the reference pads a batch to its longest member)
        with torch.no_grad():
            for b0 in range(0, n, 64):
                sel = order[b0 : b0 + 64]
                Tb = int(ln[sel].max())
                ids = np.zeros((sel.shape[0], Tb), np.int32)
                for i, j in enumerate(sel):
                    b = int(off[idv[j]])
                    ids[i, : ln[j]] = tok[b : b + ln[j]]
                e[sel] = enc.encode_tokens(torch.from_numpy(ids), torch.from_numpy(ln[sel]), batch_size=64).numpy()
        stat["chunks"] += n
        stat["enc_s"] += time.perf_counter() - t0
        return e

    q = Q[:64].cpu().numpy()
    # calibrate on one query, then size the sample
    t0 = time.perf_counter()
    orc.search(og, q[:1], 10, ef=ef, beam=beam, provider=provider)
    one = time.perf_counter() - t0
    # sample size: what the time budget buys, but at least 16 queries as long as those cost no more than 6x the budget (C2: 16
    # queries = ~80 s); a model whose single query already exceeds the budget (bge-base fp32 on 16 cores: ~40 s per query) is
    # reported from the calibration query alone -- the baseline may never cost the bench line (a 500k-chunk C5 run was lost to it)
    budget = args.cpu_baseline_seconds
    nq = int(min(63, budget // max(one, 1e-3)))
    if nq < 16 and 16 * one <= 6 * budget:
        nq = 16
    if nq < 1:
        return {"value": round(1.0 / one, 5), "unit": "queries/s", "cores": ncores, "kind": "port",
                "sample": f"1 query (the calibration query: {one:.1f}s exceeds the {budget:.0f}s budget), oracle traversal + fp32 CPU encoder; "
                          f"{stat['chunks']} chunks recomputed in {stat['enc_s']:.1f}s",
                "chunks_per_s_encoder": round(stat["chunks"] / max(stat["enc_s"], 1e-9), 1), "traversal_only": trav}
    stat = {"chunks": 0, "enc_s": 0.0}
    t0 = time.perf_counter()
    _, _, st = orc.search(og, q[1 : 1 + nq], 10, ef=ef, beam=beam, provider=provider, memo=True)  # same per-call memo as the GPU default
    el = time.perf_counter() - t0
    return {"value": round(nq / el, 5), "unit": "queries/s", "cores": ncores, "kind": "port",
            "sample": f"{nq} queries (batched lock-step, same graph/ef/beam, per-call memo as on the GPU), oracle traversal + fp32 CPU encoder; "
                      f"{st['nunique']} chunks recomputed in {stat['enc_s']:.1f}s of {el:.1f}s",
            "chunks_per_s_encoder": round(stat["chunks"] / max(stat["enc_s"], 1e-9), 1), "traversal_only": trav}


if __name__ == "__main__":
    main()
