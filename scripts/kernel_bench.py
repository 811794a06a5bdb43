#!/usr/bin/env python
"""Micro-benchmark of the hand-written kernels on an HBM-sized working set (no encoder, no graph build):
  * lm_dist_gather on UNIQUE random rows  -> known byte count, used to calibrate rocprofv3 FETCH_SIZE
  * stored-embedding search (k_expand + k_update) on a random regular graph, large batch
Run plain for HIP-event timings, or under `rocprofv3 --pmc FETCH_SIZE ...` (scripts/pmc_kernels.sh).
"""
import argparse
import ctypes as C
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import numpy as np
import torch

from leann_amd import _lib
from leann_amd.csr_format import HnswCsr
from leann_amd.index import Mi355xIndex

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--d", type=int, default=384)
ap.add_argument("--deg", type=int, default=32)
ap.add_argument("--batch", type=int, default=8192)
ap.add_argument("--ef", type=int, default=64)
ap.add_argument("--beam", type=int, default=4)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--variant", type=int, default=0)
ap.add_argument("--provider", action="store_true", help="recompute mode: embeddings served by a provider (index_select of the table)")
ap.add_argument("--memo", action="store_true")
ap.add_argument("--wave", type=int, default=-1, help="persistent kernel: 1 = wave per query, 0 = workgroup per query, -1 = auto")
ap.add_argument("--lockstep", action="store_true", help="stored-embedding mode with lock-step rounds instead of the persistent kernel")
args = ap.parse_args()

_lib.require_gpu()
lib = _lib.load()
dev = torch.device("cuda")
N, D = args.n, args.d
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.randn((N, D), generator=g, device=dev)
X /= X.norm(dim=1, keepdim=True)
Q = torch.randn((args.batch, D), generator=g, device=dev)
Q /= Q.norm(dim=1, keepdim=True)
out = {}

# ---- (a) calibration: npairs unique rows, each read exactly once -----------------------------------
npairs = 1 << 19
ids = torch.randperm(N, device=dev)[:npairs].int().contiguous()
qidx = torch.randint(0, args.batch, (npairs,), device=dev).int()
dst = torch.empty(npairs, device=dev)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
st = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    _lib.check(lib.lm_dist_gather(C.c_void_p(X.data_ptr()), 0, D, 0, C.c_void_p(Q.data_ptr()), C.c_void_p(qidx.data_ptr()),
                                  C.c_void_p(ids.data_ptr()), npairs, C.c_void_p(dst.data_ptr()), C.c_void_p(st)))
ev0.record()
for _ in range(args.reps):
    _lib.check(lib.lm_dist_gather(C.c_void_p(X.data_ptr()), 0, D, 0, C.c_void_p(Q.data_ptr()), C.c_void_p(qidx.data_ptr()),
                                  C.c_void_p(ids.data_ptr()), npairs, C.c_void_p(dst.data_ptr()), C.c_void_p(st)))
ev1.record()
torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / args.reps
out["dist_gather"] = {"pairs": npairs, "row_bytes": D * 4, "known_row_bytes_total": npairs * D * 4, "ms": round(ms, 4),
                      "GBps_rows": round(npairs * D * 4 / ms / 1e6, 1)}

# ---- (b) stored-embedding search on a random regular graph -------------------------------------------
rng = np.random.default_rng(0)
nb = rng.integers(0, N, (N, args.deg), dtype=np.int32)
levels = np.ones(N, np.int32)
node_offsets = (np.arange(N + 1, dtype=np.uint64) * 2)
level_ptr = np.empty(2 * N, np.uint64)
level_ptr[0::2] = np.arange(N, dtype=np.uint64) * args.deg
level_ptr[1::2] = (np.arange(N, dtype=np.uint64) + 1) * args.deg
csr = HnswCsr(d=D, ntotal=N, metric_type=0, levels=levels, level_ptr=level_ptr, node_offsets=node_offsets,
              neighbors=nb.reshape(-1), entry_point=0, max_level=0)
idx = Mi355xIndex.from_csr(csr)
idx.set_stream(st)
idx.attach_table(X)
idx.set_profiling(True)
idx.set_option("update_variant", args.variant)
idx.set_option("persistent_table", 0 if args.lockstep else 1)
idx.set_option("persistent_wave", args.wave)
keep = {}
if args.provider:
    from leann_amd.devmem import as_tensor

    def provider(d_ids, n, stream):
        keep["e"] = X.index_select(0, as_tensor(d_ids, (n,), "int32").long())
        return keep["e"].data_ptr()

    idx.set_provider(provider)
prm = idx.make_params(ef=args.ef, beam=args.beam, recompute=args.provider, max_batch=args.batch, recompute_memo=args.memo)
for r in range(args.reps):
    idx.search_device(Q, 10, prm)
    s = idx.stats()
out["search"] = {"batch": args.batch, "ef": args.ef, "beam": args.beam, "variant": args.variant, "wave": args.wave, "lockstep": args.lockstep, "provider": args.provider, "memo": args.memo, "nunique": s["nunique"], "ndis": s["ndis"],
                 "launches": s["update_launches"], "update_ms": round(s["update_ms"], 3), "update_span_ms": round(s["update_span_ms"], 3), "expand_ms": round(s["expand_ms"], 3),
                 "evals_per_launch": round(s["ndis"] / s["update_launches"], 1),
                 "update_GBps_algorithmic": round(s["ndis"] * (D * 4 + 4) / s["update_ms"] / 1e6, 1),
                 "update_GBps_span": round(s["ndis"] * (D * 4 + 4) / max(s["update_span_ms"], 1e-9) / 1e6, 1)}
print(json.dumps(out))
