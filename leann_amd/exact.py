"""Exact (brute-force) top-k of inner products on the GPU: the ground truth of the benchmarks' recall figures.

Measurement helper -- nothing on the search path uses it.  Written as a blocked loop on purpose: a single ``Q @ X.T`` whose OUTPUT has
more than 2^31 elements comes back wrong from the GEMM library on this stack (round 4, BASELINE configs[2] at 10M chunks: 256 queries x
10M chunks = 2.56e9 scores; rows 215 .. 255 of every 256-query block were garbage, which capped every recall figure of that run at
214.7 / 256 = 0.84 -- for the graph search, for a brute-force ADC ranking and for every candidate-list size alike)."""
from __future__ import annotations

import torch

MAX_SCORES = 1 << 28  # scores per GEMM call: far below 2^31, ~1 GiB of fp32


@torch.no_grad()
def exact_topk_ip(Q: torch.Tensor, X: torch.Tensor, k: int, q_block: int = 256):
    """(values [nq, k] f32, indices [nq, k] i64) of the k largest Q X^T per query, ties broken by torch.topk.  Q [nq, D], X [n, D] fp32."""
    nq, n = Q.shape[0], X.shape[0]
    k = min(k, n)
    x_block = max(k, min(n, MAX_SCORES // max(1, min(q_block, nq))))
    vals = torch.empty((nq, k), dtype=torch.float32, device=Q.device)
    idxs = torch.empty((nq, k), dtype=torch.int64, device=Q.device)
    for b0 in range(0, nq, q_block):
        qb = Q[b0 : b0 + q_block].float()
        bv = bi = None
        for c0 in range(0, n, x_block):
            v, i = torch.topk(qb @ X[c0 : c0 + x_block].float().T, min(k, n - c0), dim=1)
            i = i + c0
            if bv is None:
                bv, bi = v, i
            else:
                cv, ci = torch.cat((bv, v), 1), torch.cat((bi, i), 1)
                sel = torch.topk(cv, k, dim=1).indices
                bv, bi = torch.gather(cv, 1, sel), torch.gather(ci, 1, sel)
        vals[b0 : b0 + qb.shape[0]], idxs[b0 : b0 + qb.shape[0]] = bv, bi
    return vals, idxs
