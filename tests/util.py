"""Shared fixtures for the parity tests: seeded synthetic corpora + graphs."""
from __future__ import annotations

import numpy as np


def clustered(n: int, d: int, seed: int, n_centers: int = 64, sigma: float = 0.4, normalize: bool = True):
    rng = np.random.default_rng(seed)
    cent = rng.standard_normal((n_centers, d)).astype(np.float32)
    x = (cent[rng.integers(0, n_centers, n)] + sigma * rng.standard_normal((n, d))).astype(np.float32)
    if normalize:
        x /= np.linalg.norm(x, axis=1, keepdims=True)
    return np.ascontiguousarray(x, dtype=np.float32)


def queries_near(x: np.ndarray, nq: int, seed: int, noise: float = 0.05, normalize: bool = True):
    rng = np.random.default_rng(seed)
    q = x[rng.integers(0, x.shape[0], nq)] + noise * rng.standard_normal((nq, x.shape[1])).astype(np.float32)
    if normalize:
        q /= np.linalg.norm(q, axis=1, keepdims=True)
    return np.ascontiguousarray(q, dtype=np.float32)


def oracle_graph(g, d: int):
    from oracle import oracle as orc

    return orc.OracleGraph(g.node_offsets, g.level_ptr, g.neighbors, g.levels, g.entry_point, g.max_level, g.metric_type, d)


def recall_at_k(ids: np.ndarray, gt: np.ndarray) -> float:
    k = gt.shape[1]
    return float(np.mean([len(set(ids[i].tolist()) & set(gt[i].tolist())) / k for i in range(gt.shape[0])]))
