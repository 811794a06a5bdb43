"""``leann-backend-mi355x``: the LEANN backend plugin for MI355X.

Host-side mirror of the reference's HNSW backend (packages/leann-backend-hnsw/leann_backend_hnsw/
hnsw_backend.py): same factory / builder / searcher classes, the same ``search`` signature,
argument meaning, return value and error behaviour -- but the native search
(``faiss.IndexHNSW.search`` + per-hop ZMQ round trips to the embedding server) is replaced by
libleann_mi355x.so: graph in HBM, lock-step selective-recompute beam search, in-process encoder.
"""

from __future__ import annotations

import json
import logging
import pickle
import time
from pathlib import Path
from typing import Any, Literal, Optional

import numpy as np

from . import _lib
from ._compat import (
    LeannBackendBuilderInterface,
    LeannBackendFactoryInterface,
    LeannBackendSearcherInterface,
    register_backend,
)
from .csr_format import METRIC_INNER_PRODUCT, METRIC_L2, HnswCsr, read_index, write_index

logger = logging.getLogger(__name__)

METRIC_MAP = {"mips": METRIC_INNER_PRODUCT, "l2": METRIC_L2, "cosine": METRIC_INNER_PRODUCT}  # hnsw_backend.py:22-29


def normalize_l2(data: np.ndarray) -> np.ndarray:
    """hnsw_backend.py:32-35 (zero vectors stay zero)."""
    norms = np.linalg.norm(data, axis=1, keepdims=True)
    norms[norms == 0] = 1
    return data / norms


def hub_nodes(g: HnswCsr, ratio: float) -> np.ndarray:
    """The ceil(ratio*N) nodes with the highest level-0 in-degree (ties: smaller id), sorted ascending --
    the hubs whose embeddings the paper keeps cached."""
    n = g.ntotal
    k = min(n, int(np.ceil(ratio * n)))
    if k <= 0:
        return np.zeros(0, np.int32)
    p0 = g.node_offsets[:-1].astype(np.int64)
    beg, end = g.level_ptr[p0].astype(np.int64), g.level_ptr[p0 + 1].astype(np.int64)
    mask = np.zeros(g.neighbors.shape[0] + 1, np.int64)
    np.add.at(mask, beg, 1)
    np.add.at(mask, end, -1)
    lvl0 = np.cumsum(mask[:-1]) > 0
    indeg = np.bincount(g.neighbors[lvl0], minlength=n)
    order = np.lexsort((np.arange(n), -indeg))
    return np.sort(order[:k]).astype(np.int32)


@register_backend("mi355x")
class Mi355xBackend(LeannBackendFactoryInterface):
    @staticmethod
    def builder(**kwargs) -> LeannBackendBuilderInterface:
        return Mi355xBuilder(**kwargs)

    @staticmethod
    def searcher(index_path: str, **kwargs) -> LeannBackendSearcherInterface:
        return Mi355xSearcher(index_path, **kwargs)


class Mi355xBuilder(LeannBackendBuilderInterface):
    """Mirror of HNSWBuilder (hnsw_backend.py:49-117): same kwargs and defaults, writes the same
    compact-CSR ``<stem>.index`` file (pruned when ``is_recompute``)."""

    def __init__(self, **kwargs):
        self.build_params = kwargs.copy()
        self.is_compact = self.build_params.setdefault("is_compact", True)
        self.is_recompute = self.build_params.setdefault("is_recompute", True)
        self.M = self.build_params.setdefault("M", 32)
        self.efConstruction = self.build_params.setdefault("efConstruction", 200)
        self.distance_metric = self.build_params.setdefault("distance_metric", "mips")
        self.dimensions = self.build_params.get("dimensions")
        if not self.is_recompute and self.is_compact:
            logger.warning("is_recompute=False requires non-compact HNSW. Forcing is_compact=False.")
            self.is_compact = False
            self.build_params["is_compact"] = False

    def build(self, data: np.ndarray, ids: list, index_path: str, **kwargs) -> None:
        path = Path(index_path)
        path.parent.mkdir(parents=True, exist_ok=True)
        if data.dtype != np.float32:
            logger.warning(f"Converting data to float32, shape: {data.shape}")
            data = data.astype(np.float32)
        metric = self.distance_metric.lower()
        if metric not in METRIC_MAP:
            raise ValueError(f"Unsupported distance_metric '{self.distance_metric}'.")
        if metric == "cosine":
            data = normalize_l2(data)
        g = self._build_graph(data, metric)
        g.storage = np.ascontiguousarray(data, dtype=np.float32)
        # the file always uses the CSR layout; "non-compact" in the reference == embeddings kept
        write_index(path.parent / f"{path.stem}.index", g, prune_embeddings=bool(self.is_recompute))
        pq_bytes = int(self.build_params.get("pq_bytes", 0) or 0)
        if pq_bytes > 0:  # optional product quantiser for the two-level search (prune_ratio), cf. the fork's PQ pruning
            import torch

            from .pq import encode_pq, train_pq

            x = torch.from_numpy(np.ascontiguousarray(data))
            cb = train_pq(x, pq_bytes, seed=0)
            np.savez(path.parent / f"{path.stem}_pq.npz", codebooks=cb.numpy(), codes=encode_pq(x, cb).numpy())


    def _build_graph(self, data: np.ndarray, metric: str):
        """HNSW construction.  Small corpora: the sequential host builder (csrc/hnsw_build.cpp, the role of faiss' index.add in
        hnsw_backend.py:83-90).  From ``gpu_build_threshold`` vectors on (default 100 000) and with a HIP device present: the
        batched GPU builder driven by the HIP search kernel (gpu_graph_build.py) -- minutes instead of hours at 1M-60M chunks.
        ``hub_preserving_m`` > 0 additionally applies the paper's high-degree-preserving pruning (Algorithm 3): ~m links per
        node, full lists for the 2 % hub nodes -- the storage side of a recompute (pruned) index."""
        from .hnsw_builder import build_hnsw

        bp = self.build_params
        thr = int(bp.get("gpu_build_threshold", 100_000))
        m_low = int(bp.get("hub_preserving_m", 0) or 0)
        use_gpu = data.shape[0] >= thr and _lib.device_count() > 0
        if not use_gpu and m_low <= 0:
            return build_hnsw(data, metric, M=self.M, ef_construction=self.efConstruction)
        import torch

        from .gpu_graph_build import build_graph_gpu, prune_preserving_hubs

        if use_gpu:
            x = torch.from_numpy(np.ascontiguousarray(data)).to(torch.device("cuda", int(bp.get("device", 0))))
            g = build_graph_gpu(x, metric, M=self.M, ef_construction=self.efConstruction)
        else:
            x = torch.from_numpy(np.ascontiguousarray(data))
            g = build_hnsw(data, metric, M=self.M, ef_construction=self.efConstruction)
        if m_low > 0:
            g = prune_preserving_hubs(g, x, self.M, m_low, float(bp.get("hub_fraction", 0.02)))
        return g


class _NoServer:
    """Stands in for EmbeddingServerManager (leann/api.py:798-806 calls .stop_server())."""

    server_port = None

    def stop_server(self) -> None:
        return None


class Mi355xSearcher(LeannBackendSearcherInterface):
    """Mirror of HNSWSearcher (hnsw_backend.py:120-253) on top of the C ABI."""

    def __init__(self, index_path: str, **kwargs):
        self.index_path = Path(index_path)
        self.index_dir = self.index_path.parent
        self.meta = kwargs.get("meta") or self._load_meta()
        if not self.meta:
            raise ValueError("Searcher requires metadata from .meta.json.")
        self.dimensions = self.meta.get("dimensions")
        if not self.dimensions:
            raise ValueError("Dimensions not found in Leann metadata.")
        self.embedding_model = self.meta.get("embedding_model")
        self.embedding_mode = self.meta.get("embedding_mode", "sentence-transformers")
        self.embedding_server_manager = _NoServer()
        bk = self.meta.get("backend_kwargs", {})
        self.distance_metric = (kwargs.get("distance_metric") or bk.get("distance_metric", "mips")).lower()
        if self.distance_metric not in METRIC_MAP:
            raise ValueError(f"Unsupported distance_metric '{self.distance_metric}'.")
        # leann/api.py:472-479 writes these two flags only for backend_name == "hnsw"; derive them from
        # the persisted build kwargs otherwise
        self.is_compact = self.meta.get("is_compact", bk.get("is_compact", True))
        self.is_pruned = self.meta.get("is_pruned", bool(self.is_compact and bk.get("is_recompute", True)))
        self.index_file = self.index_dir / f"{self.index_path.stem}.index"
        if not self.index_file.exists() and not self._has_alternative_index_files():
            raise FileNotFoundError(f"HNSW index file not found at {self.index_file}")
        self.device = int(kwargs.get("device", 0))
        self.encoder_batch = int(kwargs.get("encoder_batch", 5461))  # chunks per encoder sub-batch (x 192 tokens = 1M tokens: measured 61.9 vs 64.4 ms per 11,264 chunks against 524k)
        self.encoder_dtype = kwargs.get("encoder_dtype", "float16")
        # fraction of the highest in-degree nodes whose embeddings stay cached in HBM (LEANN paper section 5;
        # 0 = pure recompute, the reference's behaviour)
        self.hub_cache_ratio = float(kwargs.get("hub_cache_ratio", bk.get("hub_cache_ratio", 0.0)) or 0.0)
        # Seeded random encoder weights + a stand-in vocabulary when the embedding model's checkpoint is not available
        # locally: ONLY for synthetic corpora / tests whose index was built with the same random encoder.  Off by default:
        # a production index must be searched with the weights it was built from, so a missing checkpoint raises.
        import os as _os

        self.allow_random_weights = bool(kwargs.get("allow_random_weights", bk.get(
            "allow_random_weights", _os.environ.get("LEANN_MI355X_ALLOW_RANDOM_WEIGHTS", "0") == "1")))
        self._index = None
        self._provider = None
        self._encoder = None
        self._tokenizer = None
        self._tokens = None
        if kwargs.get("enable_warmup"):
            self._ensure_index_loaded()

    # ---- loading ---------------------------------------------------------------------------
    def _has_alternative_index_files(self) -> bool:
        """Subclasses that can serve another backend's files (the DiskANN-style searcher: a stock DiskANN bundle) say so here."""
        return False

    def _load_meta(self) -> dict:
        meta_path = self.index_dir / f"{self.index_path.name}.meta.json"
        if not meta_path.exists():
            raise FileNotFoundError(f"Leann metadata file not found at {meta_path}")
        with open(meta_path, encoding="utf-8") as f:
            return json.load(f)

    def _ensure_index_loaded(self):
        """Graph -> HBM.  Equivalent of faiss.read_index(...MMAP, HNSWIndexConfig) (hnsw_backend.py:145-151);
        there is no CPU fallback: without a HIP device this raises."""
        if self._index is None:
            from .index import Mi355xIndex

            _lib.require_gpu()
            self._index = Mi355xIndex.read(str(self.index_file), device=self.device)
            if self._index.info.d != int(self.dimensions):
                raise ValueError(f"index dimension {self._index.info.d} != meta dimensions {self.dimensions}")
            pqf = self.index_dir / f"{self.index_path.stem}_pq.npz"
            self._has_pq = False
            if pqf.exists():  # product quantiser for the two-level search (prune_ratio > 0)
                z = np.load(pqf)
                self._index.attach_pq(z["codebooks"], z["codes"])
                self._has_pq = True
        return self._index

    def _torch_device(self):
        import torch

        return torch.device("cuda", self.device)

    def _ensure_encoder(self):
        if self._encoder is None:
            import torch

            from .encoder import BertEncoder

            if not self.embedding_model:
                raise ValueError("Cannot use recompute mode without 'embedding_model' in meta.json.")
            _lib.require_gpu()
            enc = BertEncoder.load(self.embedding_model, allow_random=self.allow_random_weights)
            dt = torch.float16 if self.encoder_dtype == "float16" else torch.float32
            self._encoder = enc.to(self._torch_device(), dtype=dt).eval()
        return self._encoder

    def _passages_file(self, passages_source_file: Optional[str]) -> Path:
        """Resolve the JSONL of passages (meta.json 'passage_sources', leann/api.py:144-192)."""
        cands = []
        for src in self.meta.get("passage_sources", []):
            if src.get("path_relative"):
                cands.append(self.index_dir / src["path_relative"])
            if src.get("path"):
                p = Path(src["path"])
                cands.append(p if p.is_absolute() else self.index_dir / p)
                cands.append(p)
        cands.append(Path(str(self.index_path) + ".passages.jsonl"))
        for c in cands:
            if c.exists():
                return c
        raise FileNotFoundError(f"passages file not found (tried {[str(c) for c in cands]})")

    def _read_passage_texts(self, path: Path) -> list[str]:
        """Node i <-> i-th passage (HNSW labels are insertion indices, hnsw_backend.py:89,251)."""
        texts = []
        with open(path, encoding="utf-8") as f:
            for line in f:
                if line.strip():
                    texts.append(json.loads(line).get("text", ""))
        return texts

    def attach_token_store(self, tokens, tokenizer=None) -> None:
        """Use an already tokenised corpus (synthetic benchmarks; skips the JSONL pass)."""
        self._tokens = tokens
        self._tokenizer = tokenizer
        self._provider = None

    def attach_encoder(self, encoder) -> None:
        self._encoder = encoder
        self._provider = None

    def _ensure_server_running(self, passages_source_file: str, port: Optional[int], **kwargs) -> int:
        """The reference spawns the embedding-server subprocess here (searcher_base.py:58-84,
        embedding_server_manager.py:76-104).  Ours: make the in-process encoder + HBM token store
        ready and return the port unchanged (nothing listens on it)."""
        idx = self._ensure_index_loaded()
        enc = self._ensure_encoder()
        if self._tokens is None:
            from .token_store import TokenStore
            from .tokenizer import load_tokenizer

            t0 = time.time()
            texts = self._read_passage_texts(self._passages_file(passages_source_file))
            if len(texts) != idx.info.ntotal:
                raise ValueError(f"{len(texts)} passages but the index holds {idx.info.ntotal} nodes")
            max_len = min(enc.cfg.max_seq_length, enc.cfg.max_pos)
            self._tokenizer = load_tokenizer(self.embedding_model, max_len, str(self.index_path), texts, enc.cfg.vocab_size,
                                             allow_stand_in=enc.weights_source == "random")
            seqs = self._tokenizer.encode_batch(texts)
            self._tokens = TokenStore.from_lists(seqs, device=self.device)
            logger.info(f"token store ready: {len(texts)} passages in {time.time() - t0:.2f}s ({self._tokenizer.kind})")
        if self._provider is None:
            from .recompute import RecomputeProvider

            self._provider = RecomputeProvider(enc, self._tokens, idx.info.d_padded, self._torch_device(),
                                               batch_size=self.encoder_batch)
            idx.set_provider(self._provider)
            if self.hub_cache_ratio > 0:
                self._attach_hub_cache(idx)
        return int(port) if port is not None else 0

    def _attach_hub_cache(self, idx) -> None:
        import torch

        ids = hub_nodes(read_index(self.index_file), self.hub_cache_ratio)
        if ids.shape[0] == 0:
            return
        dev = self._torch_device()
        emb = self._provider.embed_ids(torch.from_numpy(ids).to(dev))
        buf = torch.zeros((ids.shape[0], idx.info.d_padded), dtype=torch.float32, device=dev)
        buf[:, : emb.shape[1]] = emb
        idx.set_hub_cache(ids, buf)
        logger.info(f"hub cache: {ids.shape[0]} embeddings kept in HBM")

    def compute_query_embedding(self, query: str, use_server_if_available: bool = True,
                                zmq_port: Optional[int] = None) -> np.ndarray:
        """(1, D) float32 embedding of a query string with the same in-process encoder
        (searcher_base.py:86-128 goes through the server / falls back to a local model)."""
        import torch

        enc = self._ensure_encoder()
        if self._tokenizer is None:
            from .tokenizer import load_tokenizer

            self._tokenizer = load_tokenizer(self.embedding_model, min(enc.cfg.max_seq_length, enc.cfg.max_pos),
                                             str(self.index_path), None, enc.cfg.vocab_size,
                                             allow_stand_in=enc.weights_source == "random")
        ids = self._tokenizer.encode_batch([query])[0]
        dev = self._torch_device()
        t = torch.tensor([ids], dtype=torch.int32, device=dev)
        lens = torch.tensor([len(ids)], dtype=torch.int32, device=dev)
        with torch.no_grad():
            e = enc(t, lens)
        return e.float().cpu().numpy()

    # ---- search ----------------------------------------------------------------------------
    def search(self, query: np.ndarray, top_k: int, zmq_port: Optional[int] = None, complexity: int = 64,
               beam_width: int = 1, prune_ratio: float = 0.0, recompute_embeddings: bool = True,
               pruning_strategy: Literal["global", "local", "proportional"] = "global", batch_size: int = 0,
               **kwargs) -> dict[str, Any]:
        """Same contract as HNSWSearcher.search (hnsw_backend.py:153-253):
        returns {"labels": list[list[str]] (B x k), "distances": np.ndarray (B, k) float32}."""
        if not recompute_embeddings and self.is_pruned:
            raise RuntimeError(
                "Recompute is required for pruned/compact HNSW index. "
                "Re-run search with --recompute, or rebuild with --no-recompute and --no-compact.")
        if recompute_embeddings and zmq_port is None:
            raise ValueError("zmq_port must be provided if recompute_embeddings is True")
        query = np.atleast_2d(np.asarray(query))
        if query.dtype != np.float32:
            query = query.astype(np.float32)
        if self.distance_metric == "cosine":
            query = normalize_l2(query)
        idx = self._ensure_index_loaded()
        if recompute_embeddings and self._provider is None:
            self._ensure_server_running(str(self.index_dir / f"{self.index_path.name}.meta.json"), zmq_port)
        # hnsw_backend.py:209-217: OpenAI cosine models disable the relative distance check
        if prune_ratio and not getattr(self, "_has_pq", False):
            logger.warning("prune_ratio > 0 needs a product quantiser (<stem>_pq.npz, build with pq_bytes=...); ignoring it")
            prune_ratio = 0.0
        model = (self.meta.get("embedding_model") or "").lower()
        check_rel = not (self.distance_metric == "cosine" and any(m in model for m in ["text-embedding", "openai"]))
        params = idx.make_params(
            ef=int(complexity), beam=int(beam_width), check_relative_distance=check_rel, recompute=bool(recompute_embeddings),
            prune_ratio=float(prune_ratio), local_prune=(pruning_strategy == "local"),
            send_neigh_times_ratio=(1.0 if pruning_strategy == "proportional" else 0.0),
            batch_size=int(batch_size), zmq_port=int(zmq_port or 0), max_batch=int(kwargs.get("max_batch", 0)),
            # per-call memo (default on): in a multi-query call a node is recomputed at most once; `dedup_node_dis` is the reference's name
            # for its own "cache and reuse" switch (diskann_backend.py:394,413), accepted here as a synonym
            recompute_memo=bool(kwargs.get("recompute_memo", kwargs.get("dedup_node_dis", True))))
        if recompute_embeddings:
            import torch

            idx.set_stream(torch.cuda.current_stream(self._torch_device()).cuda_stream)
        t0 = time.time()
        distances, labels = idx.search(np.ascontiguousarray(query), int(top_k), params)
        logger.info(f"  Search time in Mi355xSearcher.search() backend: {time.time() - t0} seconds")
        string_labels = [[str(int(l)) for l in row] for row in labels]
        return {"labels": string_labels, "distances": distances}

    def last_stats(self) -> dict:
        return self._index.stats() if self._index is not None else {}

    def cleanup(self) -> None:
        if self._index is not None:
            self._index.close()
            self._index = None

    def __del__(self):
        try:
            self.cleanup()
        except Exception:  # noqa: BLE001
            pass


def write_leann_bundle(index_path: str, texts: list[str], embeddings: np.ndarray, embedding_model: str,
                       backend_name: str = "mi355x", **backend_kwargs) -> None:  # noqa: C901
    """Write the LEANN index bundle the way LeannBuilder.build_index does (leann/api.py:409-481):
    ``<index_path>.passages.jsonl`` + ``.passages.idx`` (pickle {id: byte offset}) +
    ``<index_path>.meta.json`` + the backend's ``<stem>.index``.  Lets the backend be used (and
    tested) end to end without importing leann-core."""
    path = Path(index_path)
    path.parent.mkdir(parents=True, exist_ok=True)
    pj = Path(str(path) + ".passages.jsonl")
    offsets = {}
    with open(pj, "w", encoding="utf-8") as f:
        for i, t in enumerate(texts):
            offsets[str(i)] = f.tell()
            f.write(json.dumps({"id": str(i), "text": t, "metadata": {}}, ensure_ascii=False) + "\n")
    with open(str(path) + ".passages.idx", "wb") as f:
        pickle.dump(offsets, f)
    if backend_name == "mi355x_diskann":
        b = Mi355xDiskannBuilder(dimensions=int(embeddings.shape[1]), **backend_kwargs)
        b.build_params.setdefault("distance_metric", "mips")
        b.is_compact, b.is_recompute = True, bool(backend_kwargs.get("is_recompute", False))
    else:
        b = Mi355xBuilder(dimensions=int(embeddings.shape[1]), **backend_kwargs)
    b.build(embeddings, [str(i) for i in range(len(texts))], str(path))
    meta = {
        "version": "1.0", "backend_name": backend_name, "embedding_model": embedding_model,
        "dimensions": int(embeddings.shape[1]), "backend_kwargs": b.build_params, "embedding_mode": "sentence-transformers",
        "passage_sources": [{"type": "jsonl", "path": pj.name, "index_path": path.name + ".passages.idx",
                             "path_relative": pj.name, "index_path_relative": path.name + ".passages.idx"}],
        "is_compact": bool(b.is_compact), "is_pruned": bool(b.is_compact and b.is_recompute),
    }
    with open(str(path) + ".meta.json", "w", encoding="utf-8") as f:
        json.dump(meta, f, indent=2)


# =============================================================================================
# DiskANN-style backend: PQ traversal + deferred rerank
# =============================================================================================


def pq_bytes_for_budget(num_vectors: int, dim: int, search_memory_gb: float | None = None) -> int:
    """PQ bytes per vector from DiskANN's memory budget rule: search_memory_maximum defaults to 1/10
    of the fp32 embedding size (diskann_backend.py:105-111) => ~ dim*4/10 bytes per vector, rounded
    to a divisor of dim that is a multiple of 4."""
    budget = (search_memory_gb * (1024**3) / max(num_vectors, 1)) if search_memory_gb else dim * 4 / 10
    # <= 96 sub-quantisers: the m*256*4-byte lookup table must stay LDS resident (160 KB/CU) next to the
    # candidate list and a beam_width=64 frontier
    cands = [m for m in range(4, min(dim, 96) + 1, 4) if dim % m == 0]
    return min(cands, key=lambda m: (abs(m - budget), m)) if cands else 4


@register_backend("mi355x_diskann")
class Mi355xDiskannBackend(LeannBackendFactoryInterface):
    @staticmethod
    def builder(**kwargs) -> LeannBackendBuilderInterface:
        return Mi355xDiskannBuilder(**kwargs)

    @staticmethod
    def searcher(index_path: str, **kwargs) -> LeannBackendSearcherInterface:
        return Mi355xDiskannSearcher(index_path, **kwargs)


class Mi355xDiskannBuilder(LeannBackendBuilderInterface):
    """Mirror of DiskannBuilder (diskann_backend.py:142-297): kwargs ``complexity`` (build list size),
    ``graph_degree``, ``search_memory_maximum`` (PQ budget), ``distance_metric``, ``num_threads``.
    Writes ``<stem>.index`` (flat graph in the compact-CSR container, entry = medoid, embeddings kept
    unless ``is_recompute``) and ``<stem>_pq.npz`` (codebooks + codes: the role of
    ``_pq_pivots.bin`` / ``_pq_compressed.bin``, whose real layouts live in the absent fork)."""

    def __init__(self, **kwargs):
        self.build_params = kwargs.copy()

    def build(self, data: np.ndarray, ids: list, index_path: str, **kwargs) -> None:
        import torch

        from .hnsw_builder import build_hnsw
        from .pq import encode_pq, flat_graph, train_pq

        path = Path(index_path)
        path.parent.mkdir(parents=True, exist_ok=True)
        if data.dtype != np.float32:
            logger.warning(f"Converting data to float32, shape: {data.shape}")
            data = data.astype(np.float32)
        bk = {**self.build_params, **kwargs}
        if "backend_kwargs" in bk:
            bk.update(bk.pop("backend_kwargs"))
        metric = bk.get("distance_metric", "mips").lower()
        if metric not in METRIC_MAP:
            raise ValueError(f"Unsupported distance_metric '{bk.get('distance_metric', 'unknown')}'.")
        if metric == "cosine":
            data = normalize_l2(data)
        degree = int(bk.get("graph_degree", 32))
        g = build_hnsw(data, metric, M=max(2, degree // 2), ef_construction=max(int(bk.get("complexity", 64)), degree),
                       num_threads=int(bk.get("num_threads", 0)))
        fg = flat_graph(g, data)
        fg.storage = np.ascontiguousarray(data)
        write_index(path.parent / f"{path.stem}.index", fg, prune_embeddings=bool(bk.get("is_recompute", False)))
        m = int(bk.get("pq_bytes") or pq_bytes_for_budget(data.shape[0], data.shape[1], bk.get("search_memory_maximum")))
        x = torch.from_numpy(data)
        cb = train_pq(x, m, seed=0)
        codes = encode_pq(x, cb)
        np.savez(path.parent / f"{path.stem}_pq.npz", codebooks=cb.numpy(), codes=codes.numpy())


class Mi355xDiskannSearcher(Mi355xSearcher):
    """Mirror of DiskannSearcher (diskann_backend.py:300-471).  Serves the bundle Mi355xDiskannBuilder writes (``<stem>.index`` +
    ``<stem>_pq.npz``) and -- copy the meta.json with ``"backend_name": "mi355x_diskann"`` -- a bundle written by the STOCK DiskANN
    backend in its default (non-recompute) mode: ``<stem>_pq_pivots.bin``, ``<stem>_pq_compressed.bin``, ``<stem>_disk.index``,
    ``<stem>_disk.index_medoids.bin`` / ``_max_base_norm.bin`` in the public DiskANN layout (leann_amd/diskann_files.py).  A stock
    recompute-mode bundle keeps its graph only in the fork's private ``_disk_graph.index`` / ``_partition.bin``: not readable."""

    def _stock_prefix(self) -> str:
        return str(self.index_dir / self.index_path.stem)

    def _has_alternative_index_files(self) -> bool:
        pre = self._stock_prefix()
        return all(Path(pre + sfx).exists() for sfx in ("_pq_pivots.bin", "_pq_compressed.bin", "_disk.index"))

    def __init__(self, index_path: str, **kwargs):
        # the base class warms up at the end of ITS constructor, which would run this class' _ensure_index_loaded before _stock / pq_file
        # exist (LeannSearcher forwards enable_warmup to the backend searcher, api.py:623-642): warm up at the end of this one instead
        warm = bool(kwargs.pop("enable_warmup", False))
        super().__init__(index_path, **kwargs)
        self.num_threads = kwargs.get("num_threads", 8)
        self.pq_file = self.index_dir / f"{self.index_path.stem}_pq.npz"
        self._stock = not self.index_file.exists()
        if self._stock:
            self.is_pruned = False  # the stock non-recompute bundle carries its vectors (_disk.index)
        elif not self.pq_file.exists():
            raise FileNotFoundError(f"PQ file not found at {self.pq_file}")
        if Path(self._stock_prefix() + "_disk_graph.index").exists() and self._stock and not self._has_alternative_index_files():
            raise FileNotFoundError("this is a recompute-mode bundle of the stock DiskANN backend: its graph (_disk_graph.index / _partition.bin) is in "
                                    "the fork's private format, which leann-backend-mi355x does not read")
        if warm:
            self._ensure_index_loaded()

    def _ensure_index_loaded(self):
        if self._stock and self._index is None:
            from .diskann_files import load_stock_bundle
            from .index import Mi355xIndex

            _lib.require_gpu()
            b = load_stock_bundle(self._stock_prefix(), int(self.dimensions), self.distance_metric)
            self._index = Mi355xIndex.from_csr(b.graph(), device=self.device)
            self._index.attach_table(b.vectors)
            self._index.attach_pq(b.codebooks, b.codes, b.chunk_offsets)
            self._has_pq = True
            return self._index
        idx = super()._ensure_index_loaded()  # attaches <stem>_pq.npz (checked to exist in __init__): one upload only
        if not self._has_pq:
            raise FileNotFoundError(f"PQ file not found at {self.pq_file}")
        return idx

    def search(self, query: np.ndarray, top_k: int, complexity: int = 64, beam_width: int = 1, prune_ratio: float = 0.0,
               recompute_embeddings: bool = False, pruning_strategy: Literal["global", "local", "proportional"] = "global",
               zmq_port: Optional[int] = None, batch_recompute: bool = False, dedup_node_dis: bool = False,
               **kwargs) -> dict[str, Any]:
        """Same contract as DiskannSearcher.search (diskann_backend.py:383-471)."""
        if recompute_embeddings and zmq_port is None:
            raise ValueError("zmq_port must be provided if recompute_embeddings is True")
        if pruning_strategy == "proportional":
            raise NotImplementedError(
                "DiskANN backend does not support 'proportional' pruning strategy. Use 'global' or 'local' instead.")
        if (prune_ratio or batch_recompute or dedup_node_dis or pruning_strategy == "local") and not getattr(self, "_inert_knobs_logged", False):
            # These parametrise the per-hop neighbour recomputation of the fork's batch_search, which the reference switches off for every
            # search (recompute_neighors = False, diskann_backend.py:444-451): the traversal is PQ-only there and here, so they change nothing.
            logger.warning("prune_ratio / pruning_strategy='local' / batch_recompute / dedup_node_dis have no effect on the DiskANN-style path: "
                           "the traversal never recomputes neighbour distances (as in the reference, diskann_backend.py:444-451); accepted for "
                           "interface compatibility")
            self._inert_knobs_logged = True
        query = np.atleast_2d(np.asarray(query))
        if query.dtype != np.float32:
            query = query.astype(np.float32)
        if self.distance_metric == "cosine":
            query = normalize_l2(query)
        idx = self._ensure_index_loaded()
        if recompute_embeddings and self._provider is None:
            self._ensure_server_running(str(self.index_dir / f"{self.index_path.name}.meta.json"), zmq_port)
        if recompute_embeddings:
            import torch

            idx.set_stream(torch.cuda.current_stream(self._torch_device()).cuda_stream)
        # traversal always on PQ distances; recompute => one final rerank via deferred fetch (:444-450)
        params = idx.make_pq_params(int(complexity), int(beam_width), use_deferred_fetch=bool(recompute_embeddings),
                                    skip_search_reorder=bool(kwargs.get("skip_search_reorder", False)),
                                    num_threads=int(self.num_threads), dedup_node_dis=bool(dedup_node_dis),
                                    prune_ratio=float(prune_ratio), batch_recompute=bool(batch_recompute),
                                    use_global_pruning=(pruning_strategy != "local"))
        labels, distances = idx.pq_search(np.ascontiguousarray(query), int(top_k), params)
        string_labels = [[str(int(l)) for l in row] for row in labels]
        return {"labels": string_labels, "distances": distances}
