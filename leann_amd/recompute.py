"""The recompute provider: node ids -> tokens (HBM gather kernel) -> BERT forward -> fp32 embeddings.

This is the in-process replacement of the reference's embedding-recompute server round trip
(ZMQ REQ [[ids],[query]] -> PassageManager lookups -> compute_embeddings -> reply;
hnsw_embedding_server.py:148-284): no process boundary, no serialisation, everything stays in HBM
and on one HIP stream.
"""

from __future__ import annotations

import torch

from .encoder import BertEncoder
from .token_store import TokenStore


class RecomputeProvider:
    """Callable ``(d_ids_ptr, n, stream_ptr) -> device pointer of fp32 [n][d_padded]``."""

    def __init__(self, encoder: BertEncoder, tokens: TokenStore, d_padded: int, device: torch.device,
                 batch_size: int = 5461, bucket: int = 32, max_seq_length: int | None = None):
        self.encoder = encoder
        self.tokens = tokens
        self.dp = d_padded
        self.device = device
        self.batch_size = batch_size
        self.bucket = bucket
        self.T = min(max_seq_length or encoder.cfg.max_seq_length, encoder.cfg.max_pos, max(tokens.max_len, 1))
        self._cap = 0
        self._ids = self._lens = self._out = None
        self.chunks = 0       # statistics: chunks encoded
        self.tokens_seen = 0  # only updated when count_tokens is set (forces a sync)
        self.count_tokens = False

    def _ensure(self, n: int):
        if n > self._cap:
            cap = max(n, int(self._cap * 1.5), 1024)
            self._ids = torch.empty((cap, self.T), dtype=torch.int32, device=self.device)
            self._lens = torch.empty((cap,), dtype=torch.int32, device=self.device)
            self._out = torch.zeros((cap, self.dp), dtype=torch.float32, device=self.device)
            self._cap = cap

    @torch.no_grad()
    def __call__(self, d_ids_ptr: int, n: int, stream_ptr: int) -> int:
        self._ensure(n)
        ids, lens = self._ids[:n], self._lens[:n]
        self.tokens.gather(d_ids_ptr, n, self.T, self.encoder.cfg.pad_id, ids, lens, stream_ptr)
        emb = self.encoder.encode_tokens(ids, lens, batch_size=self.batch_size, bucket=self.bucket)
        d = emb.shape[1]
        if d == self.dp:
            self._keep = emb.contiguous()
            out = self._keep
        else:
            self._out[:n, :d] = emb
            out = self._out
        self.chunks += n
        if self.count_tokens:
            self.tokens_seen += int(lens.sum())
        return out.data_ptr()

    @torch.no_grad()
    def embed_ids(self, ids: torch.Tensor) -> torch.Tensor:
        """Embeddings of the given node ids (device int32 tensor) -> fp32 [n, D] (new tensor)."""
        n = ids.shape[0]
        self._ensure(n)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self.tokens.gather(ids.data_ptr(), n, self.T, self.encoder.cfg.pad_id, self._ids[:n], self._lens[:n], stream)
        return self.encoder.encode_tokens(self._ids[:n], self._lens[:n], batch_size=self.batch_size, bucket=self.bucket)
