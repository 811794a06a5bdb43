"""The recompute provider: node ids -> tokens (HBM gather kernel) -> BERT forward -> fp32 embeddings.

This is the in-process replacement of the reference's embedding-recompute server round trip
(ZMQ REQ [[ids],[query]] -> PassageManager lookups -> compute_embeddings -> reply;
hnsw_embedding_server.py:148-284): no process boundary, no serialisation, everything stays in HBM
and on one HIP stream.

Two forms of the same provider.  For models inside the one-call forward's envelope (hidden 384, mean pooling, fp16: the
all-MiniLM / bge-small family) the provider is LIBRARY code (csrc/lm_recompute.hip, ``RecomputeProvider.native()``): the index calls it
directly, no interpreter in the search loop and one host synchronisation per round.  For every other model (hidden 768: bge-base,
contriever) the provider is this class's ``__call__`` -- token gather + ``BertEncoder.encode_tokens`` over the general kernels.  Both
are GPU paths over the same kernels and return bit-identical embeddings where both apply.
"""

from __future__ import annotations

import ctypes as C
import os

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
        self._chunks = 0      # statistics: chunks encoded by the Python form
        self._native_chunks0 = 0
        self.tokens_seen = 0  # only updated when count_tokens is set (forces a sync)
        self.count_tokens = False
        self._native = None  # lm_recompute handle (created on first use)
        self._native_tried = False
        import weakref

        self._attached = weakref.WeakSet()  # indexes whose provider is this handle (Mi355xIndex.set_provider)

    # ---- the library-side form (csrc/lm_recompute.hip) ------------------------------------------------
    def native(self):
        """The ``lm_recompute`` handle of this provider (created once), or None when the model / token store is outside the built-in
        provider's envelope, ``LEANN_MI355X_NATIVE_PROVIDER=0`` is set (A/B), a kernel-selection switch or the per-kernel timers are
        on (those go through the per-kernel launch path of encoder.py).  ``Mi355xIndex.set_provider`` asks for it."""
        from .encoder import KERNEL_SELECTION_KEYS, KernelTimers

        if (os.environ.get("LEANN_MI355X_NATIVE_PROVIDER", "1") != "1" or os.environ.get("LEANN_MI355X_ONECALL", "1") != "1"
                or KernelTimers.active is not None or any(k in os.environ for k in KERNEL_SELECTION_KEYS)):
            return None
        cfg = self.encoder.cfg
        fused = cfg.hidden == 384 and self.T <= 256
        # cached packs; a NEW pack object means the weights changed: the handle holds the old pointers
        pk = self.encoder.onecall_model() if fused else None
        if pk is None:
            fused, pk = False, self.encoder.general_model()  # also hidden 384 with an ffn the fused layer tail does not take
        if self._native_tried and pk is getattr(self, "_native_pack", None):
            return self._native
        self.close()
        self._native_tried, self._native_pack = True, pk
        t_max = 256 if cfg.heads and cfg.hidden == cfg.heads * 32 else 512
        if pk is None or not (0 < self.T <= t_max) or self.dp != cfg.hidden or not self.encoder.word.weight.is_cuda:
            return None
        from . import _lib

        h = C.c_void_p()
        create = _lib.load().lm_recompute_create if fused else _lib.load().lm_recompute_create_general
        _lib.check(create(C.byref(pk["model"]), self.tokens._h, int(self.T), int(self.batch_size) * 192, C.byref(h)),
                   "lm_recompute_create" if fused else "lm_recompute_create_general")
        self._native = h  # self._native_pack (the device weights) outlives it
        return h

    def native_stats(self) -> dict:
        from . import _lib

        st = _lib.RecomputeStats()
        if self._native is not None:
            _lib.check(_lib.load().lm_recompute_get_stats(self._native, C.byref(st)), "lm_recompute_get_stats")
        return {n: int(getattr(st, n)) for n, _ in _lib.RecomputeStats._fields_}

    @property
    def chunks(self) -> int:
        """Chunks encoded since the counter was last reset (both forms)."""
        return self._chunks + (self.native_stats()["chunks"] - self._native_chunks0 if self._native is not None else 0)

    @chunks.setter
    def chunks(self, v: int) -> None:
        self._chunks = int(v)
        self._native_chunks0 = self.native_stats()["chunks"] if self._native is not None else 0

    def close(self) -> None:
        """Free the library-side handle.  Indexes still attached to it lose their provider first (a recompute search on them then fails
        loudly with "no embedding provider" instead of reading freed memory)."""
        if self._native is not None:
            from . import _lib

            for idx in list(self._attached):
                if getattr(idx, "_h", None) and getattr(idx, "native_provider", False) and idx._provider_keepalive is self:
                    idx.set_provider(None)
            self._attached.clear()
            _lib.load().lm_recompute_free(self._native)
            self._native = None
        self._native_tried = False

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

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
        self._chunks += n
        if self.count_tokens:
            self.tokens_seen += int(lens.sum())
        return out.data_ptr()

    @torch.no_grad()
    def embed_ids(self, ids: torch.Tensor) -> torch.Tensor:
        """Embeddings of the given node ids (device int32 tensor) -> fp32 [n, D] (new tensor)."""
        n = ids.shape[0]
        stream = torch.cuda.current_stream(self.device).cuda_stream
        h = self.native()
        if h is not None and ids.dtype == torch.int32 and ids.is_contiguous():
            from . import _lib

            out = torch.empty((n, self.encoder.cfg.hidden), dtype=torch.float32, device=ids.device)
            _lib.check(_lib.load().lm_recompute_embed(h, C.c_void_p(ids.data_ptr()), n, C.c_void_p(out.data_ptr()), C.c_void_p(stream)),
                       "lm_recompute_embed")
            return out
        self._ensure(n)
        self.tokens.gather(ids.data_ptr(), n, self.T, self.encoder.cfg.pad_id, self._ids[:n], self._lens[:n], stream)
        return self.encoder.encode_tokens(self._ids[:n], self._lens[:n], batch_size=self.batch_size, bucket=self.bucket)
