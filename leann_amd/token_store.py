"""HBM-resident pre-tokenised passage store (Python handle on ``lm_tokens``).

Replaces the per-id ``PassageManager.get_passage`` + tokeniser work the reference does inside the
embedding server for every hop (leann/api.py:203-215, leann/embedding_compute.py:229-239).
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check


class TokenStore:
    def __init__(self, tokens: np.ndarray, offsets: np.ndarray, device: int = 0):
        """``tokens``: u16[total] packed token ids; ``offsets``: u64[n+1]."""
        self._lib = _lib.load()
        tokens = np.asarray(tokens)
        if tokens.dtype != np.uint16 and tokens.size and (int(tokens.max()) > 0xFFFF or int(tokens.min()) < 0):
            raise ValueError(f"token id {int(tokens.max())} does not fit the u16 token store (vocabularies up to 65536 entries)")
        tok = np.ascontiguousarray(tokens, dtype=np.uint16)
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        if off.ndim != 1 or off.shape[0] < 1 or int(off[-1]) != tok.shape[0]:
            raise ValueError("offsets[-1] must equal len(tokens)")
        self.n = off.shape[0] - 1
        self.max_len = int(np.diff(off.astype(np.int64)).max()) if self.n else 0
        self._h = C.c_void_p()
        check(self._lib.lm_tokens_create(tok.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), self.n, device,
                                         C.byref(self._h)), "lm_tokens_create")

    @classmethod
    def from_lists(cls, seqs, device: int = 0) -> "TokenStore":
        lens = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=len(seqs))
        off = np.zeros(len(seqs) + 1, np.uint64)
        off[1:] = np.cumsum(lens)
        tok = np.fromiter((t for s in seqs for t in s), dtype=np.int64, count=int(off[-1]))  # range-checked in __init__
        return cls(tok, off, device)

    def gather(self, d_ids_ptr: int, n: int, T: int, pad_id: int, out_ids, out_len, stream_ptr: int = 0) -> None:
        """ids (device int32[n]) -> out_ids (torch int32 [>=n, T]) / out_len (torch int32 [>=n])."""
        check(self._lib.lm_tokens_gather(self._h, C.c_void_p(d_ids_ptr), n, T, pad_id, C.c_void_p(out_ids.data_ptr()),
                                         C.c_void_p(out_len.data_ptr()), C.c_void_p(stream_ptr)), "lm_tokens_gather")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.lm_tokens_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
