"""Deterministic synthetic text corpora (token-id chunks) for benchmarks and parity tests.

BASELINE.json's configs name "1M synthetic text chunks"; there is no network for real datasets or
checkpoints, so chunks are drawn from a seeded hierarchical topic model (topic -> document ->
chunk, Zipfian word frequencies) that gives the corpus the clustered, low-intrinsic-dimension
structure of real text: chunks of one document share vocabulary, documents of one topic share a
topic vocabulary.  Lengths follow SURVEY 8(d): clip(round(Normal(180, 50)), 16, 256), framed by
[CLS]=101 / [SEP]=102, body ids in [1000, vocab).
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np

CLS_ID, SEP_ID = 101, 102


@dataclass(frozen=True)
class CorpusSpec:
    n_chunks: int
    seed: int = 1234
    vocab_size: int = 30522
    n_topics: int = 1000
    chunks_per_doc: int = 16
    topic_vocab: int = 512
    doc_vocab: int = 48
    p_topic: float = 0.45
    p_doc: float = 0.35  # remainder: global background
    len_mean: float = 180.0
    len_std: float = 50.0
    len_min: int = 16
    len_max: int = 256
    zipf_a: float = 1.1


def _zipf_ranks(rng: np.random.Generator, n: int, size: int, a: float) -> np.ndarray:
    """n draws of a rank in [0, size) with P(r) ~ 1/(r+1)^a (inverse-CDF on a precomputed table)."""
    w = 1.0 / np.power(np.arange(1, size + 1, dtype=np.float64), a)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    return np.searchsorted(cdf, rng.random(n), side="left").astype(np.int32)


class SyntheticCorpus:
    """Topic tables are a function of (spec.seed, vocab, topics); chunks / queries are generated in
    blocks so that 1M chunks need only a few hundred MB of temporaries."""

    def __init__(self, spec: CorpusSpec):
        self.spec = spec
        rng = np.random.default_rng(spec.seed)
        lo = 1000
        self.topic_words = rng.integers(lo, spec.vocab_size, (spec.n_topics, spec.topic_vocab), dtype=np.int32)
        self.n_docs = (spec.n_chunks + spec.chunks_per_doc - 1) // spec.chunks_per_doc
        self.doc_topic = rng.integers(0, spec.n_topics, self.n_docs, dtype=np.int32)
        self.doc_seed = rng.integers(0, 2**31 - 1, self.n_docs, dtype=np.int64)

    def _doc_words(self, docs: np.ndarray) -> np.ndarray:
        # per-document private vocabulary, derived from the document's own seed (order independent)
        s = self.spec
        x = self.doc_seed[docs][:, None] * 6364136223846793005 + np.arange(s.doc_vocab, dtype=np.int64)[None, :] * 1442695040888963407
        x ^= x >> 29
        x *= 0xBF58476D1CE4E5B9 & 0x7FFFFFFFFFFFFFFF
        x ^= x >> 32
        return (1000 + (np.abs(x) % (s.vocab_size - 1000))).astype(np.int32)

    def _make(self, docs: np.ndarray, rng: np.random.Generator):
        """One chunk per entry of ``docs`` -> (tokens u16 flat, offsets u64)."""
        s = self.spec
        n = docs.shape[0]
        lens = np.clip(np.rint(rng.normal(s.len_mean, s.len_std, n)), s.len_min, s.len_max).astype(np.int64)
        off = np.zeros(n + 1, np.uint64)
        off[1:] = np.cumsum(lens)
        total = int(off[-1])
        chunk_of = np.repeat(np.arange(n, dtype=np.int64), lens)
        u = rng.random(total)
        rt = _zipf_ranks(rng, total, s.topic_vocab, s.zipf_a)
        rd = _zipf_ranks(rng, total, s.doc_vocab, s.zipf_a)
        rb = _zipf_ranks(rng, total, s.vocab_size - 1000, s.zipf_a) + 1000
        dw = self._doc_words(docs)
        tok = np.where(u < s.p_topic, self.topic_words[self.doc_topic[docs][chunk_of], rt],
                       np.where(u < s.p_topic + s.p_doc, dw[chunk_of, rd], rb)).astype(np.uint16)
        starts = off[:-1].astype(np.int64)
        tok[starts] = CLS_ID
        tok[starts + lens - 1] = SEP_ID
        return tok, off

    def chunks(self, block: int = 65536):
        """Generate the corpus: (tokens u16[total], offsets u64[n+1]); chunk i belongs to doc i // chunks_per_doc."""
        s = self.spec
        toks, offs, base = [], [np.zeros(1, np.uint64)], 0
        for b0 in range(0, s.n_chunks, block):
            b1 = min(s.n_chunks, b0 + block)
            rng = np.random.default_rng([s.seed, 1, b0])
            t, o = self._make(np.arange(b0, b1, dtype=np.int64) // s.chunks_per_doc, rng)
            toks.append(t)
            offs.append(o[1:] + np.uint64(base))
            base += int(o[-1])
        return np.concatenate(toks), np.concatenate(offs)

    def chunks_torch(self, device, block: int = 1 << 20):
        """The same generative model evaluated with torch ops on ``device`` (10M-60M chunk configurations: the numpy path above
        takes ~20 s per million chunks on the host).  Same topic tables and per-document vocabularies, the random draws come
        from torch's generator (seeded per block), so the corpus is deterministic but NOT bit-identical to ``chunks()``.
        Returns (tokens u16[total], offsets u64[n+1]) as numpy arrays."""
        import torch

        s = self.spec
        dev = torch.device(device)
        tw = torch.from_numpy(self.topic_words).to(dev)
        dt = torch.from_numpy(self.doc_topic).to(dev).long()
        ds = torch.from_numpy(self.doc_seed).to(dev)

        def cdf(size):
            w = 1.0 / torch.arange(1, size + 1, dtype=torch.float64, device=dev) ** s.zipf_a
            c = torch.cumsum(w, 0)
            return (c / c[-1]).float()

        c_t, c_d, c_b = cdf(s.topic_vocab), cdf(s.doc_vocab), cdf(s.vocab_size - 1000)
        toks, lens_all = [], []
        for b0 in range(0, s.n_chunks, block):
            b1 = min(s.n_chunks, b0 + block)
            g = torch.Generator(device=dev).manual_seed(int(s.seed) * 1_000_003 + b0)
            docs = torch.arange(b0, b1, device=dev) // s.chunks_per_doc
            lens = torch.clamp(torch.round(torch.randn(b1 - b0, generator=g, device=dev) * s.len_std + s.len_mean), s.len_min, s.len_max).long()
            off = torch.zeros(b1 - b0 + 1, dtype=torch.int64, device=dev)
            off[1:] = torch.cumsum(lens, 0)
            total = int(off[-1])
            chunk_of = torch.repeat_interleave(torch.arange(b1 - b0, device=dev), lens, output_size=total)
            u = torch.rand(total, generator=g, device=dev)
            rt = torch.searchsorted(c_t, torch.rand(total, generator=g, device=dev)).clamp(max=s.topic_vocab - 1)
            rd = torch.searchsorted(c_d, torch.rand(total, generator=g, device=dev)).clamp(max=s.doc_vocab - 1)
            rb = torch.searchsorted(c_b, torch.rand(total, generator=g, device=dev)).clamp(max=s.vocab_size - 1001) + 1000
            d_of = docs[chunk_of]
            # per-document private word (same hash as _doc_words, evaluated per token)
            x = ds[d_of] * 6364136223846793005 + rd * 1442695040888963407
            x = x ^ (x >> 29)
            x = x * (0xBF58476D1CE4E5B9 & 0x7FFFFFFFFFFFFFFF)
            x = x ^ (x >> 32)
            dw = 1000 + (x.abs() % (s.vocab_size - 1000))
            t = torch.where(u < s.p_topic, tw[dt[d_of], rt].long(), torch.where(u < s.p_topic + s.p_doc, dw, rb))
            t[off[:-1]] = CLS_ID
            t[off[1:] - 1] = SEP_ID
            toks.append(t.to(torch.int32).cpu().numpy().astype(np.uint16))
            lens_all.append(lens.cpu().numpy())
        lens_np = np.concatenate(lens_all)
        offs = np.zeros(s.n_chunks + 1, np.uint64)
        offs[1:] = np.cumsum(lens_np)
        return np.concatenate(toks), offs

    def queries(self, n: int, seed: int = 4321):
        """Held-out chunks of randomly chosen existing documents."""
        rng = np.random.default_rng([self.spec.seed, 2, seed])
        docs = rng.integers(0, self.n_docs, n, dtype=np.int64)
        return self._make(docs, rng) + (docs,)


def pad_batch(tok: np.ndarray, off: np.ndarray, T: int, pad_id: int = 0):
    """(tokens, offsets) -> dense int32 [n, T] + lengths (host side helper for tests)."""
    n = off.shape[0] - 1
    lens = np.minimum(np.diff(off.astype(np.int64)), T).astype(np.int32)
    out = np.full((n, T), pad_id, np.int32)
    for i in range(n):
        b = int(off[i])
        out[i, : lens[i]] = tok[b : b + lens[i]]
    return out, lens
