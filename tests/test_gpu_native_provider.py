"""GPU tests of the built-in recompute provider (csrc/lm_recompute.hip: ids -> token store -> packed forward, library code called by the
search loop directly) against the Python provider (leann_amd/recompute.py: RecomputeProvider.__call__) and the oracle.  What the
reference does at this point: one ZMQ round trip to the embedding server per hop (hnsw_embedding_server.py:148-284).  Everything goes
through the C ABI; the same scenarios run on the CPU box in thread-per-lane emulation (tests/emulated_search_cases.py: native_recompute)."""
import os
from unittest import mock

import numpy as np
import pytest

from tests.util import oracle_graph, recall_at_k

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def world():
    import torch

    from leann_amd import _lib
    from leann_amd.encoder import BertEncoder, EncoderConfig
    from leann_amd.synth import CorpusSpec, SyntheticCorpus
    from leann_amd.token_store import TokenStore

    _lib.require_gpu()
    n = 6000
    c = SyntheticCorpus(CorpusSpec(n_chunks=n, n_topics=8))
    tok, off = c.chunks()
    ts = TokenStore(tok, off)
    cfg = EncoderConfig(vocab_size=30522, hidden=384, layers=2, heads=12, ffn=1536, max_pos=512, max_seq_length=256)
    enc = BertEncoder.random_init(cfg, seed=11).to("cuda", dtype=torch.float16).eval()
    return {"torch": torch, "corpus": c, "tokens": ts, "enc": enc, "n": n, "off": off}


def _providers(w, **kw):
    from leann_amd.recompute import RecomputeProvider

    dev = w["torch"].device("cuda")
    return RecomputeProvider(w["enc"], w["tokens"], 384, dev, **kw), RecomputeProvider(w["enc"], w["tokens"], 384, dev, **kw)


def _python_form():
    return mock.patch.dict(os.environ, {"LEANN_MI355X_NATIVE_PROVIDER": "0"})


@pytest.mark.parametrize("batch_size", [5461, 8])
def test_native_embeddings_bit_identical_to_the_python_provider(world, batch_size):
    """Same token batches into the same kernels: one forward (batch_size 5461 = 1M tokens) and many sub-batched forwards
    (batch_size 8 = 1536 tokens per forward), ragged id list with repeats, the longest and the shortest chunk."""
    torch = world["torch"]
    nat, py = _providers(world, batch_size=batch_size)
    assert nat.native() is not None
    lens = np.diff(world["off"].astype(np.int64))
    rng = np.random.default_rng(0)
    ids = np.concatenate([rng.choice(world["n"], 700, replace=False), [int(lens.argmax()), int(lens.argmin()), 0, 0, world["n"] - 1]]).astype(np.int32)
    d_ids = torch.from_numpy(ids).cuda()
    a = nat.embed_ids(d_ids)
    with _python_form():
        assert py.native() is None
        b = py.embed_ids(d_ids)
    torch.cuda.synchronize()
    assert a.shape == b.shape == (ids.shape[0], 384)
    assert torch.equal(a, b), float((a - b).abs().max())
    st = nat.native_stats()
    assert st["chunks"] == ids.shape[0] and st["tokens"] == int(np.minimum(lens[ids], 256).sum())
    assert (st["forwards"] == 1) == (batch_size == 5461)
    # a one-chunk call (a SMALL forward: the general kernels, whatever form the big call above took -- same bits as the Python provider's
    # one-chunk call, fp16-close to the row of the big call) and an empty call
    one = nat.embed_ids(d_ids[:1])
    with _python_form():
        one_py = py.embed_ids(d_ids[:1])
    assert torch.equal(one, one_py)
    assert float((one - a[:1]).abs().max()) < 3e-3
    assert nat.embed_ids(d_ids[:0]).shape == (0, 384)
    nat.close()


def _search_native_vs_python_vs_oracle(torch, nat, py, g, Q, dim, ef, beam, memo):
    """Search over the library-side provider == search over the Python provider (labels, distances, evaluation and recompute counts:
    both hand the same token batches to the same kernels) == the oracle replaying the Python provider's own per-round outputs (the
    comparison that is exact whatever kernel form a forward of a given size takes; a table of embeddings computed in ONE big forward
    is only fp16-close to what a round's small forward returns).  Returns (labels, nunique, provider sync counters before / after)."""
    from leann_amd.devmem import as_tensor
    from leann_amd.index import Mi355xIndex
    from oracle import oracle as orc

    idx = Mi355xIndex.from_csr(g)
    idx.set_stream(torch.cuda.current_stream().cuda_stream)
    idx.set_provider(nat)
    assert idx.native_provider
    prm = idx.make_params(ef=ef, beam=beam, recompute=True, recompute_memo=memo)
    idx.search_device(Q, 10, prm)  # first call grows the provider's buffers
    s0 = nat.native_stats()
    gd, gi = idx.search_device(Q, 10, prm)
    torch.cuda.synchronize()
    st, s1 = idx.stats(), nat.native_stats()
    assert s1["chunks"] - s0["chunks"] == int(st["nunique"])
    rounds = []

    def recording(d_ids, cnt, stream):
        p = py(d_ids, cnt, stream)
        torch.cuda.synchronize()
        rounds.append((as_tensor(d_ids, (cnt,), "int32").cpu().numpy().copy(), as_tensor(p, (cnt, dim), "float32").cpu().numpy().copy()))
        return p

    idx.set_provider(recording)
    assert not idx.native_provider
    pd_, pi = idx.search_device(Q, 10, prm)
    torch.cuda.synchronize()
    st_py = idx.stats()
    assert torch.equal(pi, gi) and torch.equal(pd_, gd)
    assert int(st_py["nunique"]) == int(st["nunique"]) and int(st_py["ndis"]) == int(st["ndis"]) and int(st_py["nrounds"]) == int(st["nrounds"])
    it = iter(rounds)

    def replay(idv):
        ids, emb = next(it)
        assert np.array_equal(ids, idv)
        return emb

    oi, od, ost = orc.search(oracle_graph(g, dim), Q.cpu().numpy(), 10, ef=ef, beam=beam, provider=replay, memo=bool(memo) and Q.shape[0] > 1)
    assert np.array_equal(gi.cpu().numpy(), oi) and np.array_equal(gd.cpu().numpy(), od) and int(st["ndis"]) == int(ost["ndis"])
    idx.close()
    return oi, int(st["nunique"]), s0, s1


@pytest.mark.parametrize("nq,memo", [(1, True), (24, True), (24, False)])
def test_search_over_the_native_provider_equals_python_provider_and_oracle(world, nq, memo):
    """Recompute-mode search with the library-side provider attached (lm_index_set_recompute): labels, distances, evaluation and
    recompute counts equal the search over the Python provider and the oracle (replay of the per-round embeddings); the provider adds
    no host synchronisation of its own (the search loop's per-round copy carries the token counts)."""
    torch = world["torch"]
    from leann_amd.gpu_graph_build import build_graph_gpu
    from leann_amd.recompute import RecomputeProvider
    from leann_amd.token_store import TokenStore
    from oracle import oracle as orc

    nat, py = _providers(world)
    n = world["n"]
    X = nat.embed_ids(torch.arange(n, dtype=torch.int32, device="cuda"))
    g = build_graph_gpu(X, "mips", M=12, ef_construction=60)
    qt, qo, _ = world["corpus"].queries(nq)
    qs = TokenStore(qt, qo)
    Q = RecomputeProvider(world["enc"], qs, 384, torch.device("cuda")).embed_ids(torch.arange(nq, dtype=torch.int32, device="cuda"))
    oi, nunique, s0, s1 = _search_native_vs_python_vs_oracle(torch, nat, py, g, Q, 384, 40, 2, memo)
    assert s1["host_syncs"] == s0["host_syncs"], (s0, s1)
    assert py.chunks == nunique
    gt, _ = orc.bruteforce_topk(X.cpu().numpy(), Q.cpu().numpy(), 10, 0)
    assert recall_at_k(oi, gt) > 0.9
    nat.close()


def test_speculative_prefetch_over_the_native_provider(world):
    """Option "speculate" with the real encoder behind the search: a one-query search asks the built-in provider for more chunks in fewer
    forwards.  With a table-lookup provider the results are those of S = 0 bit for bit (tests/test_gpu_parity.py); with the encoder a
    chunk's embedding can come from a different launch form (a prefetching round may exceed the small-forward limit of 8192 tokens and
    take the fused kernels): fp16-close distances, and labels that may differ only where two candidates are that close."""
    torch = world["torch"]
    from leann_amd.gpu_graph_build import build_graph_gpu
    from leann_amd.index import Mi355xIndex
    from leann_amd.recompute import RecomputeProvider
    from leann_amd.token_store import TokenStore

    nat, _ = _providers(world)
    n = world["n"]
    X = nat.embed_ids(torch.arange(n, dtype=torch.int32, device="cuda"))
    g = build_graph_gpu(X, "mips", M=12, ef_construction=60)
    qt, qo, _ = world["corpus"].queries(4)
    Q = RecomputeProvider(world["enc"], TokenStore(qt, qo), 384, torch.device("cuda")).embed_ids(torch.arange(4, dtype=torch.int32, device="cuda"))
    idx = Mi355xIndex.from_csr(g)
    idx.set_stream(torch.cuda.current_stream().cuda_stream)
    idx.set_provider(nat)
    assert idx.native_provider
    out = {}
    for S in (0, 8):
        idx.set_option("speculate", S)
        f0 = nat.native_stats()["forwards"]
        res = [idx.search_device(Q[i : i + 1].contiguous(), 10, idx.make_params(ef=48, beam=1, recompute=True)) for i in range(4)]
        torch.cuda.synchronize()
        out[S] = ([r[1].cpu().numpy() for r in res], [r[0].cpu().numpy() for r in res], nat.native_stats()["forwards"] - f0)
    for i in range(4):
        assert len(set(out[0][0][i].ravel().tolist()) & set(out[8][0][i].ravel().tolist())) >= 9, i
        assert np.abs(out[0][1][i] - out[8][1][i]).max() < 3e-3
    assert out[8][2] < out[0][2], (out[8][2], out[0][2])
    idx.close()
    nat.close()


def test_native_provider_behind_the_pq_traversal_and_as_plain_provider_fn(world):
    """The provider interface has other callers: the DiskANN-style traversal's deferred rerank (lm_pq_batch_search) and any host that
    passes lm_recompute_provider to lm_index_set_provider itself.  Both equal the Python provider's results."""
    import ctypes as C

    torch = world["torch"]
    from leann_amd import _lib
    from leann_amd.gpu_graph_build import build_graph_gpu
    from leann_amd.index import Mi355xIndex
    from leann_amd.pq import encode_pq, train_pq

    nat, py = _providers(world)
    n = world["n"]
    X = nat.embed_ids(torch.arange(n, dtype=torch.int32, device="cuda"))
    g = build_graph_gpu(X, "mips", M=12, ef_construction=60)
    cb = train_pq(X, 48, iters=4, seed=0)
    codes = encode_pq(X, cb)
    Q = X[:16] + 0.02 * torch.randn((16, 384), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    Q = Q.contiguous()

    def pq_run(provider):
        idx = Mi355xIndex.from_csr(g)
        idx.set_stream(torch.cuda.current_stream().cuda_stream)
        idx.attach_pq(cb.cpu().numpy(), codes.cpu().numpy())
        idx.set_provider(provider)
        out = idx.pq_search_device(Q, 10, idx.make_pq_params(complexity=48, beam_width=4, use_deferred_fetch=True))
        torch.cuda.synchronize()
        res = (out[0].clone(), out[1].clone(), idx.native_provider)
        idx.close()
        return res

    d1, l1, was_native = pq_run(nat)
    assert was_native
    with _python_form():
        d2, l2, was_native2 = pq_run(py)
    assert not was_native2
    assert torch.equal(l1, l2) and torch.equal(d1, d2)

    lib = _lib.load()
    idx = Mi355xIndex.from_csr(g)
    idx.set_stream(torch.cuda.current_stream().cuda_stream)
    fn = C.cast(lib.lm_recompute_provider, _lib.PROVIDER_FN)
    _lib.check(lib.lm_index_set_provider(idx._h, fn, nat.native()), "lm_index_set_provider")
    prm = idx.make_params(ef=32, beam=1, recompute=True)
    a = idx.search_device(Q, 10, prm)
    idx.set_provider(nat)
    b = idx.search_device(Q, 10, prm)
    torch.cuda.synchronize()
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    idx.close()
    nat.close()


@pytest.mark.parametrize("hidden,heads,ffn,pooling", [(768, 12, 3072, "cls"), (768, 12, 3072, "mean"), (384, 12, 1536, "cls")])
def test_native_provider_other_model_shapes(world, hidden, heads, ffn, pooling):
    """bge-base / contriever shape (hidden 768, head_dim 64: lm_bert_forward_packed behind lm_recompute_create_general) and bge-small
    (hidden 384, CLS pooling: the fused kernels + the CLS pooling kernel), two layers deep: the library-side provider, the Python
    provider's one-call forward and its per-kernel launch path return the same bits; close to the same weights in fp32 on the CPU;
    a recompute-mode search over the library-side provider equals the one over the Python provider and the oracle."""
    torch = world["torch"]
    from leann_amd.encoder import BertEncoder, EncoderConfig
    from leann_amd.gpu_graph_build import build_graph_gpu
    from leann_amd.recompute import RecomputeProvider
    from leann_amd.synth import pad_batch

    dev = torch.device("cuda")
    cfg = EncoderConfig(vocab_size=30522, hidden=hidden, layers=2, heads=heads, ffn=ffn, max_pos=512, max_seq_length=256, pooling=pooling)
    enc = BertEncoder.random_init(cfg, seed=5).to("cuda", dtype=torch.float16).eval()
    n = 1500
    ids = torch.arange(n, dtype=torch.int32, device="cuda")
    nat = RecomputeProvider(enc, world["tokens"], hidden, dev, batch_size=64)  # 12k tokens per forward: sub-batched
    py = RecomputeProvider(enc, world["tokens"], hidden, dev, batch_size=64)
    assert nat.native() is not None
    a = nat.embed_ids(ids)
    with _python_form():
        b = py.embed_ids(ids)
        with mock.patch.dict(os.environ, {"LEANN_MI355X_ONECALL": "0"}):
            c = py.embed_ids(ids)
    torch.cuda.synchronize()
    assert a.shape == (n, hidden) and torch.equal(a, b) and torch.equal(b, c), (float((a - b).abs().max()), float((b - c).abs().max()))
    assert nat.native_stats()["forwards"] > 1
    tok, off = world["corpus"].chunks()
    pi, pl = pad_batch(tok, off, 256)
    with torch.no_grad():
        ref = BertEncoder.random_init(cfg, seed=5).eval()(torch.from_numpy(pi[:200]), torch.from_numpy(pl[:200])).float()
    assert float((a[:200].cpu() - ref).abs().max()) < 8e-3
    # search over the library-side provider == over the Python provider == the oracle (replay)
    g = build_graph_gpu(a, "mips", M=10, ef_construction=50)
    Q = (a[:12] + 0.03 * torch.randn((12, hidden), device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))).contiguous()
    _search_native_vs_python_vs_oracle(torch, nat, py, g, Q, hidden, 32, 1, True)
    nat.close()


def test_native_provider_declines_outside_its_envelope(world):
    """A width the general kernels do not take (hidden 64) / fp32 weights / per-kernel timers on: native() is None and the Python
    provider runs."""
    torch = world["torch"]
    from leann_amd.encoder import BertEncoder, EncoderConfig, KernelTimers
    from leann_amd.recompute import RecomputeProvider

    dev = torch.device("cuda")
    cfg = EncoderConfig(vocab_size=30522, hidden=64, layers=1, heads=4, ffn=128, max_pos=512, max_seq_length=256)
    enc64 = BertEncoder.random_init(cfg, seed=2).to("cuda", dtype=torch.float16).eval()
    assert RecomputeProvider(enc64, world["tokens"], 64, dev).native() is None
    enc32 = BertEncoder.random_init(world["enc"].cfg, seed=2).to("cuda").eval()  # fp32 weights
    assert RecomputeProvider(enc32, world["tokens"], 384, dev).native() is None
    p = RecomputeProvider(world["enc"], world["tokens"], 384, dev)
    KernelTimers.active = KernelTimers()
    try:
        assert p.native() is None
    finally:
        KernelTimers.active = None
    assert p.native() is not None
    p.close()
