"""GPU tests of the pieces around the search kernel: native index reader, token gather kernel,
in-process recompute provider (tokens -> BERT -> embeddings), the backend plugin end to end, and
the GPU graph builder.  Everything goes through the C ABI."""
import ctypes as C
import json
from pathlib import Path

import numpy as np
import pytest

from tests.util import clustered, oracle_graph, queries_near, recall_at_k

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden"


@pytest.fixture(scope="module")
def torch_():
    import torch

    from leann_amd import _lib

    _lib.require_gpu()
    return torch


def test_native_reader_matches_python_reader(torch_):
    """lm_index_read on the fixtures written by the reference's converter."""
    from leann_amd import csr_format as cf
    from leann_amd.index import Mi355xIndex
    from oracle import oracle as orc

    for name in ("ref_csr_pruned.index", "ref_csr_full.index", "ref_original.index"):
        g = cf.read_index(G / name)
        idx = Mi355xIndex.read(str(G / name))
        assert idx.info.ntotal == g.ntotal and idx.info.d == g.d and idx.info.metric == g.metric_type
        assert idx.info.entry_point == g.entry_point and idx.info.max_level == g.max_level
        assert idx.info.n_neighbors == g.neighbors.shape[0]
        assert bool(idx.info.has_table) == (g.storage is not None)
        if g.storage is not None:  # stored embeddings were attached by the reader: search works straight away
            q = g.storage[:5] + 0.01
            d, l = idx.search(q, 4, idx.make_params(ef=16, recompute=False))
            oi, od, _ = orc.search(oracle_graph(g, g.d), q, 4, ef=16, table=g.storage)
            assert np.array_equal(l, oi) and np.array_equal(d, od)
        else:
            with pytest.raises(RuntimeError):
                idx.search(np.zeros((1, g.d), np.float32), 2, idx.make_params(ef=8, recompute=False))
            with pytest.raises(RuntimeError):  # recompute without a provider
                idx.search(np.zeros((1, g.d), np.float32), 2, idx.make_params(ef=8, recompute=True))
        idx.close()


def test_token_gather_kernel(torch_):
    torch = torch_
    from leann_amd.synth import CorpusSpec, SyntheticCorpus, pad_batch
    from leann_amd.token_store import TokenStore

    c = SyntheticCorpus(CorpusSpec(n_chunks=3000, n_topics=8))
    tok, off = c.chunks()
    ts = TokenStore(tok, off)
    ids = torch.tensor([0, 2999, 17, 17, 1234, 5], dtype=torch.int32, device="cuda")
    for T in (256, 64, 300):
        out = torch.empty((ids.shape[0], T), dtype=torch.int32, device="cuda")
        lens = torch.empty((ids.shape[0],), dtype=torch.int32, device="cuda")
        ts.gather(ids.data_ptr(), ids.shape[0], T, 0, out, lens, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        exp, el = pad_batch(tok, off, T)
        assert np.array_equal(out.cpu().numpy(), exp[ids.cpu().numpy()])
        assert np.array_equal(lens.cpu().numpy(), el[ids.cpu().numpy()])


def _tiny_encoder(torch, dim=64, pooling="mean"):
    from leann_amd.encoder import BertEncoder, EncoderConfig

    cfg = EncoderConfig(vocab_size=30522, hidden=dim, layers=2, heads=4, ffn=128, max_pos=256, max_seq_length=256, pooling=pooling)
    return BertEncoder.random_init(cfg, seed=3)


def test_encoder_gpu_matches_cpu_fp32_and_fp16(torch_):
    """Encoder drift CPU fp32 <-> GPU fp32 <-> GPU fp16 stays within the north_star distance tolerance
    budget for the embedding side (distances within 1e-4 are asserted on equal embedding bits; here we
    bound the encoder's own drift)."""
    torch = torch_
    from leann_amd.synth import CorpusSpec, SyntheticCorpus, pad_batch

    enc = _tiny_encoder(torch)
    c = SyntheticCorpus(CorpusSpec(n_chunks=300, n_topics=4))
    ids, lens = pad_batch(*c.chunks(), 256)
    ti, tl = torch.from_numpy(ids), torch.from_numpy(lens)
    with torch.no_grad():
        ref = enc.encode_tokens(ti, tl, batch_size=64)
        g32 = enc.to("cuda").encode_tokens(ti.cuda(), tl.cuda(), batch_size=64).cpu()
        g16 = enc.to("cuda", dtype=torch.float16).encode_tokens(ti.cuda(), tl.cuda(), batch_size=64).cpu()
    assert (ref - g32).abs().max() < 2e-5
    assert (ref - g16).abs().max() < 5e-3


@pytest.mark.parametrize("dim,pooling", [(64, "mean"), (768, "cls")])
def test_recompute_search_end_to_end_bit_exact_given_same_embeddings(torch_, dim, pooling):
    """Full hot path (dim 768 + CLS pooling = the bge-base shape of config C5, shallow): ids -> HBM token gather -> BERT -> fused distance/beam update.  The oracle
    consumes the GPU encoder's OWN outputs (captured per round), so traversal + distances must be
    bit-exact (hard part 2d)."""
    torch = torch_
    from leann_amd.gpu_graph_build import build_graph_gpu
    from leann_amd.index import Mi355xIndex
    from leann_amd.recompute import RecomputeProvider
    from leann_amd.synth import CorpusSpec, SyntheticCorpus
    from leann_amd.token_store import TokenStore
    from oracle import oracle as orc

    n = 4000
    c = SyntheticCorpus(CorpusSpec(n_chunks=n, n_topics=8))
    tok, off = c.chunks()
    ts = TokenStore(tok, off)
    enc = _tiny_encoder(torch, dim, pooling).to("cuda", dtype=torch.float16)
    prov = RecomputeProvider(enc, ts, dim, torch.device("cuda"), batch_size=512)
    X = prov.embed_ids(torch.arange(n, dtype=torch.int32, device="cuda"))
    g = build_graph_gpu(X, "mips", M=12, ef_construction=60)
    g.validate()
    qt, qo, _ = c.queries(24)
    qs = TokenStore(qt, qo)
    Q = RecomputeProvider(enc, qs, dim, torch.device("cuda")).embed_ids(torch.arange(24, dtype=torch.int32, device="cuda"))
    idx = Mi355xIndex.from_csr(g)
    idx.set_stream(torch.cuda.current_stream().cuda_stream)
    rounds = []

    def recording(d_ids, cnt, stream):
        from leann_amd.devmem import as_tensor

        p = prov(d_ids, cnt, stream)
        rounds.append((as_tensor(d_ids, (cnt,), "int32").cpu().numpy().copy(), as_tensor(p, (cnt, dim), "float32").cpu().numpy().copy()))
        return p

    idx.set_provider(recording)
    gd, gi = idx.search_device(Q, 10, idx.make_params(ef=48, beam=2, recompute=True))
    torch.cuda.synchronize()
    it = iter(rounds)

    def replay(idv):
        ids, emb = next(it)
        assert np.array_equal(ids, idv)
        return emb

    # (the library default keeps a per-call recompute memo for a call of more than one query: the oracle restates it, oracle.py: memo=)
    oi, od, _ = orc.search(oracle_graph(g, dim), Q.cpu().numpy(), 10, ef=48, beam=2, provider=replay, memo=True)
    assert np.array_equal(gi.cpu().numpy(), oi) and np.array_equal(gd.cpu().numpy(), od)
    # and the recompute path finds what the stored-embedding path finds (recall sanity)
    gt, _ = orc.bruteforce_topk(X.cpu().numpy(), Q.cpu().numpy(), 10, 0)
    assert recall_at_k(oi, gt) > 0.9
    assert prov.chunks == idx.stats()["nunique"]


def test_backend_plugin_end_to_end(torch_, tmp_path):
    """Drive the backend exactly the way LeannSearcher.search does (leann/api.py:703-744):
    _ensure_server_running -> compute_query_embedding -> search(recompute_embeddings=True, zmq_port=...)."""
    torch = torch_
    from leann_amd._compat import BACKEND_REGISTRY
    from leann_amd.backend import write_leann_bundle

    words = ["alpha", "beta", "gamma", "delta", "epsilon", "zeta", "eta", "theta", "iota", "kappa", "lambda", "mu"]
    rng = np.random.default_rng(0)
    texts = [" ".join(rng.choice(words, size=int(rng.integers(5, 30))).tolist()) + f" doc{i}" for i in range(300)]
    index_path = str(tmp_path / "demo.leann")
    factory = BACKEND_REGISTRY["mi355x"]
    # embeddings for the build come from the same in-process encoder (as LeannBuilder.build_index does
    # with compute_embeddings, leann/api.py:440-446)
    from leann_amd.encoder import BertEncoder
    from leann_amd.tokenizer import load_tokenizer

    model = "sentence-transformers/all-MiniLM-L6-v2"
    enc = BertEncoder.load(model, allow_random=True).to("cuda", dtype=torch.float16)
    tok = load_tokenizer(model, 256, index_path, texts, enc.cfg.vocab_size, allow_stand_in=enc.weights_source == "random")
    seqs = tok.encode_batch(texts)
    T = max(len(s) for s in seqs)
    ids = torch.zeros((len(seqs), T), dtype=torch.int32)
    for i, s in enumerate(seqs):
        ids[i, : len(s)] = torch.tensor(s, dtype=torch.int32)
    lens = torch.tensor([len(s) for s in seqs], dtype=torch.int32)
    emb = enc.encode_tokens(ids.cuda(), lens.cuda()).cpu().numpy()
    write_leann_bundle(index_path, texts, emb, model, distance_metric="mips", M=8, efConstruction=40)
    meta = json.loads(Path(index_path + ".meta.json").read_text())
    assert meta["backend_name"] == "mi355x" and meta["is_pruned"] is True

    s = factory.searcher(index_path, allow_random_weights=True)  # the bundle was built with the seeded random encoder
    with pytest.raises(ValueError):
        s.search(emb[:1], 3, recompute_embeddings=True, zmq_port=None)
    with pytest.raises(RuntimeError):
        s.search(emb[:1], 3, recompute_embeddings=False)
    port = s._ensure_server_running(index_path + ".meta.json", 5557)
    assert port == 5557
    qe = s.compute_query_embedding(texts[42], use_server_if_available=True, zmq_port=port)
    assert qe.shape == (1, 384) and qe.dtype == np.float32
    r = s.search(qe, 5, complexity=32, beam_width=1, prune_ratio=0.0, recompute_embeddings=True,
                 pruning_strategy="global", zmq_port=port, batch_size=0)
    assert set(r) == {"labels", "distances"} and len(r["labels"]) == 1 and len(r["labels"][0]) == 5
    assert all(isinstance(l, str) for l in r["labels"][0])
    assert r["distances"].shape == (1, 5) and r["distances"].dtype == np.float32
    assert r["labels"][0][0] == "42"  # the query IS passage 42
    assert np.all(np.diff(r["distances"][0]) <= 0)  # +IP, best first
    # batch of queries, beam 2
    r2 = s.search(emb[:7], 3, complexity=16, beam_width=2, recompute_embeddings=True, zmq_port=port)
    assert [row[0] for row in r2["labels"]] == [str(i) for i in range(7)]
    s.cleanup()


def test_gpu_graph_builder_quality(torch_):
    torch = torch_
    from leann_amd.gpu_graph_build import build_graph_gpu
    from leann_amd.index import Mi355xIndex
    from oracle import oracle as orc

    x = clustered(20000, 96, 0, n_centers=200, sigma=0.5)
    q = queries_near(x, 200, 1)
    gt, _ = orc.bruteforce_topk(x, q, 10, 0)
    g = build_graph_gpu(torch.from_numpy(x).cuda(), "mips", M=16, ef_construction=100)
    g.validate()
    assert g.level0_degrees().max() <= 32
    idx = Mi355xIndex.from_csr(g)
    idx.attach_table(x)
    _, l = idx.search(q, 10, idx.make_params(ef=64, recompute=False))
    assert recall_at_k(l, gt) >= 0.97


@pytest.mark.parametrize("hidden", [64, 384, 768, 1024, 1536])
def test_fused_add_layernorm_kernel(torch_, hidden):
    """lm_add_layernorm_f16 vs a plain PyTorch fp32 reference of the same op."""
    torch = torch_
    import torch.nn as nn
    import torch.nn.functional as F

    from leann_amd.encoder import fused_add_layernorm

    g = torch.Generator(device="cuda").manual_seed(hidden)
    for rows in (1, 7, 1000, 4099):
        x = torch.randn((rows, hidden), generator=g, device="cuda").half()
        r = (3 * torch.randn((rows, hidden), generator=g, device="cuda")).half()
        ln = nn.LayerNorm(hidden, eps=1e-12).to("cuda", dtype=torch.float16)
        with torch.no_grad():
            ln.weight.copy_(torch.randn(hidden, generator=g, device="cuda"))
            ln.bias.copy_(torch.randn(hidden, generator=g, device="cuda"))
        ref = F.layer_norm(x.float() + r.float(), (hidden,), ln.weight.float(), ln.bias.float(), 1e-12)
        got = fused_add_layernorm(x, r, ln)
        assert got.dtype == torch.float16 and got.shape == x.shape
        assert (got.float() - ref).abs().max() <= 4e-3 * max(1.0, float(ref.abs().max()))
        ref1 = F.layer_norm(x.float(), (hidden,), ln.weight.float(), ln.bias.float(), 1e-12)
        assert (fused_add_layernorm(x, None, ln).float() - ref1).abs().max() <= 4e-3 * max(1.0, float(ref1.abs().max()))


def test_sharded_search_merge_on_gpu(torch_):
    """60M-chunk mode in miniature (single process): two disjoint shards searched on the GPU, per-shard
    top-k merged by the lm_topk_merge kernel == exact merge by the oracle; ids are global."""
    torch = torch_
    from leann_amd.distributed import ShardedSearch, hip_merge_fn, shard_bounds
    from leann_amd.hnsw_builder import build_hnsw
    from leann_amd.index import Mi355xIndex
    from oracle import oracle as orc

    x = clustered(6000, 64, 21)
    q = queries_near(x, 50, 22)
    parts = []
    for lo, hi in shard_bounds(6000, 2):
        g = build_hnsw(x[lo:hi], "mips", M=8, ef_construction=40)
        idx = Mi355xIndex.from_csr(g)
        idx.attach_table(x[lo:hi])
        d, l = idx.search_device(torch.from_numpy(q).cuda(), 10, idx.make_params(ef=64, recompute=False))
        l = torch.where(l >= 0, l + lo, l)
        parts.append((d, l))
        idx.close()
    ids = torch.stack([p[1] for p in parts])
    dist = torch.stack([p[0] for p in parts])
    oi, od = hip_merge_fn(ids, dist, 0)
    ei, ed = orc.merge_topk(ids.cpu().numpy(), dist.cpu().numpy(), 0)
    assert np.array_equal(oi.cpu().numpy(), ei) and np.array_equal(od.cpu().numpy(), ed)
    gt, _ = orc.bruteforce_topk(x, q, 10, 0)
    assert recall_at_k(ei, gt) > 0.95
    # world_size 1 ShardedSearch degenerates to a merge of one list
    g = build_hnsw(x, "mips", M=8, ef_construction=40)
    idx = Mi355xIndex.from_csr(g)
    idx.attach_table(x)
    ss = ShardedSearch(lambda qq, k: idx.search_device(qq, k, idx.make_params(ef=64, recompute=False)), id_base=0, metric=0)
    d1, i1 = ss.search(torch.from_numpy(q).cuda(), 10)
    d2, i2 = idx.search_device(torch.from_numpy(q).cuda(), 10, idx.make_params(ef=64, recompute=False))
    assert torch.equal(i1, i2) and torch.equal(d1, d2)


@pytest.mark.parametrize("heads,maxlen", [(12, 256), (12, 70), (4, 33), (2, 1)])
def test_fused_attention_hd32_kernel(torch_, heads, maxlen):
    """lm_attn_varlen_hd32_f16 vs a plain PyTorch fp32 reference of the same op (per sequence softmax(QK^T/sqrt d)V)."""
    torch = torch_
    from leann_amd.encoder import fused_attention_hd32

    g = torch.Generator(device="cpu").manual_seed(heads * 1000 + maxlen)
    lens = torch.randint(1, maxlen + 1, (37,), generator=g)
    lens[0], lens[-1] = maxlen, 1
    cu = torch.zeros(38, dtype=torch.int32)
    cu[1:] = torch.cumsum(lens, 0)
    tot, H = int(cu[-1]), heads * 32
    qkv = (torch.randn((tot, 3 * H), generator=g) * 1.5).half().cuda()
    out = fused_attention_hd32(qkv, cu.cuda(), heads, int(lens.max()))
    assert out is not None and out.shape == (tot, H)
    q3 = qkv.float().view(tot, 3, heads, 32)
    ref = torch.empty((tot, H), device="cuda")
    for i in range(37):
        a, b = int(cu[i]), int(cu[i + 1])
        q, k, v = (q3[a:b, j].transpose(0, 1) for j in range(3))  # [heads, L, 32]
        p = torch.softmax(q @ k.transpose(1, 2) / 32**0.5, dim=-1)
        ref[a:b] = (p @ v).transpose(0, 1).reshape(b - a, H)
    err = (out.float() - ref).abs().max().item()
    assert err < 4e-3, err


def test_encoder_packed_fused_attention_matches_library_attention(torch_, monkeypatch):
    torch = torch_
    from leann_amd.encoder import BertEncoder, config_for
    from leann_amd.synth import CorpusSpec, SyntheticCorpus, pad_batch

    enc = BertEncoder.random_init(config_for("all-MiniLM-L6-v2"), 0).to("cuda", dtype=torch.float16)
    ids, lens = pad_batch(*SyntheticCorpus(CorpusSpec(n_chunks=300, n_topics=4)).chunks(), 256)
    ti, tl = torch.from_numpy(ids).cuda(), torch.from_numpy(lens).cuda()
    a = enc.encode_tokens_packed(ti, tl)
    monkeypatch.setenv("LEANN_MI355X_ATTN", "0")
    b = enc.encode_tokens_packed(ti, tl)
    assert (a - b).abs().max() < 2e-3


def test_backend_no_recompute_mode_and_hub_cache_kwarg(torch_, tmp_path):
    """recompute_embeddings=False on an index that keeps its embeddings (is_recompute=False => non-compact,
    hnsw_backend.py:58-64): stored-embedding search, no encoder involved.  Then the hub_cache_ratio kwarg on a
    pruned index: same answers as plain recompute."""
    torch = torch_
    from leann_amd._compat import BACKEND_REGISTRY
    from leann_amd.backend import write_leann_bundle

    x = clustered(400, 384, 31)
    texts = [f"passage {i} " + " ".join(f"w{(i * 7 + j) % 50}" for j in range(12)) for i in range(400)]
    p = str(tmp_path / "full.leann")
    write_leann_bundle(p, texts, x, "sentence-transformers/all-MiniLM-L6-v2", distance_metric="l2", M=8, efConstruction=40,
                       is_recompute=False)
    s = BACKEND_REGISTRY["mi355x"].searcher(p)
    assert s.is_pruned is False
    r = s.search(x[:9] + 1e-4, 3, complexity=32, recompute_embeddings=False)
    assert [row[0] for row in r["labels"]] == [str(i) for i in range(9)]
    assert np.all(np.diff(r["distances"], axis=1) >= 0) and r["distances"][:, 0].max() < 1e-3  # squared L2 ascending
    s.cleanup()
    # pruned index + hub cache: embeddings come from the in-process encoder
    from leann_amd.encoder import BertEncoder
    from leann_amd.tokenizer import load_tokenizer

    model = "sentence-transformers/all-MiniLM-L6-v2"
    p2 = str(tmp_path / "pruned.leann")
    enc = BertEncoder.load(model, allow_random=True).to("cuda", dtype=torch.float16)
    tok = load_tokenizer(model, 256, p2, texts, enc.cfg.vocab_size, allow_stand_in=enc.weights_source == "random")
    seqs = tok.encode_batch(texts)
    T = max(len(q) for q in seqs)
    ids = torch.zeros((len(seqs), T), dtype=torch.int32)
    for i, q in enumerate(seqs):
        ids[i, : len(q)] = torch.tensor(q, dtype=torch.int32)
    emb = enc.encode_tokens(ids.cuda(), torch.tensor([len(q) for q in seqs], dtype=torch.int32).cuda()).cpu().numpy()
    write_leann_bundle(p2, texts, emb, model, distance_metric="mips", M=8, efConstruction=40)
    plain = BACKEND_REGISTRY["mi355x"].searcher(p2, allow_random_weights=True)
    cached = BACKEND_REGISTRY["mi355x"].searcher(p2, hub_cache_ratio=0.2, allow_random_weights=True)
    ra = plain.search(emb[:6], 4, complexity=32, recompute_embeddings=True, zmq_port=5557)
    rb = cached.search(emb[:6], 4, complexity=32, recompute_embeddings=True, zmq_port=5557)
    assert ra["labels"] == rb["labels"] and np.allclose(ra["distances"], rb["distances"], atol=2e-3)
    assert cached.last_stats()["nunique"] < plain.last_stats()["nunique"]
    plain.cleanup()
    cached.cleanup()
