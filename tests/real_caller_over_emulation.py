"""SURVEY 8 rows a12 (+ a1, a9) on the CPU box: the REAL caller -- leann.api.LeannSearcher(index).search("...") from the reference's own
leann-core (api.py:623-642 factory, :644-796 search) -- driving our backend plugin down through the C ABI into the product's
kernel sources, compiled for the host by tests/hip_emul (libleann_mi355x_emul.so).  Run as a script by
tests/test_emulated_search.py:

    python -m tests.real_caller_over_emulation <libleann_mi355x_emul.so> <leann-core/src> <tmp dir>

What is the reference's: LeannSearcher, PassageManager, BACKEND_REGISTRY / autodiscovery, SearchResult.  What is ours: the
registered "mi355x" backend, token store, recompute provider, the fp16 packed encoder kernels and the search kernels (all inside the
emulated library).  Only THIS file pretends host tensors are device tensors (the product has no such switch): Tensor.is_cuda,
torch.cuda.current_stream and the backend's torch device are patched here."""
import json
import sys
from pathlib import Path
from unittest import mock

import numpy as np


def main(lib_path: str, leann_src: str, tmp: str) -> None:
    root = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root))
    sys.path.insert(0, leann_src)
    import torch

    from tests.emulated_search_cases import _load

    _load(lib_path)  # points leann_amd._lib at the emulated library before anything else resolves it
    from leann.api import LeannSearcher, SearchResult  # the reference's caller
    from leann.registry import BACKEND_REGISTRY

    import leann_backend_mi355x  # noqa: F401 - registers "mi355x" / "mi355x_diskann" with the reference's registry
    from leann_amd import _lib
    from leann_amd.backend import Mi355xSearcher, write_leann_bundle
    from leann_amd.encoder import BertEncoder
    from leann_amd.tokenizer import load_tokenizer

    assert "mi355x" in BACKEND_REGISTRY and _lib.device_count() >= 1
    torch.set_num_threads(1)
    # A small REAL checkpoint directory (Hugging Face BertModel + sentence-transformers files + WordPiece vocab.txt): hidden 384 =
    # 12 heads x 32 so that the hand-written kernels apply, one layer and ffn 128 so that the thread-per-lane emulation of a whole
    # recompute search stays in the tens of seconds.  The backend loads it the way it loads all-MiniLM-L6-v2 from disk.
    import transformers

    torch.manual_seed(11)
    words = ["cat", "dog", "crocodile", "banana", "robot", "violin", "glacier", "senator", "number", "the", "sat", "on", "mat"]
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + words + [str(i) for i in range(10)] + [f"##{i}" for i in range(10)]
    hc = transformers.BertConfig(vocab_size=len(vocab), hidden_size=384, num_hidden_layers=1, num_attention_heads=12, intermediate_size=128,
                                 max_position_embeddings=64)
    hf = transformers.BertModel(hc, add_pooling_layer=False)
    with torch.no_grad():
        for prm in hf.parameters():  # livelier than the 0.02 default init: distinct texts get clearly distinct embeddings
            if prm.dim() == 2:
                prm.mul_(4.0)
    ck = Path(tmp) / "tiny-minilm"
    hf.save_pretrained(ck)
    (ck / "modules.json").write_text(json.dumps([
        {"idx": 0, "name": "0", "path": "", "type": "sentence_transformers.models.Transformer"},
        {"idx": 1, "name": "1", "path": "1_Pooling", "type": "sentence_transformers.models.Pooling"},
        {"idx": 2, "name": "2", "path": "2_Normalize", "type": "sentence_transformers.models.Normalize"}]))
    (ck / "1_Pooling").mkdir()
    (ck / "1_Pooling" / "config.json").write_text(json.dumps({"word_embedding_dimension": 384, "pooling_mode_cls_token": False, "pooling_mode_mean_tokens": True}))
    (ck / "sentence_bert_config.json").write_text(json.dumps({"max_seq_length": 48, "do_lower_case": True}))
    (ck / "vocab.txt").write_text("\n".join(vocab) + "\n")
    model = str(ck)
    texts = [" ".join([words[i % 8]] * (1 + i % 5) + ["number", str(i), words[(i * 3 + 1) % 8]] * (1 + i % 3)) for i in range(40)]
    p = str(Path(tmp) / "real.leann")
    enc32 = BertEncoder.load(model).eval()
    assert enc32.weights_source == "checkpoint" and (enc32.cfg.hidden, enc32.cfg.layers, enc32.cfg.pooling, enc32.cfg.normalize) == (384, 1, "mean", True)
    tok = load_tokenizer(model, 48, p, None, enc32.cfg.vocab_size)
    assert tok.kind == "hf-local-vocab"
    seqs = tok.encode_batch(texts)
    T = max(len(t) for t in seqs)
    ids = torch.zeros((len(seqs), T), dtype=torch.int32)
    for i, t in enumerate(seqs):
        ids[i, : len(t)] = torch.tensor(t, dtype=torch.int32)
    with torch.no_grad():
        emb = enc32(ids, torch.tensor([len(t) for t in seqs], dtype=torch.int32)).float().numpy()
    write_leann_bundle(p, texts, emb, model, distance_metric="mips", M=6, efConstruction=40)

    class _Stream:
        cuda_stream = 0

    used = []
    real_check = _lib.check

    def recording_check(rc, what=""):
        used.append(what)
        return real_check(rc, what)

    cpu = torch.device("cpu")
    with mock.patch.object(torch.Tensor, "is_cuda", new=property(lambda self: True)), \
            mock.patch("torch.cuda.current_stream", new=lambda *a, **k: _Stream()), \
            mock.patch.object(Mi355xSearcher, "_torch_device", new=lambda self: cpu), \
            mock.patch.object(_lib, "check", new=recording_check):
        searcher = LeannSearcher(p)  # real weights: no random-weight permission needed
        assert isinstance(searcher.backend_impl, Mi355xSearcher)  # chosen by the reference's factory from meta.json
        q = 17
        res = searcher.search(texts[q], top_k=3, complexity=16, recompute_embeddings=True)
        st = searcher.backend_impl.last_stats()
        # the same call straight into the backend (what LeannSearcher does internally, api.py:703-755)
        qe = searcher.backend_impl.compute_query_embedding(texts[q])
        direct = searcher.backend_impl.search(qe, 3, complexity=16, recompute_embeddings=True, zmq_port=5557)
        searcher.cleanup()
    assert len(res) == 3 and all(isinstance(r, SearchResult) for r in res)
    assert [r.id for r in res] == direct["labels"][0]
    assert np.allclose([r.score for r in res], direct["distances"][0], atol=0, rtol=0)
    assert all(r.text == texts[int(r.id)] for r in res)           # PassageManager lookups of OUR labels
    assert res[0].score >= res[1].score >= res[2].score           # +IP, best first
    assert res[0].id == str(q), (res[0].id, [r.score for r in res])
    # exact ranking of the fp32 build-time embeddings: the fp16 kernels must land in its top 3 with matching scores
    exact = emb @ emb[q]
    assert abs(res[0].score - float(exact[q])) < 5e-3 * max(1.0, abs(float(exact[q])))
    assert st["nrounds"] > 3 and st["nunique"] > 10, st           # a real recompute traversal, not a one-hop lookup
    # the whole stack went through the C ABI: graph reader, token store, the built-in recompute provider (csrc/lm_recompute.hip: the search
    # rounds' token packing and fp16 forwards -- QKV GEMM, attention, fused layer tail / general kernels -- run INSIDE lm_index_search,
    # no Python per round), the search itself
    for name in ("lm_index_read", "lm_tokens_create", "lm_recompute_create", "lm_index_set_recompute", "lm_index_search"):
        assert any(u.startswith(name) for u in used), (name, sorted(set(used)))
    print(json.dumps({"ids": [r.id for r in res], "scores": [round(float(r.score), 5) for r in res], "stats": {k: st[k] for k in ("nrounds", "nunique", "ndis")},
                      "abi_calls": len(used)}))

    # ---- row a9 through the same real caller: a DiskANN-style bundle (meta.json backend_name "mi355x_diskann", recompute mode: pruned
    #      index + product quantiser) -> LeannSearcher -> Mi355xDiskannSearcher -> lm_pq_batch_search: PQ traversal kernel + ONE deferred
    #      exact rerank through the built-in provider (diskann_backend.py:444-449, 453-467) ----
    from leann_amd.backend import Mi355xDiskannSearcher

    p2 = str(Path(tmp) / "real_diskann.leann")
    write_leann_bundle(p2, texts, emb, model, backend_name="mi355x_diskann", distance_metric="mips", graph_degree=8, complexity=24, pq_bytes=48,
                       is_recompute=True)
    used.clear()
    with mock.patch.object(torch.Tensor, "is_cuda", new=property(lambda self: True)), \
            mock.patch("torch.cuda.current_stream", new=lambda *a, **k: _Stream()), \
            mock.patch.object(Mi355xSearcher, "_torch_device", new=lambda self: cpu), \
            mock.patch.object(_lib, "check", new=recording_check):
        s2 = LeannSearcher(p2)
        assert isinstance(s2.backend_impl, Mi355xDiskannSearcher)
        q2 = 23
        res2 = s2.search(texts[q2], top_k=3, complexity=24, beam_width=4, recompute_embeddings=True)
        st2 = s2.backend_impl.last_stats()
        s2.cleanup()
    assert len(res2) == 3 and res2[0].id == str(q2) and all(r.text == texts[int(r.id)] for r in res2)
    assert res2[0].score >= res2[1].score >= res2[2].score
    exact2 = emb @ emb[q2]
    assert abs(res2[0].score - float(exact2[q2])) < 5e-3 * max(1.0, abs(float(exact2[q2])))
    assert 0 < st2["nunique"] <= 24 and st2["ndis"] >= st2["nunique"] and st2["update_launches"] == 1, st2  # PQ evaluations along the path (ndis), ONE exact rerank of the <= complexity final candidates
    for name in ("lm_index_read", "lm_pq_attach", "lm_recompute_create", "lm_index_set_recompute", "lm_pq_batch_search"):
        assert any(u.startswith(name) for u in used), (name, sorted(set(used)))
    print(json.dumps({"diskann_style": {"ids": [r.id for r in res2], "scores": [round(float(r.score), 5) for r in res2],
                                        "stats": {k: st2[k] for k in ("ndis", "nunique", "nrounds")}, "abi_calls": len(used)}}))
    print("REAL CALLER OK")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
