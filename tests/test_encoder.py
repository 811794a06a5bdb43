"""Encoder (leann_amd/encoder.py) on CPU: same arithmetic as the Hugging Face BertModel that sentence-transformers
wraps (compute_embeddings_sentence_transformers, leann/embedding_compute.py:229-239; manual mean-pool path :323-334)."""
import numpy as np
import pytest
import torch

from leann_amd.encoder import PRESETS, BertEncoder, EncoderConfig, config_for, hf_reference_embed


def _hf(hidden=64, layers=2, heads=4, ffn=128, seed=0):
    from transformers import BertConfig, BertModel

    torch.manual_seed(seed)
    hc = BertConfig(vocab_size=1000, hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads,
                    intermediate_size=ffn, max_position_embeddings=64)
    m = BertModel(hc, add_pooling_layer=False).eval()
    with torch.no_grad():  # make biases / LayerNorm parameters non-trivial
        for p in m.parameters():
            p.add_(0.05 * torch.randn_like(p))
    return m


@pytest.mark.parametrize("pooling,normalize", [("mean", True), ("cls", True), ("mean", False)])
def test_matches_huggingface_bert(pooling, normalize):
    hf = _hf()
    cfg = EncoderConfig(vocab_size=1000, hidden=64, layers=2, heads=4, ffn=128, max_pos=64, pooling=pooling, normalize=normalize)
    enc = BertEncoder.from_hf_state_dict(cfg, hf.state_dict())
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(1, 1000, (9, 40), generator=g)
    lens = torch.tensor([40, 3, 17, 40, 1, 25, 33, 2, 40])
    for i, l in enumerate(lens):
        ids[i, l:] = 0
    ref = hf_reference_embed(hf, ids, lens, pooling, normalize)
    with torch.no_grad():
        got = enc(ids.int(), lens)
    assert (got - ref).abs().max() < 1e-5
    # length-bucketed batching == one padded batch
    assert (enc.encode_tokens(ids.int(), lens, batch_size=3, bucket=8) - ref).abs().max() < 1e-5
    assert enc.encode_tokens(ids[:0].int(), lens[:0]).shape == (0, 64)


def test_random_init_is_deterministic_and_architecture_presets():
    a = BertEncoder.random_init(PRESETS["all-minilm-l6-v2"], seed=0)
    b = BertEncoder.random_init(PRESETS["all-minilm-l6-v2"], seed=0)
    c = BertEncoder.random_init(PRESETS["all-minilm-l6-v2"], seed=1)
    sa, sb, sc = a.state_dict(), b.state_dict(), c.state_dict()
    assert all(torch.equal(sa[k], sb[k]) for k in sa) and not torch.equal(sa["word.weight"], sc["word.weight"])
    # parameter counts of SURVEY Appendix D: 22.6 M total / 10.65 M encoder (MiniLM-L6)
    total = sum(p.numel() for p in a.parameters())
    enc_only = sum(p.numel() for p in a.layers.parameters())
    assert abs(total - 22.7e6) < 0.2e6 and abs(enc_only - 10.65e6) < 0.05e6
    assert config_for("sentence-transformers/all-MiniLM-L6-v2").hidden == 384
    assert config_for("BAAI/bge-base-en-v1.5").pooling == "cls" and config_for("BAAI/bge-base-en-v1.5").hidden == 768
    assert config_for("facebook/contriever").normalize is False
    assert config_for("unknown-model").layers == 6  # default preset


def test_flops_formula_matches_survey():
    """2*P_enc*T + 4*L*T^2*H: 6.06 / 12.11 / 45.96 GFLOP per 256-token chunk."""
    for name, gf in (("all-minilm-l6-v2", 6.06), ("bge-small-en-v1.5", 12.11), ("bge-base-en-v1.5", 45.96)):
        assert abs(PRESETS[name].flops_per_chunk(256) / 1e9 - gf) < 0.03 * gf


def test_load_falls_back_to_seeded_weights_offline():
    enc = BertEncoder.load("sentence-transformers/all-MiniLM-L6-v2", allow_random=True)
    ref = BertEncoder.random_init(PRESETS["all-minilm-l6-v2"], seed=0)
    if not all(torch.equal(v, ref.state_dict()[k]) for k, v in enc.state_dict().items()):
        pytest.skip("a real checkpoint is available locally")
    ids = torch.randint(1000, 30000, (3, 20), dtype=torch.int32)
    with torch.no_grad():
        e = enc(ids, torch.tensor([20, 5, 11]))
    assert e.shape == (3, 384) and np.allclose(e.norm(dim=1).numpy(), 1.0, atol=1e-5)


def test_load_without_checkpoint_raises_unless_random_is_allowed():
    """no silent random weights in the production path."""
    with pytest.raises(RuntimeError, match="not available locally"):
        BertEncoder.load("sentence-transformers/definitely-not-a-cached-model")
    enc = BertEncoder.load("sentence-transformers/definitely-not-a-cached-model", allow_random=True)
    assert enc.weights_source == "random"
    with pytest.raises(ValueError, match="no architecture preset"):
        config_for("some/unknown-model", strict=True)


def _save_tiny_checkpoint(tmp_path, model_type="bert", pooling="cls", normalize=False, max_seq_length=77):
    import json

    import transformers

    if model_type == "bert":
        hc = transformers.BertConfig(vocab_size=120, hidden_size=32, num_hidden_layers=1, num_attention_heads=2, intermediate_size=64,
                                     max_position_embeddings=96)
        hf = transformers.BertModel(hc, add_pooling_layer=False)
    else:
        hc = transformers.MPNetConfig(vocab_size=120, hidden_size=32, num_hidden_layers=1, num_attention_heads=2, intermediate_size=64,
                                      max_position_embeddings=96)
        hf = transformers.MPNetModel(hc, add_pooling_layer=False)
    d = tmp_path / f"ckpt_{model_type}_{pooling}"
    hf.save_pretrained(d)
    mods = [{"idx": 0, "name": "0", "path": "", "type": "sentence_transformers.models.Transformer"},
            {"idx": 1, "name": "1", "path": "1_Pooling", "type": "sentence_transformers.models.Pooling"}]
    if normalize:
        mods.append({"idx": 2, "name": "2", "path": "2_Normalize", "type": "sentence_transformers.models.Normalize"})
    (d / "modules.json").write_text(json.dumps(mods))
    (d / "1_Pooling").mkdir()
    (d / "1_Pooling" / "config.json").write_text(json.dumps({"word_embedding_dimension": 32, "pooling_mode_cls_token": pooling == "cls",
                                                               "pooling_mode_mean_tokens": pooling == "mean"}))
    (d / "sentence_bert_config.json").write_text(json.dumps({"max_seq_length": max_seq_length, "do_lower_case": True}))
    (d / "vocab.txt").write_text("\n".join(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + [f"w{i}" for i in range(115)]) + "\n")
    return d, hf


def test_load_reads_pooling_and_length_from_the_checkpoints_sentence_transformers_files(tmp_path):
    """pooling / normalize / max_seq_length come from the checkpoint, not from a name match --
    this directory name matches no preset at all."""
    d, hf = _save_tiny_checkpoint(tmp_path, pooling="cls", normalize=False, max_seq_length=77)
    enc = BertEncoder.load(str(d))
    assert enc.weights_source == "checkpoint"
    assert (enc.cfg.pooling, enc.cfg.normalize, enc.cfg.max_seq_length, enc.cfg.hidden, enc.cfg.vocab_size) == ("cls", False, 77, 32, 120)
    ids = torch.randint(5, 120, (4, 12), dtype=torch.int32)
    lens = torch.tensor([12, 3, 7, 1])
    with torch.no_grad():
        got = enc(ids, lens)
    ref = hf_reference_embed(hf.eval(), ids, lens, "cls", False)
    assert (got - ref).abs().max() < 1e-5
    d2, _ = _save_tiny_checkpoint(tmp_path, pooling="mean", normalize=True)
    e2 = BertEncoder.load(str(d2))
    assert (e2.cfg.pooling, e2.cfg.normalize) == ("mean", True)


def test_load_rejects_non_bert_architectures(tmp_path):
    """all-mpnet-base-v2 (the reference server's default --model-name) is an MPNetModel: different state dict."""
    d, _ = _save_tiny_checkpoint(tmp_path, model_type="mpnet", pooling="mean")
    with pytest.raises(RuntimeError, match="model_type='mpnet'"):
        BertEncoder.load(str(d))
    with pytest.raises(RuntimeError, match="model_type='mpnet'"):
        BertEncoder.load(str(d), allow_random=True)  # a present-but-unsupported checkpoint never degrades to random weights


def test_tokenizer_comes_from_the_checkpoint_vocabulary_or_raises(tmp_path):
    """vocab.txt-only checkpoints get a real WordPiece pipeline; a stand-in vocabulary is only
    admissible with random weights; a vocabulary larger than the embedding table is rejected."""
    from leann_amd.tokenizer import load_tokenizer

    d, _ = _save_tiny_checkpoint(tmp_path)
    t = load_tokenizer(str(d), 16, None, None, vocab_size=120)
    assert t.kind == "hf-local-vocab" and t.vocab_size == 120
    assert t.encode_batch(["w3 w4 zzz"])[0] == [2, 8, 9, 1, 3]  # [CLS] w3 w4 [UNK] [SEP]
    with pytest.raises(ValueError, match="embedding table"):
        load_tokenizer(str(d), 16, None, None, vocab_size=100)
    with pytest.raises(FileNotFoundError, match="stand-in"):
        load_tokenizer("sentence-transformers/definitely-not-a-cached-model", 16, str(tmp_path / "i.leann"), ["a b c"], 30522)
    s = load_tokenizer("sentence-transformers/definitely-not-a-cached-model", 16, str(tmp_path / "i.leann"), ["a b c", "c d"], 30522,
                       allow_stand_in=True)
    assert s.kind == "stand-in-trained"


def test_token_store_rejects_ids_beyond_u16():
    """multilingual vocabularies (119k) must not wrap silently."""
    from leann_amd.token_store import TokenStore

    with pytest.raises(ValueError, match="u16 token store"):
        TokenStore.from_lists([[101, 70000, 102]])
