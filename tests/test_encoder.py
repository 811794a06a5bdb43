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
    enc = BertEncoder.load("sentence-transformers/all-MiniLM-L6-v2")
    ref = BertEncoder.random_init(PRESETS["all-minilm-l6-v2"], seed=0)
    if not all(torch.equal(v, ref.state_dict()[k]) for k, v in enc.state_dict().items()):
        pytest.skip("a real checkpoint is available locally")
    ids = torch.randint(1000, 30000, (3, 20), dtype=torch.int32)
    with torch.no_grad():
        e = enc(ids, torch.tensor([20, 5, 11]))
    assert e.shape == (3, 384) and np.allclose(e.norm(dim=1).numpy(), 1.0, atol=1e-5)
