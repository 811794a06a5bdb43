"""Opt-in kernels that have passed the thread-per-lane emulation (tests/emulated_search_cases.py) but have NOT run on an MI355X yet.
Nothing here is on the default path or in bench.py; `pytest -m gpu` does not select these tests, `pytest -m gpu_next` does -- run it
first thing in the next GPU session, then `leann_amd/lib/bin/kbench 262107 20 tailqkv` for the timing, and make the kernel the default
only if it wins.

Today: the fused layer tail with the NEXT layer's QKV projection behind it (LEANN_MI355X_QKV_IN_TAIL=1,
csrc/lm_mlp_fused_v3.hip: k_attn_out_mlp_qkv_h384)."""
import pytest


def _has_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        return False


pytestmark = [pytest.mark.gpu_next, pytest.mark.skipif(not _has_gpu(), reason="needs an MI355X")]


@pytest.mark.parametrize("tokens", [1, 128, 129, 5000])
def test_layer_tail_with_next_qkv_projection(tokens, monkeypatch):
    import torch

    from leann_amd.encoder import EncoderConfig, _Layer, fused_attn_out_mlp, fused_linear_h384

    torch.manual_seed(tokens)
    cfg = EncoderConfig(hidden=384, layers=2, heads=12, ffn=1536)
    layer, nxt = _Layer(cfg).to("cuda", dtype=torch.float16), _Layer(cfg).to("cuda", dtype=torch.float16)
    with torch.no_grad():
        for ln in (layer.ln1, layer.ln2):
            ln.weight.copy_(1 + 0.1 * torch.randn(384))
            ln.bias.copy_(0.1 * torch.randn(384))
        nxt.qkv.bias.copy_(0.2 * torch.randn(1152))
    a = torch.randn((tokens, 384), device="cuda").half()
    res = torch.randn((tokens, 384), device="cuda").half()
    with torch.no_grad():
        y0 = fused_attn_out_mlp(a, res, layer)
        monkeypatch.setenv("LEANN_MI355X_QKV_IN_TAIL", "1")
        r = fused_attn_out_mlp(a, res, layer, nxt)
        assert isinstance(r, tuple)
        y1, qkv = r
        ref = fused_linear_h384(y1, nxt.qkv)  # the weight-stationary GEMM on the same fp16 y
        ref32 = y1.float() @ nxt.qkv.weight.float().t() + nxt.qkv.bias.float()
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)  # the layer output is the same code
    assert not torch.isnan(qkv).any()
    assert (qkv.float() - ref32).abs().max().item() <= 6e-3 * max(1.0, float(ref32.abs().max()))
    assert (qkv.float() - ref.float()).abs().max().item() <= 6e-3 * max(1.0, float(ref32.abs().max()))


def test_encoder_forward_with_qkv_in_the_tail(monkeypatch):
    import torch

    from leann_amd.encoder import BertEncoder, config_for
    from leann_amd.synth import CorpusSpec, SyntheticCorpus, pad_batch

    enc = BertEncoder.random_init(config_for("all-MiniLM-L6-v2"), 0).to("cuda", dtype=torch.float16)
    ids, lens = pad_batch(*SyntheticCorpus(CorpusSpec(n_chunks=300, n_topics=4)).chunks(), 256)
    ti, tl = torch.from_numpy(ids).cuda(), torch.from_numpy(lens).cuda()
    d = enc.encode_tokens_packed(ti, tl)
    monkeypatch.setenv("LEANN_MI355X_QKV_IN_TAIL", "1")
    q = enc.encode_tokens_packed(ti, tl)
    assert (d - q).abs().max() < 2e-3 and not torch.isnan(q).any()
