"""Kernels that are written and cross-compiled but NOT yet validated on an MI355X: they are opt-in in the product
(environment switches) and their tests carry the ``gpu_next`` marker, which neither ``-m gpu`` nor a CPU run
executes (no GPU -> skipped).  First thing to do with GPU time:

    python -m pytest tests/test_gpu_next.py -m gpu_next -x -q && python scripts/attn_bench.py

Once a kernel passes and wins, flip its default and move the test into test_gpu_pipeline.py."""
import pytest


def _has_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        return False


pytestmark = [pytest.mark.gpu_next, pytest.mark.skipif(not _has_gpu(), reason="needs an MI355X")]


def _attention_case(torch, heads, maxlen, nseq=37):
    g = torch.Generator(device="cpu").manual_seed(heads * 1000 + maxlen)
    lens = torch.randint(1, maxlen + 1, (nseq,), generator=g)
    lens[0], lens[-1] = maxlen, 1
    if nseq > 4:
        lens[1] = max(1, maxlen - 1)          # odd/even tails of the key-pair staging
        lens[2] = max(1, (maxlen // 32) * 32)  # exactly full tiles: no masked tile at all
    cu = torch.zeros(nseq + 1, dtype=torch.int32)
    cu[1:] = torch.cumsum(lens, 0)
    tot, H = int(cu[-1]), heads * 32
    qkv = (torch.randn((tot, 3 * H), generator=g) * 1.5).half().cuda()
    q3 = qkv.float().view(tot, 3, heads, 32)
    ref = torch.empty((tot, H), device="cuda")
    for i in range(nseq):
        a, b = int(cu[i]), int(cu[i + 1])
        q, k, v = (q3[a:b, j].transpose(0, 1) for j in range(3))  # [heads, L, 32]
        p = torch.softmax(q @ k.transpose(1, 2) / 32**0.5, dim=-1)
        ref[a:b] = (p @ v).transpose(0, 1).reshape(b - a, H)
    return qkv, cu.cuda(), int(lens.max()), ref


@pytest.mark.parametrize("heads,maxlen", [(12, 256), (12, 255), (12, 200), (12, 70), (12, 64), (4, 33), (4, 32), (2, 2), (2, 1)])
def test_attention_revision2_matches_fp32_reference_and_revision1(heads, maxlen, monkeypatch):
    """lm_attn_v2.hip (LEANN_MI355X_ATTN=2) vs a plain PyTorch fp32 reference of the same op, and vs revision 1."""
    import torch

    from leann_amd.encoder import fused_attention_hd32

    qkv, cu, mx, ref = _attention_case(torch, heads, maxlen)
    monkeypatch.setenv("LEANN_MI355X_ATTN", "1")
    o1 = fused_attention_hd32(qkv, cu, heads, mx)
    monkeypatch.setenv("LEANN_MI355X_ATTN", "2")
    o2 = fused_attention_hd32(qkv, cu, heads, mx)
    torch.cuda.synchronize()
    assert o2 is not None and o2.shape == ref.shape
    assert not torch.isnan(o2).any()
    err = (o2.float() - ref).abs().max().item()
    assert err < 4e-3, err
    assert (o2.float() - o1.float()).abs().max().item() < 2e-3


def test_encoder_forward_with_attention_revision2(monkeypatch):
    import torch

    from leann_amd.encoder import BertEncoder, config_for
    from leann_amd.synth import CorpusSpec, SyntheticCorpus, pad_batch

    enc = BertEncoder.random_init(config_for("all-MiniLM-L6-v2"), 0).to("cuda", dtype=torch.float16)
    ids, lens = pad_batch(*SyntheticCorpus(CorpusSpec(n_chunks=300, n_topics=4)).chunks(), 256)
    ti, tl = torch.from_numpy(ids).cuda(), torch.from_numpy(lens).cuda()
    monkeypatch.setenv("LEANN_MI355X_ATTN", "0")
    b = enc.encode_tokens_packed(ti, tl)
    monkeypatch.setenv("LEANN_MI355X_ATTN", "2")
    a = enc.encode_tokens_packed(ti, tl)
    assert (a - b).abs().max() < 2e-3
