"""CPU checks of the fused feed-forward kernel's design (csrc/lm_mlp_fused.hip) without a GPU: the lane-level
index algebra through tests/mfma_emulation.py, the host-side W2 packing, and the GELU approximation."""
import numpy as np
import pytest

from tests import mfma_emulation as me


def _gelu_exact(v):
    from scipy.special import erf

    v = v.astype(np.float64)
    return (0.5 * v * (1.0 + erf(v / np.sqrt(2.0)))).astype(np.float32)


def test_mfma_emulation_is_a_matrix_product():
    rng = np.random.default_rng(0)
    amat = rng.standard_normal((32, 16)).astype(np.float16)
    bmat = rng.standard_normal((16, 32)).astype(np.float16)
    lane = np.arange(64)
    a = np.stack([amat[lane % 32, 8 * (lane // 32) + e] for e in range(8)], axis=1)
    b = np.stack([bmat[8 * (lane // 32) + e, lane % 32] for e in range(8)], axis=1)
    d = me.mfma_32x32x16(a, b, np.zeros((64, 16), np.float32))
    ref = amat.astype(np.float64) @ bmat.astype(np.float64)
    for r in range(16):
        m = (r & 3) + 8 * (r >> 2) + 4 * (lane // 32)
        assert np.allclose(d[:, r], ref[m, lane % 32], atol=1e-5)


def test_k_permutation_is_a_bijection_matching_the_accumulator_layout():
    torch = pytest.importorskip("torch")
    from leann_amd.encoder import fused_mlp_k_permutation

    perm = fused_mlp_k_permutation().numpy()
    assert sorted(perm.tolist()) == list(range(32))
    # B fragment of k-step u, lane group g, element jj = accumulator register 8u + jj = hidden unit
    # (r & 3) + 8 (r >> 2) + 4 g; the A fragment reads packed positions 16u + 8g + jj
    for u in range(2):
        for g in range(2):
            for jj in range(8):
                r = 8 * u + jj
                assert perm[16 * u + 8 * g + jj] == (r & 3) + 8 * (r >> 2) + 4 * g
    assert torch.is_tensor(fused_mlp_k_permutation())


@pytest.mark.parametrize("ffn", [32, 64, 160])
def test_fused_mlp_lane_level_data_flow_matches_a_plain_mlp(ffn):
    torch = pytest.importorskip("torch")
    from leann_amd.encoder import pack_w2_fused_mlp

    rng = np.random.default_rng(ffn)
    h = me.ML_H
    x = (rng.standard_normal((32, h))).astype(np.float16)
    w1 = (rng.standard_normal((ffn, h)) / np.sqrt(h)).astype(np.float16)
    w2 = (rng.standard_normal((h, ffn)) / np.sqrt(ffn)).astype(np.float16)
    b1 = rng.standard_normal(ffn).astype(np.float32) * 0.1
    b2 = rng.standard_normal(h).astype(np.float32) * 0.1
    gamma = (1 + 0.1 * rng.standard_normal(h)).astype(np.float16)
    beta = (0.1 * rng.standard_normal(h)).astype(np.float16)
    w2p = pack_w2_fused_mlp(torch.from_numpy(w2)).numpy()
    assert w2p.shape == (ffn // 32, h, 32)
    y = me.emulate_mlp_wave(x, w1, b1, w2p, b2, gamma, beta, 1e-12, _gelu_exact)
    # the cross-slab pipelined variant moves the same bytes through the two LDS stages on a different timetable
    yp = me.emulate_mlp_wave_pipelined(x, w1, b1, w2p, b2, gamma, beta, 1e-12, _gelu_exact)
    assert np.array_equal(y.view(np.uint16), yp.view(np.uint16))
    # plain MLP with the same rounding points (fp32 accumulation, fp16 GELU output)
    hid = x.astype(np.float64) @ w1.astype(np.float64).T + b1
    p = _gelu_exact(hid.astype(np.float32)).astype(np.float16)
    z = p.astype(np.float64) @ w2.astype(np.float64).T + b2 + x.astype(np.float64)
    mu = z.mean(1, keepdims=True)
    var = ((z - mu) ** 2).mean(1, keepdims=True)
    ref = (z - mu) / np.sqrt(var + 1e-12) * gamma.astype(np.float64) + beta.astype(np.float64)
    err = np.abs(y.astype(np.float64) - ref).max()
    assert err < 6e-3, err  # fp16 output rounding (|y| up to ~4) + rare 1-ulp flips of the fp16 GELU outputs


def test_gelu_approximation_is_within_one_fp16_ulp_of_erf_gelu():
    """Abramowitz-Stegun 7.1.26 form used by gelu2 (cancellation-free: max(x,0) - 0.5|x| t P(t) exp(-x^2/2)), fp32."""
    f = np.float32
    x = np.linspace(-12, 12, 400001).astype(f)
    ax = np.abs(x)
    d = (ax * f(0.3275911 * 0.70710678) + f(1)).astype(f)
    t = (f(1) / d).astype(f)
    p = (t * f(1.061405429) + f(-1.453152027)).astype(f)
    for c in (1.421413741, -0.284496736, 0.254829592):
        p = (p * t + f(c)).astype(f)
    ph = (p * ((ax * t).astype(f) * f(0.5))).astype(f)
    w = ((x * x).astype(f) * f(-0.72134752)).astype(f)
    ex = np.exp2(w.astype(np.float64)).astype(f)
    got = (np.maximum(x, f(0)) - ph * ex).astype(f)
    ref = _gelu_exact(x).astype(np.float64)
    assert (np.abs(got - ref) <= 4e-7 + 1.2e-7 * np.abs(ref)).all()  # 3.4e-7 absolute, plus fp32 rounding of large results
    ulps = np.abs(got.astype(np.float16).view(np.int16).astype(np.int32) - ref.astype(np.float16).view(np.int16).astype(np.int32))
    assert ulps.max() <= 1


@pytest.mark.parametrize("passes", [1, 3])
def test_linear_h384_lane_level_data_flow_matches_a_plain_linear(passes):
    torch = pytest.importorskip("torch")
    from leann_amd.encoder import pack_w_linear_h384

    rng = np.random.default_rng(passes)
    h, n = me.ML_H, me.ML_H * passes
    x = rng.standard_normal((32, h)).astype(np.float16)
    w = (rng.standard_normal((n, h)) / np.sqrt(h)).astype(np.float16)
    b = (0.1 * rng.standard_normal(n)).astype(np.float32)
    wp = pack_w_linear_h384(torch.from_numpy(w)).numpy()
    assert wp.shape == (passes, 12, h, 32)
    got = me.emulate_linear_wave(x, wp, b)
    ref = x.astype(np.float64) @ w.astype(np.float64).T + b
    assert np.abs(got.astype(np.float64) - ref).max() < 4e-3


def test_accumulator_order_packing_of_w1_and_natural_slabs_of_wo():
    """Host-side packing for the fused layer tail (csrc/lm_mlp_fused_v3.hip: k_attn_out_mlp_h384): W1's columns go into the order in
    which a lane's accumulator registers hold the LayerNorm output (fragment 2j+u, lane group g, element e <- feature
    32j + 16u + 8(e>>2) + 4g + (e&3)), W_o into natural-order 32-wide k slabs; both are pure permutations, so
    x_acc_order @ W1_acc_order^T == x @ W1^T exactly."""
    torch = pytest.importorskip("torch")
    from leann_amd.encoder import fused_mlp_k_permutation, pack_w1_acc_order, pack_wo_slabs

    rng = np.random.default_rng(5)
    h, f = me.ML_H, 96
    w1 = torch.from_numpy(rng.standard_normal((f, h)).astype(np.float32))
    w1a = pack_w1_acc_order(w1)
    assert w1a.shape == w1.shape
    # the kernel's B fragment of k-step ks = 2j+u holds, in lane group g, element e, feature phi(ks, g, e); the A fragment of W1
    # reads packed column 16 ks + 8 g + e -- so that column must be feature phi
    for ks in range(h // 16):
        j, u = divmod(ks, 2)
        for g in range(2):
            for e in range(8):
                phi = 32 * j + 16 * u + 8 * (e >> 2) + 4 * g + (e & 3)
                assert torch.equal(w1a[:, 16 * ks + 8 * g + e], w1[:, phi])
    perm = fused_mlp_k_permutation().numpy()
    cols = np.concatenate([32 * j + perm for j in range(h // 32)])
    assert sorted(cols.tolist()) == list(range(h))
    x = torch.from_numpy(rng.standard_normal((7, h)).astype(np.float32))
    assert torch.allclose(x[:, cols] @ w1a.T, x @ w1.T, atol=1e-5)
    wo = torch.from_numpy(rng.standard_normal((h, h)).astype(np.float32))
    wos = pack_wo_slabs(wo)
    assert wos.shape == (h // 32, h, 32)
    for s in (0, 5, 11):
        assert torch.equal(wos[s], wo[:, 32 * s : 32 * s + 32])
