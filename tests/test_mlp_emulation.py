"""CPU checks of the fused layer tail's operand layouts (csrc/lm_layer_tail_h384.hip) without a GPU: the MFMA lane layout model of
tests/mfma_emulation.py and the host-side W1 / W2 / W_o packings built on it."""
import numpy as np
import pytest

from tests import mfma_emulation as me


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


def test_accumulator_order_packing_of_w1_and_natural_slabs_of_wo():
    """Host-side packing for the fused layer tail (csrc/lm_layer_tail_h384.hip: k_layer_tail_h384): W1's columns go into the order in
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
