"""Lane-level CPU emulation of the MFMA data flow used by the hand-written encoder kernels.

``mfma_32x32x16`` implements the operand / result layout of ``v_mfma_f32_32x32x16_f16`` as the kernels use it
(the conventions every kernel of csrc/ relies on, validated on hardware):

    A operand  lane l, element e  ->  A[m = l % 32][k = 8 * (l // 32) + e]
    B operand  lane l, element e  ->  B[k = 8 * (l // 32) + e][n = l % 32]
    D result   lane l, register r ->  D[m = (r & 3) + 8 * (r >> 2) + 4 * (l // 32)][n = l % 32]

The host-side weight packings of the fused layer tail (leann_amd/encoder.py: fused_mlp_k_permutation, pack_w1_acc_order) are checked
against this layout in tests/test_mlp_emulation.py; the kernels themselves run thread-per-lane in tests/hip_emul.  Test infrastructure only.
"""
import numpy as np

LANES = 64


def mfma_32x32x16(a: np.ndarray, b: np.ndarray, c: np.ndarray) -> np.ndarray:
    """a, b: [64, 8] (fp16 values), c: [64, 16] fp32 -> d [64, 16] fp32."""
    lane = np.arange(LANES)
    amat = np.zeros((32, 16), np.float64)
    bmat = np.zeros((16, 32), np.float64)
    for e in range(8):
        amat[lane % 32, 8 * (lane // 32) + e] = a[:, e]
        bmat[8 * (lane // 32) + e, lane % 32] = b[:, e]
    prod = amat @ bmat  # [m, n]
    d = np.array(c, dtype=np.float64, copy=True)
    for r in range(16):
        m = (r & 3) + 8 * (r >> 2) + 4 * (lane // 32)
        d[:, r] += prod[m, lane % 32]
    return d.astype(np.float32)


ML_H = 384  # the hidden size of the hidden-384 kernels (csrc/lm_h384_common.h)
