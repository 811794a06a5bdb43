"""Lane-level CPU emulation of the MFMA data flow used by the hand-written encoder kernels.

``mfma_32x32x16`` implements the operand / result layout of ``v_mfma_f32_32x32x16_f16`` as the kernels use it
(the same conventions csrc/lm_encoder_ops.hip: k_attn_varlen_hd32 relies on, which is validated on hardware):

    A operand  lane l, element e  ->  A[m = l % 32][k = 8 * (l // 32) + e]
    B operand  lane l, element e  ->  B[k = 8 * (l // 32) + e][n = l % 32]
    D result   lane l, register r ->  D[m = (r & 3) + 8 * (r >> 2) + 4 * (l // 32)][n = l % 32]

``emulate_mlp_wave`` replays csrc/lm_mlp_fused.hip: k_mlp_fused_h384 for one 32-token wave with exactly the index
expressions of the kernel (fragment addresses into byte-accurate LDS images, the staged chunk -> LDS offset
formulas, the bias -> accumulator mapping, the in-place GELU -> B fragment packing against the host-side W2
permutation, and the epilogue feature mapping).  It checks the *index algebra*, not the instruction schedule.
Test infrastructure only.
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


# ---- constants of csrc/lm_mlp_fused.hip ----
ML_H = 384
ML_KS = ML_H // 16
ML_NJ = ML_H // 32
ML_W1_STRIDE = ML_H + 8
ML_W2_STRIDE = 40
ML_W1_BYTES = 32 * ML_W1_STRIDE * 2
ML_W2_BYTES = ML_H * ML_W2_STRIDE * 2
ML_BUF = ML_W1_BYTES + ML_W2_BYTES
ML_CHUNKS = 32 * ML_H * 2 // 16
ML_NPRE = ML_CHUNKS // 256


def stage_slab(w1_bytes: np.ndarray, w2p_bytes: np.ndarray, s: int) -> np.ndarray:
    """One LDS stage image (uint8[ML_BUF]) filled the way the 256 threads do: thread t copies 16-byte chunks
    t + 256 i of each matrix' slab s to off1[i] / off2[i]."""
    lds = np.zeros(ML_BUF, np.uint8)
    for tid in range(256):
        for i in range(ML_NPRE):
            c = tid + 256 * i
            off1 = (c // 48) * (ML_W1_STRIDE * 2) + (c % 48) * 16
            off2 = ML_W1_BYTES + (c >> 2) * (ML_W2_STRIDE * 2) + (c & 3) * 16
            src = (s * ML_CHUNKS + c) * 16
            lds[off1:off1 + 16] = w1_bytes[src:src + 16]
            lds[off2:off2 + 16] = w2p_bytes[src:src + 16]
    return lds


def _half8_at(lds_halfs: np.ndarray, idx: np.ndarray) -> np.ndarray:
    """Per-lane 8-half fragment starting at half index idx[lane] (must be 16-byte aligned, as ds_read_b128 needs)."""
    assert np.all(idx % 8 == 0)
    return np.stack([lds_halfs[idx + e] for e in range(8)], axis=1)


def emulate_mlp_wave(x, w1, b1, w2p, b2, gamma, beta, eps, gelu):
    """x: [32, 384] fp16 tokens of one wave; w1 [F, 384] fp16; w2p: packed [F/32, 384, 32] fp16; biases fp32.
    Returns y [32, 384] fp16 as the kernel's lanes would store it."""
    F = w1.shape[0]
    lane = np.arange(LANES)
    r31, g = lane % 32, lane // 32
    w1_bytes = np.ascontiguousarray(w1).view(np.uint8).reshape(-1)
    w2p_bytes = np.ascontiguousarray(w2p).view(np.uint8).reshape(-1)
    # x^T fragments: lane (token r31, g) holds x[token][16 ks + 8 g .. + 8]
    xf = [np.stack([x[r31, 16 * ks + 8 * g + e] for e in range(8)], axis=1) for ks in range(ML_KS)]
    o = [np.zeros((LANES, 16), np.float32) for _ in range(ML_NJ)]
    for s in range(F // 32):
        cur = stage_slab(w1_bytes, w2p_bytes, s).view(np.float16)  # half-indexed view of the stage
        # acc <- bias: registers 4q .. 4q+3 <- b1s[32 s + 4 g + 8 q .. + 4]
        acc = np.zeros((LANES, 16), np.float32)
        for q in range(4):
            for i in range(4):
                acc[:, 4 * q + i] = b1[32 * s + 4 * g + 8 * q + i]
        w1s = r31 * ML_W1_STRIDE + 8 * g
        for ks in range(ML_KS):
            acc = mfma_32x32x16(_half8_at(cur, w1s + 16 * ks), xf[ks], acc)
        pf = [np.zeros((LANES, 8), np.float16) for _ in range(2)]
        for u in range(2):
            for jj in range(8):
                pf[u][:, jj] = gelu(acc[:, 8 * u + jj]).astype(np.float16)
        w2s = ML_W1_BYTES // 2 + r31 * ML_W2_STRIDE + 8 * g
        for n in range(2 * ML_NJ):
            u, j = n // ML_NJ, n % ML_NJ
            o[j] = mfma_32x32x16(_half8_at(cur, w2s + 32 * j * ML_W2_STRIDE + 16 * u), pf[u], o[j])
    # epilogue: lane (token r31, g), tile j, register 4q + i <-> feature 32 j + 8 q + 4 g + i
    y = np.zeros((32, ML_H), np.float16)
    tot = np.zeros(LANES, np.float32)
    for j in range(ML_NJ):
        for q in range(4):
            for i in range(4):
                f = 32 * j + 8 * q + 4 * g + i
                v = o[j][:, 4 * q + i] + (x[r31, f].astype(np.float32) + b2[f])
                o[j][:, 4 * q + i] = v
                tot += v
    tot = tot + tot[lane ^ 32]
    mean = tot / ML_H
    sq = np.zeros(LANES, np.float32)
    for j in range(ML_NJ):
        for r in range(16):
            d = o[j][:, r] - mean
            sq += d * d
    sq = sq + sq[lane ^ 32]
    rstd = 1.0 / np.sqrt(sq / ML_H + eps)
    for j in range(ML_NJ):
        for q in range(4):
            for i in range(4):
                f = 32 * j + 8 * q + 4 * g + i
                val = (o[j][:, 4 * q + i] - mean) * rstd * gamma[f].astype(np.float32) + beta[f].astype(np.float32)
                y[r31, f] = val.astype(np.float16)
    return y


def _stage_store(lds: np.ndarray, base: int, which: int, src_bytes: np.ndarray, slab: int) -> None:
    """All 256 threads store their 6 chunks of one matrix' slab (which = 1: W1 -> off1, 2: W2 -> off2) at stage base."""
    for tid in range(256):
        for i in range(ML_NPRE):
            c = tid + 256 * i
            off = (c // 48) * (ML_W1_STRIDE * 2) + (c % 48) * 16 if which == 1 else ML_W1_BYTES + (c >> 2) * (ML_W2_STRIDE * 2) + (c & 3) * 16
            src = (slab * ML_CHUNKS + c) * 16
            lds[base + off:base + off + 16] = src_bytes[src:src + 16]


def emulate_mlp_wave_pipelined(x, w1, b1, w2p, b2, gamma, beta, eps, gelu):
    """k_mlp_fused_h384_p: same arithmetic, but W1 is staged one slab ahead of W2 and the first product of slab
    s+1 is computed during iteration s.  The two LDS stages are modelled as one byte array that is only written at
    the program points where the kernel writes it, so a wrong stage parity or slab index shows up as a wrong result."""
    F = w1.shape[0]
    nslab = F // 32
    lane = np.arange(LANES)
    r31, g = lane % 32, lane // 32
    w1_bytes = np.ascontiguousarray(w1).view(np.uint8).reshape(-1)
    w2p_bytes = np.ascontiguousarray(w2p).view(np.uint8).reshape(-1)
    xf = [np.stack([x[r31, 16 * ks + 8 * g + e] for e in range(8)], axis=1) for ks in range(ML_KS)]
    lds = np.full(2 * ML_BUF, 0xEE, np.uint8)  # poison: reading a stage that was never written gives garbage
    _stage_store(lds, 0, 1, w1_bytes, 0)
    _stage_store(lds, 0, 2, w2p_bytes, 0)
    if nslab > 1:
        _stage_store(lds, ML_BUF, 1, w1_bytes, 1)

    def first_product(stage_base: int, slab: int) -> np.ndarray:
        acc = np.zeros((LANES, 16), np.float32)
        for q in range(4):
            for i in range(4):
                acc[:, 4 * q + i] = b1[32 * slab + 4 * g + 8 * q + i]
        halfs = lds.view(np.float16)
        w1s = stage_base // 2 + r31 * ML_W1_STRIDE + 8 * g
        for ks in range(ML_KS):
            acc = mfma_32x32x16(_half8_at(halfs, w1s + 16 * ks), xf[ks], acc)
        return acc

    o = [np.zeros((LANES, 16), np.float32) for _ in range(ML_NJ)]
    accn = first_product(0, 0)
    for s in range(nslab):
        cur, oth = (s & 1) * ML_BUF, ((s + 1) & 1) * ML_BUF
        more, more2 = s + 1 < nslab, s + 2 < nslab
        acc = accn.copy()
        if more:
            accn = first_product(oth, s + 1)
        pf = [np.zeros((LANES, 8), np.float16) for _ in range(2)]
        for pr in range(8):  # pairs 0..5 behind the first product, 6..7 behind the second: same values either way
            for e in range(2):
                pf[pr // 4][:, 2 * (pr % 4) + e] = gelu(acc[:, 2 * pr + e]).astype(np.float16)
        if more2:
            _stage_store(lds, cur, 1, w1_bytes, s + 2)
        halfs = lds.view(np.float16)
        w2s = (cur + ML_W1_BYTES) // 2 + r31 * ML_W2_STRIDE + 8 * g
        for n in range(2 * ML_NJ):
            u, j = n // ML_NJ, n % ML_NJ
            o[j] = mfma_32x32x16(_half8_at(halfs, w2s + 32 * j * ML_W2_STRIDE + 16 * u), pf[u], o[j])
        if more:
            _stage_store(lds, oth, 2, w2p_bytes, s + 1)
    # epilogue identical to emulate_mlp_wave
    y = np.zeros((32, ML_H), np.float16)
    tot = np.zeros(LANES, np.float32)
    for j in range(ML_NJ):
        for q in range(4):
            for i in range(4):
                f = 32 * j + 8 * q + 4 * g + i
                v = o[j][:, 4 * q + i] + (x[r31, f].astype(np.float32) + b2[f])
                o[j][:, 4 * q + i] = v
                tot += v
    tot = tot + tot[lane ^ 32]
    mean = tot / ML_H
    sq = np.zeros(LANES, np.float32)
    for j in range(ML_NJ):
        for r in range(16):
            d = o[j][:, r] - mean
            sq += d * d
    sq = sq + sq[lane ^ 32]
    rstd = 1.0 / np.sqrt(sq / ML_H + eps)
    for j in range(ML_NJ):
        for q in range(4):
            for i in range(4):
                f = 32 * j + 8 * q + 4 * g + i
                val = (o[j][:, 4 * q + i] - mean) * rstd * gamma[f].astype(np.float32) + beta[f].astype(np.float32)
                y[r31, f] = val.astype(np.float16)
    return y


# ---- csrc/lm_linear_h384.hip ----
LN_STRIDE = 40
LN_BUF = ML_H * LN_STRIDE * 2
LN_CHUNKS = ML_H * 32 * 2 // 16
LN_NPRE = LN_CHUNKS // 256
LN_SLABS = ML_H // 32


def emulate_linear_wave(x, wp, bias):
    """k_linear_h384<0> for one 32-token wave: x [32, 384] fp16, wp packed [P, 12, 384, 32] fp16, bias fp32 [384 P].
    Returns out [32, 384 P] fp16.  The two LDS stages are written only where the kernel writes them."""
    P = wp.shape[0]
    lane = np.arange(LANES)
    r31, g = lane % 32, lane // 32
    wbytes = np.ascontiguousarray(wp).view(np.uint8).reshape(-1)
    xf = [np.stack([x[r31, 16 * ks + 8 * g + e] for e in range(8)], axis=1) for ks in range(ML_KS)]
    lds = np.full(2 * LN_BUF, 0xEE, np.uint8)

    def store(stage: int, slab: int) -> None:
        for tid in range(256):
            for i in range(LN_NPRE):
                c = tid + 256 * i
                off = (c >> 2) * (LN_STRIDE * 2) + (c & 3) * 16
                src = (slab * LN_CHUNKS + c) * 16
                lds[stage * LN_BUF + off:stage * LN_BUF + off + 16] = wbytes[src:src + 16]

    store(0, 0)
    out = np.zeros((32, ML_H * P), np.float16)
    ws = r31 * LN_STRIDE + 8 * g
    for p in range(P):
        o = [np.zeros((LANES, 16), np.float32) for _ in range(ML_NJ)]
        for s in range(LN_SLABS):
            halfs = lds.view(np.float16)
            cur = ws + (s & 1) * (LN_BUF // 2)
            more = s + 1 < LN_SLABS or p + 1 < P
            for n in range(2 * ML_NJ):
                u, j = n // ML_NJ, n % ML_NJ
                o[j] = mfma_32x32x16(_half8_at(halfs, cur + 32 * j * LN_STRIDE + 16 * u), xf[2 * s + u], o[j])
            if more:
                store((s + 1) & 1, LN_SLABS * p + s + 1)
        for j in range(ML_NJ):
            for q in range(4):
                for i in range(4):
                    col = ML_H * p + 32 * j + 8 * q + 4 * g + i
                    out[r31, col] = (o[j][:, 4 * q + i] + bias[col]).astype(np.float16)
    return out
