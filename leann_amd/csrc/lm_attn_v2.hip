// lm_attn_v2.hip -- fused self-attention for packed variable-length sequences, fp16 in/out: head_dim 32 (MiniLM-L6 / bge-small:
// 384 / 12, lengths <= 256) and 64 (bge-base / contriever, lengths <= 512).
//
// One 256-thread workgroup per (sequence, head).  K rows and V^T are staged once in LDS (row strides padded: conflict-free fragment
// reads); each wave owns 32-row Q blocks.  Scores are computed SWAPPED, S^T = K Q^T with v_mfma_f32_32x32x16_f16, so that a lane
// holds 16 keys x 1 query row per 32-key tile: softmax max / sum are in-lane reductions plus one exchange with lane ^ 32, and the
// packed P registers are directly the B operand of the second MFMA  O^T = V^T P^T  (the k-slot -> key assignment of an MFMA operand
// is free as long as A and B use the same one): no cross-lane movement.  Online softmax over chunks of two 32-key tiles.
// Role in the reference: part of compute_embeddings' BERT forward (leann/embedding_compute.py:229-239).
//
// At head_dim 32 a 32 x 32 score tile costs 2 MFMAs (64 cycles) against 16 registers of softmax: the kernel is VALU bound, not
// MFMA / LDS / HBM bound (the first revision of this kernel spent ~17 VALU issue slots per score; it is gone).  The softmax diet:
//   * key masking (compare + select per score) only in tiles that straddle the sequence end -- a
//     workgroup-uniform branch; full tiles need none because K rows are real;
//   * (s - max) * c  ->  one fused multiply-add with a per-chunk constant, issued as v_pk_fma_f32 on pairs;
//   * row sums accumulate with v_pk_add_f32 on pairs;
//   * MFMA results stay in VGPRs (-mllvm -amdgpu-mfma-vgpr-form, this file only): no v_accvgpr copies;
//   * K / V staging issues all global loads of a thread before the first LDS store (one HBM latency per
//     workgroup instead of one per 256-row slice), and V^T is written as packed key pairs (ds_write_b32).
// Role in the reference: part of compute_embeddings' BERT forward (leann/embedding_compute.py:229-239).
#include <hip/hip_fp16.h>

#include <cstdlib>

#include "lm_internal.h"

namespace lm {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float2v __attribute__((ext_vector_type(2)));

// HD = head dimension: 32 (hidden 384 = 12 x 32: MiniLM, bge-small) or 64 (hidden 768 = 12 x 64: bge-base, contriever).  K rows are
// padded by 8 halfs in LDS (80 / 144 bytes: consecutive rows fall on different 16-byte slots of the 256-byte bank row, so the
// ds_read_b128 fragment reads are conflict free).  At HD = 64 the score tile takes four MFMA k-steps instead of two and the
// output is two 32 x 32 accumulator tiles (d = 0..31, 32..63); the softmax (the VALU-bound part) is unchanged, so the kernel's
// MFMA share doubles.
template <int NT, int HD = 32>  // NT = number of 32-key tiles (max_len <= 32*NT), 1..8 (HD = 64: ..16)
__global__ __launch_bounds__(256) void k_attn_varlen_hd32_v2(const __half* __restrict__ qkv, const int32_t* __restrict__ cu,
                                                             __half* __restrict__ out, int heads, float scale_log2e, int n_units) {
    extern __shared__ __align__(16) unsigned char smem[];
    // XCD-aware unit order.  Workgroup b runs on XCD b % 8 and every XCD has its own L2; a (sequence, head) unit reads 64-byte
    // row segments of the [T][3H] activations, i.e. HALF of each 128-byte line -- the other half belongs to the neighbouring
    // head.  With the plain order (unit = b) neighbouring heads sit on different XCDs and every line is fetched from HBM twice
    // (round-2 counters: 805 MB fetched for 403 MB of operands at 131k tokens, the kernel ran AT the HBM read rate).  So the
    // units are dealt out in contiguous runs per XCD: XCD x works through units [x * per, (x + 1) * per) in dispatch order, and
    // the twelve heads of a sequence are resident on one XCD at about the same time.  (grid = 8 * per >= n_units.)
    // n_units < 0: plain order over -n_units units (LEANN_MI355X_ATTN_XCD=0, kept for A/B timing).
    const int per = gridDim.x >> 3;
    const int unit = n_units < 0 ? (int)blockIdx.x : (int)((blockIdx.x & 7) * per + (blockIdx.x >> 3));
    if (unit >= (n_units < 0 ? -n_units : n_units)) return;
    const int seq = unit / heads, h = unit % heads;
    const int tok0 = cu[seq];
    const int len = cu[seq + 1] - tok0;
    constexpr int A2_HD = HD, A2_KSTRIDE = HD + 8, PARTS = HD / 8, KS = HD / 16, DT = HD / 32;
    const int H = heads * A2_HD;
    const int64_t rstride = 3 * (int64_t)H;  // halfs per token row of qkv
    constexpr int Tp = 32 * NT;
    constexpr int VSTRIDE = Tp + 4;                  // halfs per V^T row (even: key pairs are dword aligned)
    _Float16* Ks = (_Float16*)smem;                  // [Tp][A2_KSTRIDE]
    _Float16* Vt = Ks + (size_t)Tp * A2_KSTRIDE;     // [HD][VSTRIDE]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const _Float16* base = (const _Float16*)qkv + (int64_t)tok0 * rstride + h * A2_HD;

    // ---- stage K (row major, padded) and V^T (transposed, padded); rows >= len are zero ----
    // work item = (key pair, 8-half part): Tp/2 * PARTS items; all loads of a thread are issued before its stores
    {
        constexpr int ITEMS = Tp / 2 * PARTS, NIT = (ITEMS + 255) / 256;
        half8 k0[NIT], k1[NIT], v0[NIT], v1[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = tid + 256 * it;
            const int key = (c / PARTS) * 2, part = c % PARTS;
            const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            k0[it] = z; k1[it] = z; v0[it] = z; v1[it] = z;
            if (c < ITEMS && key < len) {
                const _Float16* p = base + (int64_t)key * rstride + part * 8;
                k0[it] = *(const half8*)(p + H);
                v0[it] = *(const half8*)(p + 2 * H);
                if (key + 1 < len) {
                    k1[it] = *(const half8*)(p + rstride + H);
                    v1[it] = *(const half8*)(p + rstride + 2 * H);
                }
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = tid + 256 * it;
            const int key = (c / PARTS) * 2, part = c % PARTS;
            if (c < ITEMS) {
                *(half8*)(Ks + key * A2_KSTRIDE + part * 8) = k0[it];
                *(half8*)(Ks + (key + 1) * A2_KSTRIDE + part * 8) = k1[it];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    half2v pr = {v0[it][i], v1[it][i]};
                    *(half2v*)(Vt + (part * 8 + i) * VSTRIDE + key) = pr;
                }
            }
        }
    }
    __syncthreads();

    const int r31 = lane & 31, g = lane >> 5;
    const float2v c2 = {scale_log2e, scale_log2e};
    for (int qb = wv; qb * 32 < len; qb += 4) {
        // Q^T fragment (B operand): lane (n = q row, g) holds hd slots 16*ks + 8*g .. +8
        const int qrow = qb * 32 + r31;
        half8 qf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            qf[ks] = qrow < len ? *(const half8*)(base + (int64_t)qrow * rstride + ks * 16 + g * 8) : z;
        }
        // per-lane mask limit; the empty asm keeps the compiler from hoisting all 16*NT key comparisons out of the
        // q-block loop (revision 1 does: 176 SGPR pairs spilled to VGPR lanes and a ~700-instruction preamble)
        int lenv = len - 4 * g;
        LM_KEEP_LOCAL(lenv);
        // ---- online softmax over chunks of CH 32-key tiles ----
        constexpr int CH = NT < 2 ? NT : 2;
        float mx = -3.0e38f, sum = 0.f;
        float16v o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
        for (int c0 = 0; c0 < NT; c0 += CH) {
            if (c0 * 32 >= len) break;
            float16v s[CH];
            // S^T tiles: keys x q.  lane (q = r31, g) holds keys 32t + (reg&3) + 8*(reg>>2) + 4g
#pragma unroll
            for (int tt = 0; tt < CH; ++tt) {
                const int t = c0 + tt;
                float16v acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                if (t < NT) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        half8 kf = *(const half8*)(Ks + (t * 32 + r31) * A2_KSTRIDE + ks * 16 + g * 8);  // A: m = key
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], acc, 0, 0, 0);
                    }
                }
                s[tt] = acc;
            }
            // mask keys >= len: only tiles that reach past the sequence end (uniform branch)
#pragma unroll
            for (int tt = 0; tt < CH; ++tt) {
                if (32 * (c0 + tt + 1) > len) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int keyc = 32 * (c0 + tt) + (r & 3) + 8 * (r >> 2);  // + 4g is folded into lenv
                        s[tt][r] = keyc < lenv ? s[tt][r] : -3.0e38f;
                    }
                }
            }
            float cm = s[0][0];
#pragma unroll
            for (int tt = 0; tt < CH; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) cm = fmaxf(cm, s[tt][r]);
            cm = fmaxf(cm, __shfl_xor(cm, 32));
            const float mnew = fmaxf(mx, cm);
            const float alpha = __builtin_amdgcn_exp2f((mx - mnew) * scale_log2e);  // 0 on the first chunk (mx = -3e38)
            mx = mnew;
            const float nb = -mx * scale_log2e;
            const float2v nb2 = {nb, nb};
            float2v cs2 = {0.f, 0.f};
#pragma unroll
            for (int tt = 0; tt < CH; ++tt)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    float2v v = {s[tt][r], s[tt][r + 1]};
                    v = __builtin_elementwise_fma(v, c2, nb2);  // s*c - max*c
                    float2v p = {__builtin_amdgcn_exp2f(v[0]), __builtin_amdgcn_exp2f(v[1])};
                    s[tt][r] = p[0];
                    s[tt][r + 1] = p[1];
                    cs2 += p;
                }
            float cs = cs2[0] + cs2[1];
            cs += __shfl_xor(cs, 32);
            sum = sum * alpha + cs;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            // O^T += V^T P^T : A = V^T (m = d), B = P^T (n = q); k-slots (g, j) <-> keys the lane already holds
#pragma unroll
            for (int tt = 0; tt < CH; ++tt) {
                const int t = c0 + tt;
                if (t < NT) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        half8 pf;
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) pf[jj] = (_Float16)s[tt][8 * u + jj];
                        // regs 8u..8u+3 -> keys 32t+16u+4g+{0..3} ; regs 8u+4..8u+7 -> keys 32t+16u+8+4g+{0..3}
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt) {
                            const _Float16* vrow = Vt + (32 * dt + r31) * VSTRIDE + 32 * t + 16 * u + 4 * g;
                            half4 va = *(const half4*)vrow, vb = *(const half4*)(vrow + 8);
                            half8 vf = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
                            o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o[dt], 0, 0, 0);
                        }
                    }
                }
            }
        }
        const float inv = 1.0f / sum;
        // lane (q = r31, g) holds d = 32 dt + (reg&3) + 8*(reg>>2) + 4g
        if (qrow < len) {
            _Float16* orow = (_Float16*)out + (int64_t)(tok0 + qrow) * H + h * A2_HD;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    half4 w = {(_Float16)(o[dt][4 * r4] * inv), (_Float16)(o[dt][4 * r4 + 1] * inv), (_Float16)(o[dt][4 * r4 + 2] * inv),
                               (_Float16)(o[dt][4 * r4 + 3] * inv)};
                    *(half4*)(orow + 32 * dt + 8 * r4 + 4 * g) = w;
                }
        }
    }
}

}  // namespace lm

#ifndef LM_HOST_EMULATION
// launched by lm_attn_varlen_hd32_f16 / lm_attn_varlen_f16 (below)
template <int HD>
static int attn_v2_launch_hd(const void* d_qkv, const int32_t* d_cu_seqlens, int32_t n_seqs, int32_t heads, int32_t max_len, void* d_out,
                             void* stream) {
    using namespace lm;
    if constexpr (HD == 32) {  // generation 3 (lm_attn_v3.hip) is the default at head_dim 32 since round 5; LEANN_MI355X_ATTN=2 = this file's kernel (A/B)
        const char* gen = getenv("LEANN_MI355X_ATTN");
        const char* var = getenv("LEANN_MI355X_ATTN3");  // (9 = generation 2 as well: a switch that leaves the one-call forwards on, for A/B runs of bench.py)
        if (!(gen && gen[0] == '2') && !(var && var[0] == '9')) return lm_attn_v3_launch_hd32(d_qkv, d_cu_seqlens, n_seqs, heads, max_len, d_out, 0, stream);
    }
    const int nt = (max_len + 31) / 32;
    const size_t shmem = ((size_t)32 * nt * (HD + 8) + (size_t)HD * (32 * nt + 4)) * 2;
    const float scale_log2e = 1.4426950408889634f / sqrtf((float)HD);
    const char* xo = getenv("LEANN_MI355X_ATTN_XCD");
    const int n_units = (xo && xo[0] == '0') ? -(n_seqs * heads) : n_seqs * heads;
    dim3 grid((unsigned)((n_seqs * heads + 7) / 8 * 8)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const __half* q = (const __half*)d_qkv;
    __half* o = (__half*)d_out;
    kt_attn_work(d_cu_seqlens, n_seqs, heads * HD, stream);  // the flops depend on the sequence lengths (device memory): summed there
    KtScope kt(LM_KT_ATTN, stream, 0.0);
    switch (nt) {
#define CASEA(n)                                                                                                                         \
    case n: {                                                                                                                            \
        if (shmem > 48 * 1024) {                                                                                                         \
            static DynLdsAttr attr;                                                                                                      \
            LM_HIP(ensure_dyn_lds(attr, (const void*)k_attn_varlen_hd32_v2<n, HD>, shmem));                                              \
        }                                                                                                                                \
        hipLaunchKernelGGL((k_attn_varlen_hd32_v2<n, HD>), grid, block, shmem, st, q, d_cu_seqlens, o, heads, scale_log2e, n_units);     \
    } break
        CASEA(1); CASEA(2); CASEA(3); CASEA(4); CASEA(5); CASEA(6); CASEA(7); CASEA(8);
        default:
            if constexpr (HD == 64) {
                switch (nt) {
                    CASEA(9); CASEA(10); CASEA(11); CASEA(12); CASEA(13); CASEA(14); CASEA(15); CASEA(16);
                    default: LM_FAIL(LM_EINVAL, "lm_attn_varlen_f16 (head_dim 64) supports sequence lengths 1..512");
                }
            } else {
                LM_FAIL(LM_EINVAL, "lm_attn_varlen_hd32_f16 supports sequence lengths 1..256");
            }
#undef CASEA
    }
    LM_HIP(hipGetLastError());
    return LM_OK;
}

extern "C" int lm_attn_varlen_hd32_f16(const void* d_qkv, const int32_t* d_cu_seqlens, int32_t n_seqs, int32_t heads, int32_t max_len,
                                       void* d_out, void* stream) {
    if (n_seqs == 0) return LM_OK;
    if (!d_qkv || !d_cu_seqlens || !d_out || n_seqs < 0 || heads <= 0) LM_FAIL(LM_EINVAL, "bad attention arguments");
    if (max_len <= 0 || max_len > 256) LM_FAIL(LM_EINVAL, "lm_attn_varlen_hd32_f16 supports sequence lengths 1..256");
    return attn_v2_launch_hd<32>(d_qkv, d_cu_seqlens, n_seqs, heads, max_len, d_out, stream);
}

extern "C" int lm_attn_varlen_f16(const void* d_qkv, const int32_t* d_cu_seqlens, int32_t n_seqs, int32_t heads, int32_t head_dim,
                                  int32_t max_len, void* d_out, void* stream) {
    if (n_seqs == 0) return LM_OK;
    if (!d_qkv || !d_cu_seqlens || !d_out || n_seqs < 0 || heads <= 0 || max_len <= 0) LM_FAIL(LM_EINVAL, "bad attention arguments");
    if (head_dim == 32) {
        if (max_len > 256) LM_FAIL(LM_EINVAL, "lm_attn_varlen_f16 (head_dim 32) supports sequence lengths 1..256");
        return attn_v2_launch_hd<32>(d_qkv, d_cu_seqlens, n_seqs, heads, max_len, d_out, stream);
    }
    if (head_dim == 64) {
        if (max_len > 512) LM_FAIL(LM_EINVAL, "lm_attn_varlen_f16 (head_dim 64) supports sequence lengths 1..512");
        return attn_v2_launch_hd<64>(d_qkv, d_cu_seqlens, n_seqs, heads, max_len, d_out, stream);
    }
    LM_FAIL(LM_EINVAL, "lm_attn_varlen_f16: head_dim must be 32 or 64");
}
#endif  // LM_HOST_EMULATION
