// lm_encoder_ops.hip -- hand-written fused elementwise/normalisation kernels for the encoder forward
// (the GEMMs and attention stay in hipBLASLt / SDPA through PyTorch; MFMA is used only there).
//
//   lm_add_layernorm_f16 : out = LayerNorm(x + residual) * gamma + beta      (fp16 in/out, fp32 math)
// replaces torch's `x + y` kernel followed by vectorized_layer_norm_kernel, which runs at ~1.1 TB/s
// for hidden=384 (rocprofv3, profiles/r1_encoder_packed_kernel_stats.csv); this kernel is one pass:
// 2 reads + 1 write of [rows, H] fp16, one 64-lane wave per row, 16-byte loads, wave-shuffle reductions.
// Role in the reference: part of compute_embeddings' BERT forward (leann/embedding_compute.py:229-239).
#include <cstdlib>

#include <hip/hip_fp16.h>

#include "lm_internal.h"

#ifndef LM_HOST_EMULATION  // tests/hip_emul compiles only the MFMA kernels of this file for the host
namespace lm {

template <int NV>  // NV = ceil(H / 512): 16-byte vectors (8 halfs) per lane
__global__ __launch_bounds__(256) void k_add_layernorm_f16(const __half* __restrict__ x, const __half* __restrict__ res,
                                                           const __half* __restrict__ gamma, const __half* __restrict__ beta,
                                                           __half* __restrict__ out, int64_t rows, int H, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nvec = H >> 3;  // H % 8 == 0
    float v[NV][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nvec) {
            uint4 a = ((const uint4*)(x + row * H))[c];
            const __half2* ah = (const __half2*)&a;
            if (res) {
                uint4 b = ((const uint4*)(res + row * H))[c];
                const __half2* bh = (const __half2*)&b;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float2 fa = __half22float2(ah[j]), fb = __half22float2(bh[j]);
                    v[i][2 * j] = fa.x + fb.x;
                    v[i][2 * j + 1] = fa.y + fb.y;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float2 fa = __half22float2(ah[j]);
                    v[i][2 * j] = fa.x;
                    v[i][2 * j + 1] = fa.y;
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += v[i][j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
        }
    }
    for (int m = 32; m >= 1; m >>= 1) sum += __shfl_xor(sum, m);
    const float mean = sum / (float)H;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nvec) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float d = v[i][j] - mean;
                sq += d * d;
            }
        }
    }
    for (int m = 32; m >= 1; m >>= 1) sq += __shfl_xor(sq, m);
    const float rstd = rsqrtf(sq / (float)H + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        if (c < nvec) {
            uint4 g = ((const uint4*)gamma)[c], b = ((const uint4*)beta)[c], o;
            const __half2* gh = (const __half2*)&g;
            const __half2* bh = (const __half2*)&b;
            __half2* oh = (__half2*)&o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float2 fg = __half22float2(gh[j]), fb = __half22float2(bh[j]);
                float y0 = (v[i][2 * j] - mean) * rstd * fg.x + fb.x;
                float y1 = (v[i][2 * j + 1] - mean) * rstd * fg.y + fb.y;
                oh[j] = __floats2half2_rn(y0, y1);
            }
            ((uint4*)(out + row * H))[c] = o;
        }
    }
}

}  // namespace lm

extern "C" int lm_add_layernorm_f16(const void* d_x, const void* d_residual, const void* d_gamma, const void* d_beta,
                                    void* d_out, int64_t rows, int32_t hidden, float eps, void* stream) {
    using namespace lm;
    if (rows == 0) return LM_OK;
    if (!d_x || !d_gamma || !d_beta || !d_out || rows < 0) LM_FAIL(LM_EINVAL, "bad add_layernorm arguments");
    if (hidden <= 0 || hidden % 8 || hidden > 2048) LM_FAIL(LM_EINVAL, "hidden must be a multiple of 8, <= 2048");
    {
        const char* rev = getenv("LEANN_MI355X_LN");  // default: 16 lanes per row (lm_encoder_ops2.hip); "1" = first generation (A/B)
        if (!(rev && rev[0] == '1' && rev[1] == 0) && hidden <= 768)
            return lm_add_layernorm_r16_launch(d_x, d_residual, d_gamma, d_beta, d_out, rows, hidden, eps, stream);
    }
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const __half *x = (const __half*)d_x, *r = (const __half*)d_residual, *g = (const __half*)d_gamma, *b = (const __half*)d_beta;
    __half* o = (__half*)d_out;
    const int nv = (hidden / 8 + 63) / 64;
    switch (nv) {
        case 1: hipLaunchKernelGGL((k_add_layernorm_f16<1>), grid, block, 0, st, x, r, g, b, o, rows, hidden, eps); break;
        case 2: hipLaunchKernelGGL((k_add_layernorm_f16<2>), grid, block, 0, st, x, r, g, b, o, rows, hidden, eps); break;
        case 3: hipLaunchKernelGGL((k_add_layernorm_f16<3>), grid, block, 0, st, x, r, g, b, o, rows, hidden, eps); break;
        default: hipLaunchKernelGGL((k_add_layernorm_f16<4>), grid, block, 0, st, x, r, g, b, o, rows, hidden, eps); break;
    }
    LM_HIP(hipGetLastError());
    return LM_OK;
}

#endif  // LM_HOST_EMULATION

// =============================================================================================
// lm_attn_varlen_hd32_f16 -- fused self-attention for packed variable-length sequences,
// head_dim = 32 (MiniLM-L6 / bge-small: 384 / 12), sequence length <= 256, fp16 in/out.
//
// One 256-thread workgroup per (sequence, head).  K rows and V^T are staged once in LDS (row
// strides padded to 80 B / T+4 halfs: conflict-free for the ds_read_b128 / ds_read_b64 fragment
// reads); each wave owns 32-row Q blocks.  Scores are computed SWAPPED, S^T = K Q^T with
// v_mfma_f32_32x32x16_f16, so that a lane holds 16 keys x 1 query row per 32-key tile: softmax
// max/sum are in-lane reductions plus one exchange with lane^32, and the packed P registers are
// directly the B operand of the second MFMA  O^T = V^T P^T  (the k-slot -> key assignment of an
// MFMA operand is free as long as A and B use the same one), no cross-lane movement, no online
// rescaling (T <= 256: all scores of a row live in registers).
// Replaces torch's generic flash kernel (attn_fwd: ~20 % of the forward at hd=32,
// profiles/r1_encoder_packed_kernel_stats.csv).  Part of compute_embeddings' BERT forward
// (leann/embedding_compute.py:229-239).
// =============================================================================================
namespace lm {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

constexpr int ATT_HD = 32;
constexpr int ATT_KSTRIDE = 40;  // halfs per K row in LDS (80 B)

template <int NT>  // NT = number of 32-key tiles (max_len <= 32*NT), 1..8
__global__ __launch_bounds__(256) void k_attn_varlen_hd32(const __half* __restrict__ qkv, const int32_t* __restrict__ cu,
                                                          __half* __restrict__ out, int heads, float scale_log2e) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int seq = blockIdx.x / heads, h = blockIdx.x % heads;
    const int tok0 = cu[seq];
    const int len = cu[seq + 1] - tok0;
    const int H = heads * ATT_HD;
    const int64_t rstride = 3 * (int64_t)H;  // halfs per token row of qkv
    const int Tp = 32 * NT;
    const int VSTRIDE = Tp + 4;
    _Float16* Ks = (_Float16*)smem;                  // [Tp][ATT_KSTRIDE]
    _Float16* Vt = Ks + (size_t)Tp * ATT_KSTRIDE;    // [32][VSTRIDE]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const _Float16* base = (const _Float16*)qkv + (int64_t)tok0 * rstride + h * ATT_HD;

    // ---- stage K (row major, padded) and V^T (transposed, padded); rows >= len are zero ----
    for (int c = tid; c < Tp * 4; c += 256) {
        const int key = c >> 2, part = c & 3;
        half8 kv = {0, 0, 0, 0, 0, 0, 0, 0}, vv = {0, 0, 0, 0, 0, 0, 0, 0};
        if (key < len) {
            kv = *(const half8*)(base + (int64_t)key * rstride + H + part * 8);
            vv = *(const half8*)(base + (int64_t)key * rstride + 2 * H + part * 8);
        }
        *(half8*)(Ks + key * ATT_KSTRIDE + part * 8) = kv;
#pragma unroll
        for (int i = 0; i < 8; ++i) Vt[(part * 8 + i) * VSTRIDE + key] = vv[i];
    }
    __syncthreads();

    const int r31 = lane & 31, g = lane >> 5;
    for (int qb = wv; qb * 32 < len; qb += 4) {
        // Q^T fragment (B operand): lane (n = q row, g) holds hd slots 16*ks + 8*g .. +8
        const int qrow = qb * 32 + r31;
        half8 qf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            qf[ks] = qrow < len ? *(const half8*)(base + (int64_t)qrow * rstride + ks * 16 + g * 8) : z;
        }
        // ---- online softmax over chunks of CH 32-key tiles (32 score registers per lane) ----
        constexpr int CH = NT < 2 ? NT : 2;
        float mx = -3.0e38f, sum = 0.f;
        float16v o = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
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
                    for (int ks = 0; ks < 2; ++ks) {
                        half8 kf = *(const half8*)(Ks + (t * 32 + r31) * ATT_KSTRIDE + ks * 16 + g * 8);  // A: m = key
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], acc, 0, 0, 0);
                    }
                }
                s[tt] = acc;
            }
            float cm = -3.0e38f;
#pragma unroll
            for (int tt = 0; tt < CH; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = 32 * (c0 + tt) + (r & 3) + 8 * (r >> 2) + 4 * g;
                    float v = key < len ? s[tt][r] : -3.0e38f;
                    s[tt][r] = v;
                    cm = fmaxf(cm, v);
                }
            cm = fmaxf(cm, __shfl_xor(cm, 32));
            const float mnew = fmaxf(mx, cm);
            const float alpha = __builtin_amdgcn_exp2f((mx - mnew) * scale_log2e);  // 0 on the first chunk (mx = -3e38)
            mx = mnew;
            float cs = 0.f;
#pragma unroll
            for (int tt = 0; tt < CH; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float p = __builtin_amdgcn_exp2f((s[tt][r] - mx) * scale_log2e);
                    s[tt][r] = p;
                    cs += p;
                }
            cs += __shfl_xor(cs, 32);
            sum = sum * alpha + cs;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] *= alpha;
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
                        const _Float16* vrow = Vt + r31 * VSTRIDE + 32 * t + 16 * u + 4 * g;
                        half4 v0 = *(const half4*)vrow, v1 = *(const half4*)(vrow + 8);
                        half8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                        o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o, 0, 0, 0);
                    }
                }
            }
        }
        const float inv = 1.0f / sum;
        // lane (q = r31, g) holds d = (reg&3) + 8*(reg>>2) + 4g
        if (qrow < len) {
            _Float16* orow = (_Float16*)out + (int64_t)(tok0 + qrow) * H + h * ATT_HD;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                half4 w = {(_Float16)(o[4 * r4] * inv), (_Float16)(o[4 * r4 + 1] * inv), (_Float16)(o[4 * r4 + 2] * inv),
                           (_Float16)(o[4 * r4 + 3] * inv)};
                *(half4*)(orow + 8 * r4 + 4 * g) = w;
            }
        }
    }
}

}  // namespace lm

#ifndef LM_HOST_EMULATION
extern "C" int lm_attn_varlen_hd32_f16(const void* d_qkv, const int32_t* d_cu_seqlens, int32_t n_seqs, int32_t heads,
                                       int32_t max_len, void* d_out, void* stream) {
    using namespace lm;
    if (n_seqs == 0) return LM_OK;
    if (!d_qkv || !d_cu_seqlens || !d_out || n_seqs < 0 || heads <= 0) LM_FAIL(LM_EINVAL, "bad attention arguments");
    if (max_len <= 0 || max_len > 256) LM_FAIL(LM_EINVAL, "lm_attn_varlen_hd32_f16 supports sequence lengths 1..256");
    {
        const char* rev = getenv("LEANN_MI355X_ATTN");  // default: revision 2 (lm_attn_v2.hip); "1" = revision 1 (A/B)
        if (!(rev && rev[0] == '1' && rev[1] == 0)) return lm_attn_v2_launch(d_qkv, d_cu_seqlens, n_seqs, heads, max_len, d_out, stream);
    }
    const int nt = (max_len + 31) / 32;
    const size_t shmem = ((size_t)32 * nt * ATT_KSTRIDE + (size_t)32 * (32 * nt + 4)) * 2;
    const float scale_log2e = 1.4426950408889634f / sqrtf((float)ATT_HD);
    dim3 grid((unsigned)(n_seqs * heads)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const __half* q = (const __half*)d_qkv;
    __half* o = (__half*)d_out;
    switch (nt) {
#define CASEA(n) case n: hipLaunchKernelGGL((k_attn_varlen_hd32<n>), grid, block, shmem, st, q, d_cu_seqlens, o, heads, scale_log2e); break
        CASEA(1); CASEA(2); CASEA(3); CASEA(4); CASEA(5); CASEA(6); CASEA(7); CASEA(8);
#undef CASEA
    }
    LM_HIP(hipGetLastError());
    return LM_OK;
}
#endif  // LM_HOST_EMULATION
