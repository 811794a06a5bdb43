// lm_small_layer_h384.hip -- everything of a hidden-384 encoder layer behind its attention, for SMALL forwards, in ONE launch:
//
//     x1  = LayerNorm1( x + a W_o^T + b_o )                         a = attention output, x = the layer's input (residual)
//     x2  = LayerNorm2( x1 + GELU(x1 W1^T + b1) W2^T + b2 )         -> d_out (may be x: in place)
//     qkv = x2 W_qkv'^T + b_qkv'                                    optional: the NEXT layer's QKV projection
//
// Why: a one-query search round recomputes ~10 chunks (1-2 k tokens); its forward is a chain of ~44 dependent launches of ~10 us each and
// that chain IS the latency of the search.  The fused layer tail of the large forwards (lm_layer_tail_h384.hip)
// cannot help: its workgroup is one 64 us dependency chain however few tokens it holds.  Here a 384-thread workgroup (six waves) owns 32
// tokens x ALL features of every intermediate, which stay on the chip: the token tile and x1 / the GELU outputs / x2 live in LDS in the
// XOR-swizzled tile layout the B-fragment reads want (the layout of lm_rowgemm_ln_h384.hip, whose product loop this file reuses four
// times), weights come straight from L2 as A fragments eight k-steps ahead, the two LayerNorms meet through LDS (one pass: sum and sum of
// squares).  Per layer: attention + this kernel = 2 launches instead of 7; per workgroup ~20 k cycles of MFMA chain (fc1 and fc2 6.1 k
// each at ffn 1536, out-projection 1.5 k, QKV 4.6 k).
//
// LDS (ffn 1536: 145.5 KB): R0 [0, 24 K) attention tile, later the x2 tile | R1 [24 K, 48 K) x1 tile | R2 ffn / 384 blocks of 24 KB: GELU
// outputs | 1.5 KB of row statistics.
//
// STATUS: written in round 4 after the GPU budget was spent -- validated in thread-per-lane emulation (tests/emulated_search_cases.py:
// small_layer) against numpy and against the unfused small-forward form, NOT yet run or timed on an MI355X.  Off by default:
// LEANN_MI355X_SMALL_LAYER=1 switches the small-forward form of lm_bert_h384_forward_packed onto it (scripts/next_gpu_session.sh).
// Role in the reference: part of compute_embeddings' BERT forward (leann/embedding_compute.py:229-239).
#include "lm_h384_stream.h"

namespace lm {

constexpr int SL_THREADS = 384;  // six waves
constexpr int SL_PF = 8;         // k-steps of W fragments per chunk (two chunks live)

// exact-erf GELU as in lm_gemm_f16.hip (gm_gelu: one transcendental, |error| < 1e-6) -- the SAME arithmetic, so that this kernel's
// intermediate agrees with the unfused small-forward form's to the last bit before its fp16 rounding
__device__ __forceinline__ float sl_gelu(float x) {
    const float u = fabsf(x);
    float p = fmaf(u, -0.0004881171917077154f, 0.007198805455118418f);
    p = fmaf(p, u, -0.052146803587675095f);
    p = fmaf(p, u, -0.4595957100391388f);
    p = fmaf(p, u, -1.1510006189346313f);
    const float w = __builtin_amdgcn_exp2f(fmaf(p, u, -1.0f));
    return fmaf(-u, w, fmaxf(x, 0.0f));
}

// byte offset of (token row n, features f .. f + 3) inside a 24 KB tile block of 32 tokens x 384 features: 16-byte chunk c = f / 8 of row n
// sits at position (c & ~15) | ((c ^ n) & 15)
__device__ __forceinline__ int sl_tile_off(int n, int f) {
    const int c = f >> 3;
    return n * 768 + (((c & ~15) | ((c ^ n) & 15)) << 4) + (f & 7) * 2;
}

// acc[t] += W[rows of tile t][0 .. 16 nks) x tile^T for the wave's two 32-feature tiles: A fragments (W rows wr0 / wr1: lane = row r31,
// k-group g) from global memory PF k-steps ahead, B fragments (lane = token r31, k-group g) from the LDS tile blocks at xb
template <int PF>
__device__ __forceinline__ void sl_pair_gemm(float16v (&acc)[2], const __half* wr0, const __half* wr1, int nks, const unsigned char* xb, int r31, int g) {
    const int nchunk = nks / PF;
    half8 wa[2][PF][2];
#pragma unroll
    for (int j = 0; j < PF; ++j) {
        wa[0][j][0] = *(const half8*)(wr0 + 16 * j);
        wa[0][j][1] = *(const half8*)(wr1 + 16 * j);
    }
    auto chunk = [&](half8 (&cur)[PF][2], half8 (&nxt)[PF][2], int c) {
        if (c + 1 < nchunk) {
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                nxt[j][0] = *(const half8*)(wr0 + 16 * ((c + 1) * PF + j));
                nxt[j][1] = *(const half8*)(wr1 + 16 * ((c + 1) * PF + j));
            }
        }
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int ks = c * PF + j, b = ks / 24, kk = ks - 24 * b, ch = 2 * kk + g;
            const half8 bf = *(const half8*)(xb + (size_t)b * T4_SLAB + (((ch & ~15) | ((ch ^ r31) & 15)) << 4));
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[j][0], bf, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur[j][1], bf, acc[1], 0, 0, 0);
        }
    };
    for (int c = 0; c < nchunk; c += 2) {
        chunk(wa[0], wa[1], c);
        if (c + 1 < nchunk) chunk(wa[1], wa[0], c + 1);
    }
}

__device__ __forceinline__ void sl_zero(float16v (&acc)[2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
}

// row statistics of v (the wave's 64 features of token r31, both lane halves) over all six waves: mean and 1 / sqrt(var + eps)
__device__ __forceinline__ void sl_row_stats(const float16v (&v)[2], float* red, int wv, int r31, int g, float eps, float& mean, float& rstd) {
    float s = 0.f, s2 = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s += v[t][r];
            s2 += v[t][r] * v[t][r];
        }
    s += __shfl_xor(s, 32);
    s2 += __shfl_xor(s2, 32);
    if (g == 0) {
        red[(wv * 32 + r31) * 2] = s;
        red[(wv * 32 + r31) * 2 + 1] = s2;
    }
    __syncthreads();
    float ts = 0.f, ts2 = 0.f;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        ts += red[(k * 32 + r31) * 2];
        ts2 += red[(k * 32 + r31) * 2 + 1];
    }
    mean = ts * (1.0f / ML_H);
    float var = ts2 * (1.0f / ML_H) - mean * mean;
    var = var > 0.f ? var : 0.f;
    rstd = 1.0f / sqrtf(var + eps);
}

struct SlArgs {
    const __half* attn;   // [T][384]
    const __half* resid;  // [T][384]
    const __half* wo;     // [384][384]
    const float* bo;
    const __half* g1;
    const __half* be1;
    const __half* w1;     // [F][384]
    const float* b1;
    const __half* w2;     // [384][F]
    const float* b2;
    const __half* g2;
    const __half* be2;
    const __half* wqkv;   // [1152][384] of the NEXT layer, or NULL
    const float* bqkv;
    __half* out;          // [T][384]
    __half* qkv;          // [T][1152]
    int T, F;
    float eps1, eps2;
};

template <int PF>
__global__ __launch_bounds__(SL_THREADS) void k_small_layer_h384(SlArgs p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, r31 = lane & 31, g = lane >> 5;
    const int tok0 = (int)blockIdx.x * 32;
    const int rows_valid = p.T - tok0 < 32 ? p.T - tok0 : 32;
    const bool valid = r31 < rows_valid;
    const int nblk = p.F / ML_H;
    unsigned char* R0 = smem;
    unsigned char* R1 = smem + T4_SLAB;
    unsigned char* R2 = smem + 2 * T4_SLAB;
    float* red = (float*)(smem + (size_t)(2 + nblk) * T4_SLAB);
    const int64_t trow = (int64_t)(tok0 + (valid ? r31 : rows_valid - 1)) * ML_H;

    // ---- the attention tile -> R0 (24 coalesced 1 KB pieces, four per wave; rows past the end repeat the last valid one) ----
    {
        const unsigned char* rows = (const unsigned char*)p.attn + (size_t)tok0 * ML_H * 2;
        for (int pp = wv; pp < 24; pp += 6) {
            const int L = 64 * pp + lane, row = L / 48, pos = L - 48 * row;
            const int rc = row < rows_valid ? row : rows_valid - 1;
            lm_dma16_sv(rows, (unsigned)(rc * 768 + (((pos & ~15) | ((pos ^ row) & 15)) << 4)), R0 + 1024 * pp);
        }
    }
    T4_WAIT_VM(0);
    __syncthreads();

    // ---- 1: attention output projection + residual + LayerNorm 1.  The wave owns features 64 wv + 32 t + 8 q + 4 g + i (tile t, register 4 q + i) ----
    float16v x1[2];
    {
        sl_zero(x1);
        const __half* wr = p.wo + (size_t)(64 * wv + r31) * ML_H + 8 * g;
        sl_pair_gemm<PF>(x1, wr, wr + (size_t)32 * ML_H, ML_H / 16, R0 + r31 * 768, r31, g);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int f = 64 * wv + 32 * t + 8 * q + 4 * g;
                const float4v bv = *(const float4v*)(p.bo + f);
                const half4 rv = *(const half4*)(p.resid + trow + f);
#pragma unroll
                for (int i = 0; i < 4; ++i) x1[t][4 * q + i] = x1[t][4 * q + i] + bv[i] + (float)rv[i];
            }
        float mean, rstd;
        sl_row_stats(x1, red, wv, r31, g, p.eps1, mean, rstd);  // (its barrier: every wave is done reading R0)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int f = 64 * wv + 32 * t + 8 * q + 4 * g;
                const half4 gm = *(const half4*)(p.g1 + f), bt = *(const half4*)(p.be1 + f);
                half4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    o[i] = (_Float16)((x1[t][4 * q + i] - mean) * rstd * (float)gm[i] + (float)bt[i]);
                    x1[t][4 * q + i] = (float)o[i];  // the residual of the second LayerNorm is the fp16 x1, as in the unfused form
                }
                *(half4*)(R1 + sl_tile_off(r31, f)) = o;
            }
    }
    __syncthreads();  // R1 (x1) complete

    // ---- 2: fc1 + GELU -> R2.  The wave owns hidden units [wv F / 6, (wv + 1) F / 6) in passes of 64 ----
    {
        const int per_wave = p.F / 6;
        for (int u0 = wv * per_wave; u0 < (wv + 1) * per_wave; u0 += 64) {
            float16v h[2];
            sl_zero(h);
            const __half* wr = p.w1 + (size_t)(u0 + r31) * ML_H + 8 * g;
            sl_pair_gemm<PF>(h, wr, wr + (size_t)32 * ML_H, ML_H / 16, R1 + r31 * 768, r31, g);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int u = u0 + 32 * t + 8 * q + 4 * g;
                    const float4v bv = *(const float4v*)(p.b1 + u);
                    half4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = (_Float16)sl_gelu(h[t][4 * q + i] + bv[i]);
                    const int b = u / ML_H;
                    *(half4*)(R2 + (size_t)b * T4_SLAB + sl_tile_off(r31, u - b * ML_H)) = o;
                }
        }
    }
    __syncthreads();  // R2 (GELU outputs) complete

    // ---- 3: fc2 + residual (x1) + LayerNorm 2 -> out (and, with a following QKV projection, -> R0) ----
    {
        float16v y[2];
        sl_zero(y);
        const __half* wr = p.w2 + (size_t)(64 * wv + r31) * p.F + 8 * g;
        sl_pair_gemm<PF>(y, wr, wr + (size_t)32 * p.F, p.F / 16, R2 + r31 * 768, r31, g);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4v bv = *(const float4v*)(p.b2 + 64 * wv + 32 * t + 8 * q + 4 * g);
#pragma unroll
                for (int i = 0; i < 4; ++i) y[t][4 * q + i] = y[t][4 * q + i] + bv[i] + x1[t][4 * q + i];
            }
        float mean, rstd;
        sl_row_stats(y, red, wv, r31, g, p.eps2, mean, rstd);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int f = 64 * wv + 32 * t + 8 * q + 4 * g;
                const half4 gm = *(const half4*)(p.g2 + f), bt = *(const half4*)(p.be2 + f);
                half4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (_Float16)((y[t][4 * q + i] - mean) * rstd * (float)gm[i] + (float)bt[i]);
                if (valid) *(half4*)(p.out + trow + f) = o;
                if (p.wqkv) *(half4*)(R0 + sl_tile_off(r31, f)) = o;
            }
    }
    if (!p.wqkv) return;  // (wave uniform, workgroup uniform)
    __syncthreads();      // R0 (x2) complete

    // ---- 4: the next layer's QKV projection: the wave owns features [192 wv, 192 wv + 192) in passes of 64 ----
    for (int f0 = 192 * wv; f0 < 192 * wv + 192; f0 += 64) {
        float16v z[2];
        sl_zero(z);
        const __half* wr = p.wqkv + (size_t)(f0 + r31) * ML_H + 8 * g;
        sl_pair_gemm<PF>(z, wr, wr + (size_t)32 * ML_H, ML_H / 16, R0 + r31 * 768, r31, g);
        if (!valid) continue;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int f = f0 + 32 * t + 8 * q + 4 * g;
                const float4v bv = *(const float4v*)(p.bqkv + f);
                half4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (_Float16)(z[t][4 * q + i] + bv[i]);
                *(half4*)(p.qkv + (int64_t)(tok0 + r31) * 1152 + f) = o;
            }
    }
}

}  // namespace lm

extern "C" int lm_small_layer_h384_f16(const void* d_attn, const void* d_resid, const void* d_wo, const float* d_bo, const void* d_gamma1, const void* d_beta1,
                                       float eps1, const void* d_w1, const float* d_b1, const void* d_w2, const float* d_b2, const void* d_gamma2,
                                       const void* d_beta2, float eps2, int32_t ffn, void* d_out, const void* d_wqkv_next, const float* d_bqkv_next,
                                       void* d_qkv_out, int64_t tokens, void* stream) {
    using namespace lm;
    if (tokens == 0) return LM_OK;
    if (!d_attn || !d_resid || !d_wo || !d_bo || !d_gamma1 || !d_beta1 || !d_w1 || !d_b1 || !d_w2 || !d_b2 || !d_gamma2 || !d_beta2 || !d_out || tokens < 0 ||
        tokens > 0x7fffffff)
        LM_FAIL(LM_EINVAL, "bad small-layer arguments");
    if ((d_wqkv_next != nullptr) != (d_bqkv_next != nullptr) || (d_wqkv_next != nullptr) != (d_qkv_out != nullptr))
        LM_FAIL(LM_EINVAL, "lm_small_layer_h384_f16: the next layer's QKV weight, bias and output go together");
    if (ffn < ML_H || ffn % ML_H || ffn > 4 * ML_H) LM_FAIL(LM_EINVAL, "lm_small_layer_h384_f16: ffn must be a multiple of 384 in [384, 1536]");
    SlArgs a{(const __half*)d_attn, (const __half*)d_resid, (const __half*)d_wo, d_bo, (const __half*)d_gamma1, (const __half*)d_beta1, (const __half*)d_w1, d_b1,
             (const __half*)d_w2, d_b2, (const __half*)d_gamma2, (const __half*)d_beta2, (const __half*)d_wqkv_next, d_bqkv_next, (__half*)d_out, (__half*)d_qkv_out,
             (int)tokens, (int)ffn, eps1, eps2};
    const size_t shmem = (size_t)(2 + ffn / ML_H) * T4_SLAB + 6 * 32 * 2 * sizeof(float);
    static DynLdsAttr attr;
    LM_HIP(ensure_dyn_lds(attr, (const void*)k_small_layer_h384<SL_PF>, shmem));
    hipLaunchKernelGGL(k_small_layer_h384<SL_PF>, dim3((unsigned)((tokens + 31) / 32)), dim3(SL_THREADS), shmem, (hipStream_t)stream, a);
    LM_HIP(hipGetLastError());
    return LM_OK;
}
