// lm_encoder_ops2.hip -- second batch of hand-written encoder kernels around the GEMMs.
//
// STATUS: default since round 2 (each one agreed with the first-generation path to 3e-5 and was faster in the round-1
// driver run on an MI355X); the switches select the first-generation path for A/B (tests/test_gpu_encoder_kernels.py):
//   LEANN_MI355X_LN=1      lm_add_layernorm_f16 -> one wave per row instead of k_add_layernorm_f16_r16 (16 lanes per row)
//   LEANN_MI355X_POOL=0    mean pooling through torch index_add_ instead of lm_meanpool_varlen_f16
//   LEANN_MI355X_EMBED=0   torch embedding gathers + lm_add_layernorm_f16 instead of lm_embed_layernorm_f16
//   LEANN_MI355X_PACK=0    boolean-mask packing instead of lm_pack_tokens
//
// Why (rocprofv3 of the default bench, profiles/r1_final_bench_default_kernel_stats.csv):
//   * k_add_layernorm_f16<1> runs at ~3 TB/s for hidden 384: one wave per row keeps only 48 of 64 lanes busy and
//     2 x 16 B loads in flight per lane.  16 lanes per row: every lane busy, 6 loads in flight (9 % of the forward);
//   * mean pooling is index_add_ with fp32 atomics + an fp16->fp32 copy of the whole activation (0.9 ms per
//     forward, 3.2 %); packed sequences are contiguous, so a segmented sum needs neither;
//   * the embedding front end is three gathers/adds plus the LayerNorm kernel (~0.5 ms per forward).
// Role in the reference: parts of compute_embeddings' BERT forward and mean pooling
// (leann/embedding_compute.py:229-239, 323-334).
#include <cstdlib>

#include <hip/hip_fp16.h>

#include "lm_internal.h"

namespace lm {

__device__ inline void unpack8(const uint4& a, float* f) {
    const __half2* h = (const __half2*)&a;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float2 t = __half22float2(h[j]);
        f[2 * j] = t.x;
        f[2 * j + 1] = t.y;
    }
}

// shared tail of the two LayerNorm kernels: v[NV][8] holds the lane's slice of the row (zero where c >= nvec);
// gamma / beta slices were loaded up front (gv, bv) so that no global load sits behind the reductions
template <int NV, bool EXACT>
__device__ inline void ln16_finish(float (&v)[NV][8], const uint4 (&gv)[NV], const uint4 (&bv)[NV], int lane16, int nvec, int H,
                                   float eps, __half* __restrict__ orow) {
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += v[i][j];
    for (int m = 8; m >= 1; m >>= 1) sum += __shfl_xor(sum, m, 16);
    const float mean = sum / (float)H;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float w = (EXACT || lane16 + 16 * i < nvec) ? 1.0f : 0.0f;  // select, not a branch
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float d = (v[i][j] - mean) * w;
            sq += d * d;
        }
    }
    for (int m = 8; m >= 1; m >>= 1) sq += __shfl_xor(sq, m, 16);
    const float rstd = rsqrtf(sq / (float)H + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane16 + 16 * i;
        float fg[8], fb[8];
        unpack8(gv[i], fg);
        unpack8(bv[i], fb);
        uint4 o;
        __half2* oh = (__half2*)&o;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            oh[j] = __floats2half2_rn((v[i][2 * j] - mean) * rstd * fg[2 * j] + fb[2 * j],
                                      (v[i][2 * j + 1] - mean) * rstd * fg[2 * j + 1] + fb[2 * j + 1]);
        if (EXACT || c < nvec) ((uint4*)orow)[c] = o;
    }
}

__device__ inline uint4 zero_if(uint4 a, bool z) {
    a.x = z ? 0u : a.x;
    a.y = z ? 0u : a.y;
    a.z = z ? 0u : a.z;
    a.w = z ? 0u : a.w;
    return a;
}

// out = LayerNorm(x + residual) * gamma + beta; 16 lanes per row, 16 rows per 256-thread workgroup.
// Every global load of a lane (x, residual, gamma, beta slices) is issued before the first use: out-of-range
// columns read a clamped (valid) address and are zeroed with selects, so there is no branch -- and no
// s_waitcnt -- between the loads.
template <int NV, bool HAS_RES, bool EXACT>  // NV = ceil(H / 128): 16-byte vectors per lane; EXACT: H == 128 * NV (straight-line code)
__global__ __launch_bounds__(256) void k_add_layernorm_f16_r16(const __half* __restrict__ x, const __half* __restrict__ res,
                                                               const __half* __restrict__ gamma, const __half* __restrict__ beta,
                                                               __half* __restrict__ out, int64_t rows, int H, float eps) {
    const int lane16 = threadIdx.x & 15;
    const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (row >= rows) return;  // a 16-lane group leaves together; the shuffles below stay inside the group
    const int nvec = H >> 3;
    uint4 a[NV], b[NV], gv[NV], bv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = EXACT ? lane16 + 16 * i : min(lane16 + 16 * i, nvec - 1);
        a[i] = ((const uint4*)(x + row * H))[c];
        if (HAS_RES) b[i] = ((const uint4*)(res + row * H))[c];
        gv[i] = ((const uint4*)gamma)[c];
        bv[i] = ((const uint4*)beta)[c];
    }
    float v[NV][8];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const bool oob = !EXACT && lane16 + 16 * i >= nvec;
        float fa[8], fb[8];
        unpack8(zero_if(a[i], oob), fa);
        if (HAS_RES) unpack8(zero_if(b[i], oob), fb);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i][j] = HAS_RES ? fa[j] + fb[j] : fa[j];
    }
    ln16_finish<NV, EXACT>(v, gv, bv, lane16, nvec, H, eps, out + row * H);
}

// out[row] = LayerNorm(half(word[tok[row]] + type0) + pos_table[pos[row]]) -- the embedding front end in one pass
template <int NV, bool EXACT>
__global__ __launch_bounds__(256) void k_embed_layernorm_f16(const int32_t* __restrict__ tok, const int32_t* __restrict__ pos,
                                                             const __half* __restrict__ word, const __half* __restrict__ posw,
                                                             const __half* __restrict__ type0, const __half* __restrict__ gamma,
                                                             const __half* __restrict__ beta, __half* __restrict__ out, int64_t rows,
                                                             int H, float eps) {
    const int lane16 = threadIdx.x & 15;
    const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (row >= rows) return;
    const int nvec = H >> 3;
    const int64_t wrow = tok[row], prow = pos[row];
    uint4 a[NV], b[NV], t[NV], gv[NV], bv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = EXACT ? lane16 + 16 * i : min(lane16 + 16 * i, nvec - 1);
        a[i] = ((const uint4*)(word + wrow * H))[c];
        b[i] = ((const uint4*)(posw + prow * H))[c];
        t[i] = ((const uint4*)type0)[c];
        gv[i] = ((const uint4*)gamma)[c];
        bv[i] = ((const uint4*)beta)[c];
    }
    float v[NV][8];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const bool oob = !EXACT && lane16 + 16 * i >= nvec;
        float fa[8], fb[8], ft[8];
        unpack8(zero_if(a[i], oob), fa);
        unpack8(zero_if(b[i], oob), fb);
        unpack8(zero_if(t[i], oob), ft);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i][j] = __half2float(__float2half_rn(fa[j] + ft[j])) + fb[j];  // (word + type) is an fp16 tensor upstream
    }
    ln16_finish<NV, EXACT>(v, gv, bv, lane16, nvec, H, eps, out + row * H);
}

// mean (and optional L2 normalisation) over the tokens of each packed sequence; one workgroup per sequence.
// thread = (row group r0, 16-byte column c): sums rows r0, r0+RG, ... in fp32 in a fixed order (deterministic,
// unlike atomics), row groups are combined through LDS.
__global__ __launch_bounds__(256) void k_meanpool_varlen_f16(const __half* __restrict__ x, const int32_t* __restrict__ cu,
                                                             float* __restrict__ out, int H, int normalize, int first_only = 0) {
    __shared__ float red[256 * 8];
    __shared__ float wsum[4];
    const int seq = blockIdx.x, tid = threadIdx.x;
    const int tok0 = cu[seq];
    // first_only: CLS pooling = the "mean" over the sequence's first token alone (x * 1.0f is exact)
    const int len = first_only ? min(cu[seq + 1] - tok0, 1) : cu[seq + 1] - tok0;
    const int nvec = H >> 3;      // <= 256
    const int rg = 256 / nvec;    // row groups
    const int c = tid % nvec, r0 = tid / nvec;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (r0 < rg) {
        const __half* base = x + (int64_t)tok0 * H;
#pragma unroll 4
        for (int row = r0; row < len; row += rg) {
            uint4 a = ((const uint4*)(base + (int64_t)row * H))[c];
            float f[8];
            unpack8(a, f);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += f[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[tid * 8 + j] = acc[j];
    __syncthreads();
    float m[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float ss = 0.f;
    if (tid < nvec) {
        for (int g = 0; g < rg; ++g)
#pragma unroll
            for (int j = 0; j < 8; ++j) m[j] += red[(g * nvec + tid) * 8 + j];
        const float inv = 1.0f / (float)(len > 0 ? len : 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            m[j] *= inv;
            ss += m[j] * m[j];
        }
    }
    float scale = 1.0f;
    if (normalize) {  // uniform branch
        for (int k = 32; k >= 1; k >>= 1) ss += __shfl_xor(ss, k);
        if ((tid & 63) == 0) wsum[tid >> 6] = ss;
        __syncthreads();
        const float tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        scale = 1.0f / fmaxf(sqrtf(tot), 1e-12f);  // torch.nn.functional.normalize: x / max(||x||, eps)
    }
    if (tid < nvec) {
        float4* o = (float4*)(out + (int64_t)seq * H + tid * 8);
        o[0] = make_float4(m[0] * scale, m[1] * scale, m[2] * scale, m[3] * scale);
        o[1] = make_float4(m[4] * scale, m[5] * scale, m[6] * scale, m[7] * scale);
    }
}

// padded [n][t] token ids + lengths + cumulative lengths -> packed token ids / positions (one wave per chunk).
// Replaces the boolean-mask selects of the packing front end (three nonzero/gather passes and their host syncs).
__global__ __launch_bounds__(256) void k_pack_tokens(const int32_t* __restrict__ ids, const int32_t* __restrict__ lens,
                                                     const int32_t* __restrict__ cu, int32_t n, int32_t t, int32_t* __restrict__ tok,
                                                     int32_t* __restrict__ pos) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= n) return;
    const int len = min(lens[w], t), base = cu[w];
    const int32_t* row = ids + (int64_t)w * t;
    for (int j = lane; j < len; j += 64) {
        tok[base + j] = row[j];
        pos[base + j] = j;
    }
}

}  // namespace lm

#ifndef LM_HOST_EMULATION
// lm_add_layernorm_f16 dispatches here when LEANN_MI355X_LN=2 and hidden <= 768
int lm_add_layernorm_r16_launch(const void* d_x, const void* d_residual, const void* d_gamma, const void* d_beta, void* d_out,
                                int64_t rows, int32_t hidden, float eps, void* stream) {
    using namespace lm;
    dim3 grid((unsigned)((rows + 15) / 16)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const __half *x = (const __half*)d_x, *r = (const __half*)d_residual, *g = (const __half*)d_gamma, *b = (const __half*)d_beta;
    __half* o = (__half*)d_out;
    const bool ex = hidden % 128 == 0;
    switch ((hidden + 127) / 128) {
#define CASEL(n)                                                                                                       \
    case n:                                                                                                            \
        if (r && ex) hipLaunchKernelGGL((k_add_layernorm_f16_r16<n, true, true>), grid, block, 0, st, x, r, g, b, o, rows, hidden, eps);        \
        else if (r) hipLaunchKernelGGL((k_add_layernorm_f16_r16<n, true, false>), grid, block, 0, st, x, r, g, b, o, rows, hidden, eps);        \
        else if (ex) hipLaunchKernelGGL((k_add_layernorm_f16_r16<n, false, true>), grid, block, 0, st, x, r, g, b, o, rows, hidden, eps);       \
        else hipLaunchKernelGGL((k_add_layernorm_f16_r16<n, false, false>), grid, block, 0, st, x, r, g, b, o, rows, hidden, eps);              \
        break
        CASEL(1); CASEL(2); CASEL(3); CASEL(4); CASEL(5); CASEL(6);
#undef CASEL
        default: LM_FAIL(LM_EINVAL, "16-lane LayerNorm supports hidden <= 768");
    }
    LM_HIP(hipGetLastError());
    return LM_OK;
}

extern "C" int lm_embed_layernorm_f16(const int32_t* d_tok, const int32_t* d_pos, const void* d_word, const void* d_pos_table,
                                      const void* d_type0, const void* d_gamma, const void* d_beta, void* d_out, int64_t rows,
                                      int32_t hidden, float eps, void* stream) {
    using namespace lm;
    if (rows == 0) return LM_OK;
    if (!d_tok || !d_pos || !d_word || !d_pos_table || !d_type0 || !d_gamma || !d_beta || !d_out || rows < 0)
        LM_FAIL(LM_EINVAL, "bad embed_layernorm arguments");
    if (hidden <= 0 || hidden % 8 || hidden > 768) LM_FAIL(LM_EINVAL, "hidden must be a multiple of 8, <= 768");
    dim3 grid((unsigned)((rows + 15) / 16)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const __half *w = (const __half*)d_word, *p = (const __half*)d_pos_table, *t = (const __half*)d_type0;
    const __half *g = (const __half*)d_gamma, *b = (const __half*)d_beta;
    __half* o = (__half*)d_out;
    const bool ex = hidden % 128 == 0;
    switch ((hidden + 127) / 128) {
#define CASEE(n)                                                                                                                       \
    case n:                                                                                                                            \
        if (ex) hipLaunchKernelGGL((k_embed_layernorm_f16<n, true>), grid, block, 0, st, d_tok, d_pos, w, p, t, g, b, o, rows, hidden, eps);  \
        else hipLaunchKernelGGL((k_embed_layernorm_f16<n, false>), grid, block, 0, st, d_tok, d_pos, w, p, t, g, b, o, rows, hidden, eps);    \
        break
        CASEE(1); CASEE(2); CASEE(3); CASEE(4); CASEE(5); CASEE(6);
#undef CASEE
    }
    LM_HIP(hipGetLastError());
    return LM_OK;
}

extern "C" int lm_pack_tokens(const int32_t* d_ids, const int32_t* d_lens, const int32_t* d_cu_seqlens, int32_t n, int32_t t,
                              int32_t* d_tok, int32_t* d_pos, void* stream) {
    using namespace lm;
    if (n == 0) return LM_OK;
    if (!d_ids || !d_lens || !d_cu_seqlens || !d_tok || !d_pos || n < 0 || t <= 0) LM_FAIL(LM_EINVAL, "bad pack_tokens arguments");
    hipLaunchKernelGGL(k_pack_tokens, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, d_ids, d_lens, d_cu_seqlens, n, t,
                       d_tok, d_pos);
    LM_HIP(hipGetLastError());
    return LM_OK;
}

extern "C" int lm_meanpool_varlen_f16(const void* d_x, const int32_t* d_cu_seqlens, int32_t n_seqs, int32_t hidden, int32_t normalize,
                                      float* d_out, void* stream) {
    using namespace lm;
    if (n_seqs == 0) return LM_OK;
    if (!d_x || !d_cu_seqlens || !d_out || n_seqs < 0) LM_FAIL(LM_EINVAL, "bad meanpool arguments");
    if (hidden <= 0 || hidden % 8 || hidden > 2048) LM_FAIL(LM_EINVAL, "hidden must be a multiple of 8, <= 2048");
    hipLaunchKernelGGL(k_meanpool_varlen_f16, dim3((unsigned)n_seqs), dim3(256), 0, (hipStream_t)stream, (const __half*)d_x,
                       d_cu_seqlens, d_out, hidden, normalize, 0);
    LM_HIP(hipGetLastError());
    return LM_OK;
}

// CLS pooling (+ optional L2 normalisation): d_out[s] = float(x[first token of sequence s]), the same kernel restricted to one row
extern "C" int lm_clspool_varlen_f16(const void* d_x, const int32_t* d_cu_seqlens, int32_t n_seqs, int32_t hidden, int32_t normalize,
                                     float* d_out, void* stream) {
    using namespace lm;
    if (n_seqs == 0) return LM_OK;
    if (!d_x || !d_cu_seqlens || !d_out || n_seqs < 0) LM_FAIL(LM_EINVAL, "bad clspool arguments");
    if (hidden <= 0 || hidden % 8 || hidden > 2048) LM_FAIL(LM_EINVAL, "hidden must be a multiple of 8, <= 2048");
    hipLaunchKernelGGL(k_meanpool_varlen_f16, dim3((unsigned)n_seqs), dim3(256), 0, (hipStream_t)stream, (const __half*)d_x,
                       d_cu_seqlens, d_out, hidden, normalize, 1);
    LM_HIP(hipGetLastError());
    return LM_OK;
}
#endif  // LM_HOST_EMULATION
