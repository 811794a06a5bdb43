// lm_rowgemm_ln_h384.hip -- ROW-COMPLETE linear layer with 384 outputs + residual + LayerNorm, for SMALL forwards:
//
//     out[t] = LayerNorm( resid[t] + x[t] W^T + b ) * gamma + beta          x [T][K] fp16, W [384][K] fp16 (nn.Linear layout), out [T][384] fp16
//
// Why: a one-query search round recomputes ~10 chunks (1-2 k tokens) and is a chain of ~50 dependent launches of ~10 us each
// (DESIGN.md section 8, item 6).  The small-forward form of a hidden-384 layer runs its two 384-output products -- attention output
// projection (K = 384) and fc2 (K = ffn) -- on the general GEMM (lm_gemm_f16: 128 x 128 tiles, K in tiles of 64 with a barrier each: a
// 24-step dependency chain at K = 1536 however few tokens there are) and each is followed by an lm_add_layernorm_f16 launch, because
// a 128-wide tile does not hold a whole row.  Here a workgroup owns 32 tokens x ALL 384 features: six waves x 64 features (two 32 x 32
// MFMA tiles each, out^T = W x^T as in the other hidden-384 kernels), the 32 x K token tile staged ONCE in LDS (coalesced 1 KB LDS-DMA
// pieces, XOR-swizzled: conflict-free B-fragment reads), W fragments straight from L2 into registers (no reuse inside a workgroup to stage
// them for), PF k-steps of them requested one chunk ahead; the row statistics meet in LDS (one pass: sum and sum of squares), and the
// normalised row leaves as fp16.  One launch instead of two, a K/16-step MFMA chain instead of K/64 barrier-separated tiles.
// May run IN PLACE on the residual (out == resid): a lane reads exactly the elements it writes.
//
// STATUS: written in round 4 after the GPU budget was spent -- validated in thread-per-lane emulation (tests/emulated_search_cases.py:
// rowgemm_ln) against numpy, NOT yet run or timed on an MI355X.  Off by default: LEANN_MI355X_SMALL_ROWLN=1 switches the small-forward
// form of lm_bert_h384_forward_packed / leann_amd/encoder.py onto it (scripts/next_gpu_session.sh measures it).
// Role in the reference: part of compute_embeddings' BERT forward (leann/embedding_compute.py:229-239).
#include "lm_h384_stream.h"

namespace lm {

constexpr int RG_THREADS = 384;  // six waves
constexpr int RG_PF = 8;         // k-steps of W fragments per chunk (two chunks live: 2 x 8 x 2 tiles x 4 registers = 128 VGPRs)

template <int PF>
__global__ __launch_bounds__(RG_THREADS) void k_rowgemm_ln_h384(const __half* __restrict__ x, const __half* __restrict__ w, const float* __restrict__ bias,
                                                                const __half* resid, const __half* __restrict__ gamma, const __half* __restrict__ beta,
                                                                __half* out, int T, int K, float eps) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, r31 = lane & 31, g = lane >> 5;
    const int tok0 = (int)blockIdx.x * 32;
    const int rows_valid = T - tok0 < 32 ? T - tok0 : 32;
    const int nblk = K / ML_H;        // 384-wide column blocks of the token tile: one 24 KB LDS block each
    const int nks = K >> 4;           // k-steps of 16
    float* red = (float*)(smem + (size_t)nblk * T4_SLAB);  // [6 waves][32 tokens][2]

    // ---- the token tile -> LDS: block b, piece p (64 lanes x 16 B): position (row, pos) of the block receives source chunk
    //      (pos & ~15) | ((pos ^ row) & 15) of row `row` (rows past the end repeat the last valid one); pieces dealt out over the six waves ----
    const unsigned char* xrows = (const unsigned char*)x + (size_t)tok0 * K * 2;
    for (int p = wv; p < nblk * 24; p += 6) {
        const int b = p / 24, pp = p - 24 * b;
        const int L = 64 * pp + lane, row = L / 48, pos = L - 48 * row;
        const int rc = row < rows_valid ? row : rows_valid - 1;
        lm_dma16_sv(xrows, (unsigned)(rc * K * 2 + b * 768 + (((pos & ~15) | ((pos ^ row) & 15)) << 4)), smem + (size_t)b * T4_SLAB + 1024 * pp);
    }

    // ---- accumulators start from the bias: lane (token r31, k-group g), tile t, register r = 4q + i  <->  feature 64 wv + 32 t + 8 q + 4 g + i ----
    float16v acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4v bv = *(const float4v*)(bias + 64 * wv + 32 * t + 8 * q + 4 * g);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[t][4 * q + i] = bv[i];
        }
    // W fragments: tile t, lane (feature row 64 wv + 32 t + r31, k-group g): eight consecutive k of that row per k-step
    const __half* wr0 = w + (size_t)(64 * wv + r31) * K + 8 * g;
    const __half* wr1 = wr0 + (size_t)32 * K;
    const int nchunk = nks / PF;  // K is a multiple of 384 = 24 k-steps = 3 chunks of 8
    half8 wa[2][PF][2];
#pragma unroll
    for (int j = 0; j < PF; ++j) {
        wa[0][j][0] = *(const half8*)(wr0 + 16 * j);
        wa[0][j][1] = *(const half8*)(wr1 + 16 * j);
    }
    T4_WAIT_VM(0);
    __syncthreads();  // every wave's pieces of the token tile have landed

    const unsigned char* xb = smem + r31 * 768;
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

    // ---- + residual, row statistics (one pass), LayerNorm ----
    const bool valid = r31 < rows_valid;
    const int64_t trow = (int64_t)(tok0 + (valid ? r31 : rows_valid - 1)) * ML_H;
    float s = 0.f, s2 = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f = 64 * wv + 32 * t + 8 * q + 4 * g;
            half4 rv = {0, 0, 0, 0};
            if (resid) rv = *(const half4*)(resid + trow + f);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = acc[t][4 * q + i] + (float)rv[i];
                acc[t][4 * q + i] = v;
                s += v;
                s2 += v * v;
            }
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
    const float mean = ts * (1.0f / ML_H);
    float var = ts2 * (1.0f / ML_H) - mean * mean;
    var = var > 0.f ? var : 0.f;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (!valid) return;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f = 64 * wv + 32 * t + 8 * q + 4 * g;
            const half4 gm = *(const half4*)(gamma + f), bt = *(const half4*)(beta + f);
            half4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (_Float16)((acc[t][4 * q + i] - mean) * rstd * (float)gm[i] + (float)bt[i]);
            *(half4*)(out + trow + f) = o;
        }
}

}  // namespace lm

extern "C" int lm_rowgemm_ln_h384_f16(const void* d_x, const void* d_w, const float* d_bias, int32_t k_in, const void* d_residual, const void* d_gamma,
                                      const void* d_beta, float eps, void* d_out, int64_t tokens, void* stream) {
    using namespace lm;
    if (tokens == 0) return LM_OK;
    if (!d_x || !d_w || !d_bias || !d_gamma || !d_beta || !d_out || tokens < 0 || tokens > 0x7fffffff) LM_FAIL(LM_EINVAL, "bad linear arguments");
    if (k_in < ML_H || k_in % ML_H || k_in > 6 * ML_H) LM_FAIL(LM_EINVAL, "lm_rowgemm_ln_h384_f16: k_in must be a multiple of 384 in [384, 2304]");
    const size_t shmem = (size_t)(k_in / ML_H) * T4_SLAB + 6 * 32 * 2 * sizeof(float);
    static DynLdsAttr attr;
    LM_HIP(ensure_dyn_lds(attr, (const void*)k_rowgemm_ln_h384<RG_PF>, shmem));
    hipLaunchKernelGGL(k_rowgemm_ln_h384<RG_PF>, dim3((unsigned)((tokens + 31) / 32)), dim3(RG_THREADS), shmem, (hipStream_t)stream, (const __half*)d_x,
                       (const __half*)d_w, d_bias, (const __half*)d_residual, (const __half*)d_gamma, (const __half*)d_beta, (__half*)d_out, (int)tokens,
                       (int)k_in, eps);
    LM_HIP(hipGetLastError());
    return LM_OK;
}
