// lm_encoder_ops.hip -- the general-width residual + LayerNorm kernel of the encoder forward (one wave per row; hidden <= 2048).
// Hidden sizes up to 768 -- every model BASELINE.json names -- take the 16-lanes-per-row kernel of lm_encoder_ops2.hip instead
// (lm_add_layernorm_f16 below dispatches); this one serves wider rows and LEANN_MI355X_LN=1 (A/B).
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
