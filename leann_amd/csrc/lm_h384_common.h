// lm_h384_common.h -- pieces shared by the hidden-384 MFMA kernels (lm_mlp_fused.hip, lm_linear_h384.hip):
// vector typedefs, tile constants and the bias + residual + LayerNorm epilogue on the transposed accumulator layout.
#pragma once
#include <hip/hip_fp16.h>

#include "lm_internal.h"

namespace lm {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int ML_H = 384;                 // hidden size
constexpr int ML_KS = ML_H / 16;          // 24 k-steps of the first product
constexpr int ML_NJ = ML_H / 32;          // 12 row tiles of out^T

// epilogue: + b2 + residual, LayerNorm over the 384 features of the token (lane pair
// r31 / r31+32); lane (token r31, g), tile j, register r = 4q + i  <->  feature 32j + 8q + 4g + i
__device__ inline void mlp_epilogue(float16v (&o)[ML_NJ], const __half* __restrict__ x, const float* __restrict__ b2,
                                    const __half* __restrict__ gamma, const __half* __restrict__ beta, __half* __restrict__ out,
                                    int token, bool valid, int g, float eps) {
    // keep the epilogue's loads below the main loop: in a fully unrolled kernel (one basic block) the scheduler would
    // otherwise hoist all ~200 of them to the top and spill them (868 B of scratch per lane in k_linear_h384<1>)
    __builtin_amdgcn_sched_barrier(0);
    const _Float16* xres = (const _Float16*)x + (int64_t)(valid ? token : 0) * ML_H + 4 * g;
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < ML_NJ; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f0 = 32 * j + 8 * q;
            half4 xr = *(const half4*)(xres + f0);
            float4v bb = *(const float4v*)(b2 + f0 + 4 * g);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = o[j][4 * q + i] + ((float)xr[i] + bb[i]);
                o[j][4 * q + i] = v;
                sum += v;
            }
        }
    sum += __shfl_xor(sum, 32);
    const float mean = sum * (1.0f / ML_H);
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < ML_NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float d = o[j][r] - mean;
            sq += d * d;
        }
    sq += __shfl_xor(sq, 32);
    const float rstd = rsqrtf(sq * (1.0f / ML_H) + eps);
    if (valid) {
        _Float16* yr = (_Float16*)out + (int64_t)token * ML_H + 4 * g;
        const _Float16* gm = (const _Float16*)gamma + 4 * g;
        const _Float16* bt = (const _Float16*)beta + 4 * g;
#pragma unroll
        for (int j = 0; j < ML_NJ; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int f0 = 32 * j + 8 * q;
                half4 gv = *(const half4*)(gm + f0), bv = *(const half4*)(bt + f0), y;
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i] = (_Float16)((o[j][4 * q + i] - mean) * rstd * (float)gv[i] + (float)bv[i]);
                *(half4*)(yr + f0) = y;
            }
    }
}

}  // namespace lm
