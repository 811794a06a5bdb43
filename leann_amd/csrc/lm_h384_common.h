// lm_h384_common.h -- pieces shared by the hidden-384 MFMA kernels (lm_layer_tail_h384.hip, lm_qkv_h384.hip, lm_gemm_ws_h384.hip):
// vector typedefs, tile constants and the bias + residual + LayerNorm epilogue on the transposed accumulator layout.
#pragma once
#include <hip/hip_fp16.h>

#include <cstring>

#include "lm_internal.h"

namespace lm {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// v_permlane32_swap_b32: lanes 32..63 of `a` trade places with lanes 0..31 of `b` (one instruction, no LDS).  With a = the
// value a lane pair (l, l ^ 32) wants to keep in its lower half and b = the one it wants to keep in its upper half, the pair
// ends up with (a_lo, b_lo) in lane l and (a_hi, b_hi) in lane l ^ 32.
__device__ __forceinline__ void lane32_swap(uint32_t& a, uint32_t& b) {
#ifdef LM_EMULATED_DEVICE
    const bool hi = (threadIdx.x & 63) >= 32;
    const uint32_t pa = __shfl_xor(a, 32), pb = __shfl_xor(b, 32);
    const uint32_t na = hi ? pb : a, nb = hi ? b : pa;
    a = na;
    b = nb;
#else
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
#endif
}

// One LDS-DMA piece: 64 lanes x 16 B, lane l from gsrc (its own address) to LDS byte lds_wave_base (wave uniform) + 16 l.
// Inline assembly ON PURPOSE.  The compiler models global_load_lds (the __builtin_amdgcn_global_load_lds form) as a FLAT access
// that may touch LDS, and from then on every wait it inserts for an LDS read in the same loop is s_waitcnt lgkmcnt(0) -- also for
// a fragment read four MFMAs ago, i.e. right behind the NEXT fragment read: every group of four MFMAs paid a full LDS round
// trip (s_memtime stamps of the fused MLP with the builtin: 3200 cycles per 48-MFMA iteration against a 1536-cycle matrix-pipe
// floor; the same loop without a DMA gets counted lgkmcnt(3) waits).  The asm form is invisible to that bookkeeping; what orders
// the pieces is each kernel's own counted s_waitcnt vmcnt + barrier protocol, exactly as before.
#ifdef LM_EMULATED_DEVICE
__device__ inline void lm_dma16(const void* gsrc, unsigned char* lds_wave_base) { std::memcpy(lds_wave_base + 16 * (threadIdx.x & 63), gsrc, 16); }
__device__ inline void lm_dma16_sv(const void* sbase, unsigned voff, unsigned char* lds_wave_base) { lm_dma16((const unsigned char*)sbase + voff, lds_wave_base); }
__device__ inline void lm_dma16_sv_nt(const void* sbase, unsigned voff, unsigned char* lds_wave_base) { lm_dma16((const unsigned char*)sbase + voff, lds_wave_base); }
#else
// Wait states between the M0 write and the first LDS-DMA instruction of an asm block, as the s_nop operand (s_nop N = N + 1 wait states).
// TWO rules of the gfx9 ISA meet here: (a) SALU writes M0 -> an instruction that uses M0 as its LDS address: 1 wait state (s_nop 0 would do);
// (b) a VALU instruction wrote an SGPR (v_readfirstlane / v_readlane: how the compiler makes a wave-uniform base or reloads a spilled one)
// -> a VMEM instruction reads that SGPR: 5 wait states.  The compiler inserts (b) for memory instructions it knows and cannot for the inside
// of an asm block, so EVERY block whose VMEM instruction takes an "s" operand carries s_mov (1) + s_nop 3 (4) = 5 -- decided by the rule, not
// by what a build happens to emit (round 5 had s_nop 3 in lm_dma16_sv only).  Cost, measured in round 6's first GPU session with the diagnosis
// build at LM_DMA_NOP=0 against 3, interleaved: within run-to-run noise on the layer tail and the QKV kernel.
// Every block that writes M0 says so in its clobber list ("m0" is a reserved register to this compiler, which re-materialises M0 in front of each of
// its own uses but may MERGE equal initialisations across a region -- an undeclared write in between would go unnoticed; the clobber costs a
// -Winline-asm note, silenced here; tests/test_asm_hazards.py::test_m0_writes_are_declared keeps the lists honest).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
#ifndef LM_DMA_NOP
#define LM_DMA_NOP 3
#endif
#define LM_STR2(x) #x
#define LM_STR(x) LM_STR2(x)
#define LM_DMA_NOPS "s_nop " LM_STR(LM_DMA_NOP) "\n\t"
__device__ __forceinline__ void lm_dma16(const void* gsrc, unsigned char* lds_wave_base) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_wave_base);  // low half of the flat address = LDS offset
    // (the address is a VGPR pair: only rule (a) applies)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(m0v) : "memory", "m0");
}
// The same piece with the source as (wave-uniform 64-bit base in SGPRs) + (per-lane 32-bit byte offset): no 64-bit VALU add per
// piece; with a wave-uniform lds_wave_base the M0 value is SALU arithmetic as well (the readfirstlane folds away).
// LM_DMA_NOPS (s_nop 3), not s_nop 0: rule (b) above -- the compiler may hand over `sbase` in SGPRs it has just written with a VALU instruction
// (k_gemm_f16 reloads spilled SGPRs with v_readlane right in front of its pieces).  No wrong result was ever seen with s_nop 0 -- every GEMM
// test compares bits -- but the ISA does not promise it; tests/test_asm_hazards.py checks the assembly of every kernel for the pattern.
__device__ __forceinline__ void lm_dma16_sv(const void* sbase, unsigned voff, unsigned char* lds_wave_base) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_wave_base);
    asm volatile("s_mov_b32 m0, %2\n\t" LM_DMA_NOPS "global_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(m0v) : "memory", "m0");
}
// ... with the non-temporal hint (streamed activations: read once; A/B switch LM_T4_NT of the layer tail)
__device__ __forceinline__ void lm_dma16_sv_nt(const void* sbase, unsigned voff, unsigned char* lds_wave_base) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_wave_base);
    asm volatile("s_mov_b32 m0, %2\n\t" LM_DMA_NOPS "global_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(sbase), "s"(m0v) : "memory", "m0");
}
#pragma clang diagnostic pop
#endif

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
    // packed fp32 pairs and two independent partial sums per statistic: 192 values per lane would otherwise be two
    // 192-long dependent v_add_f32 chains
    float2v sa = {0.f, 0.f}, sb = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < ML_NJ; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f0 = 32 * j + 8 * q;
            half4 xr = *(const half4*)(xres + f0);
            float4v bb = *(const float4v*)(b2 + f0 + 4 * g);
            float2v v0 = (float2v){o[j][4 * q], o[j][4 * q + 1]} + ((float2v){(float)xr[0], (float)xr[1]} + (float2v){bb[0], bb[1]});
            float2v v1 = (float2v){o[j][4 * q + 2], o[j][4 * q + 3]} + ((float2v){(float)xr[2], (float)xr[3]} + (float2v){bb[2], bb[3]});
            o[j][4 * q] = v0[0];
            o[j][4 * q + 1] = v0[1];
            o[j][4 * q + 2] = v1[0];
            o[j][4 * q + 3] = v1[1];
            sa += v0;
            sb += v1;
        }
    float sum = (sa[0] + sa[1]) + (sb[0] + sb[1]);
    sum += __shfl_xor(sum, 32);
    const float mean = sum * (1.0f / ML_H);
    const float2v nm = {-mean, -mean};
    float2v qa = {0.f, 0.f}, qb = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < ML_NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; r += 4) {
            float2v d0 = (float2v){o[j][r], o[j][r + 1]} + nm, d1 = (float2v){o[j][r + 2], o[j][r + 3]} + nm;
            qa = __builtin_elementwise_fma(d0, d0, qa);
            qb = __builtin_elementwise_fma(d1, d1, qb);
        }
    float sq = (qa[0] + qa[1]) + (qb[0] + qb[1]);
    sq += __shfl_xor(sq, 32);
    const float rstd = rsqrtf(sq * (1.0f / ML_H) + eps);
    const float2v rs = {rstd, rstd};
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
                float2v n0 = ((float2v){o[j][4 * q], o[j][4 * q + 1]} + nm) * rs;
                float2v n1 = ((float2v){o[j][4 * q + 2], o[j][4 * q + 3]} + nm) * rs;
                float2v y0 = __builtin_elementwise_fma(n0, (float2v){(float)gv[0], (float)gv[1]}, (float2v){(float)bv[0], (float)bv[1]});
                float2v y1 = __builtin_elementwise_fma(n1, (float2v){(float)gv[2], (float)gv[3]}, (float2v){(float)bv[2], (float)bv[3]});
                y[0] = (_Float16)y0[0];
                y[1] = (_Float16)y0[1];
                y[2] = (_Float16)y1[0];
                y[3] = (_Float16)y1[1];
                *(half4*)(yr + f0) = y;
            }
    }
}

}  // namespace lm
