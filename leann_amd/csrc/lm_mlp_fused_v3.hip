// lm_mlp_fused_v3.hip -- third variant of the fused feed-forward block (hidden 384):
//
//     y = LayerNorm( x + GELU(x W1^T + b1) W2^T + b2 ) * gamma + beta          x, y: [T, 384] fp16
//
// Same mathematics, operand layouts, weight packing and epilogue as lm_mlp_fused.hip (one 256-thread workgroup = 4 waves =
// 128 tokens, everything transposed, the 1536-wide intermediate lives in MFMA accumulators).  What the PMC counters of
// variants 1 / 2 on the MI355X showed (profiles/r2_pmc_encoder_kernels_131k_tokens.txt, 131k tokens):
//     matrix pipe busy 29 % of the wave cycles; 24 % parked at s_waitcnt / barriers, 33 % issue stalls;
//     71 M VALU instructions for 9.4 M MFMAs -- the exact-erf GELU is as much issue time as the two products, and it was
//     written with packed-fp32 operations, which cost ~22 extra cycles each when issued beside MFMAs
//     (MI355X_MICROARCH.md, "price of one filler beside MFMAs"); 13.6 % of the LDS cycles were bank conflicts;
//     weights went HBM/L2 -> VGPR -> LDS (48 registers, a vmcnt(0) and 48 ds_write_b128 per slab).
// This variant
//   * streams W1 / W2 slabs L2 -> LDS with global_load_lds_dwordx4 (no registers, no ds_write), three stages per matrix,
//     counted vmcnt + one raw s_barrier per slab; LDS images are lane-linear per DMA instruction, made bank-conflict
//     free by an XOR swizzle of the SOURCE chunk and of the fragment READ (same scheme as lm_gemm_h384.hip);
//   * evaluates GELU in SCALAR fp32 (no packed ops) as a stream of 256 micro-operations per slab, four values in flight so
//     that neighbouring instructions are independent, 5-6 of them behind EACH of the 48 MFMAs of an iteration; to have 48
//     gaps for them the pipeline is skewed by two slabs:
//         iteration s = { first product of slab s+1 | GELU of slab s | second product of slab s-1 };
//   * fragment reads run four MFMAs ahead, across the boundary between the two products.
// Role in the reference: the FFN inside compute_embeddings' BERT forward (leann/embedding_compute.py:229-239).
#include <cstdlib>
#include <cstring>
#include <utility>

#include "lm_h384_common.h"

namespace lm {

constexpr int M3_SLAB = 24576;                 // bytes of a W1 slab (32 hidden x 384 k) and of a W2 slab (384 rows x 32 hidden)
constexpr int M3_STAGES = 3;
constexpr int M3_W1_OFF = 0;
constexpr int M3_W2_OFF = M3_STAGES * M3_SLAB;  // 73728
constexpr int M3_B1_OFF = 2 * M3_STAGES * M3_SLAB;  // 147456: b1 as floats behind the six stages

#ifdef LM_EMULATED_DEVICE
__device__ inline void m3_dma16(const void* gsrc, unsigned char* lds_wave_base) { std::memcpy(lds_wave_base + 16 * (threadIdx.x & 63), gsrc, 16); }
#define M3_WAIT_VM(n) ((void)0)
#define M3_BARRIER() __syncthreads()
#else
__device__ __forceinline__ void m3_dma16(const void* gsrc, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
#define M3_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define M3_BARRIER() __builtin_amdgcn_s_barrier()
#endif

// W1 slab [32 rows][48 chunks] -> stage: LDS chunk L = 256 i + tid = (row = L / 48, pos = L % 48) holds source chunk
// (pos & ~15) | ((pos ^ row) & 15) of that row (rows are 768 B = 3 x 256 B apart: a 16-chunk XOR swizzle)
__device__ __forceinline__ void m3_w1_offsets(int tid, int (&off)[6]) {  // per-thread source offsets of the 6 pieces, computed once
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int L = 256 * i + tid, row = L / 48, pos = L - 48 * row;
        off[i] = row * 768 + (((pos & ~15) | ((pos ^ row) & 15)) << 4);
    }
}
__device__ __forceinline__ void m3_issue_w1(const unsigned char* slab, unsigned char* stage, int tid, const int (&off)[6]) {
    const int wv = tid >> 6;
#pragma unroll
    for (int i = 0; i < 6; ++i) m3_dma16(slab + off[i], stage + (256 * i + 64 * wv) * 16);
}
// W2 slab [384 rows][4 chunks] -> stage: LDS chunk L = (row = L >> 2, pos = L & 3) holds source chunk pos ^ ((row >> 2) & 3)
__device__ __forceinline__ void m3_issue_w2(const unsigned char* slab, unsigned char* stage, int tid) {
    const int wv = tid >> 6;
    const unsigned char* src = slab + (tid >> 2) * 64 + (((tid & 3) ^ ((tid >> 4) & 3)) << 4);
#pragma unroll
    for (int i = 0; i < 6; ++i) m3_dma16(src + i * 4096, stage + (256 * i + 64 * wv) * 16);
}

// exact (erf) GELU in scalar fp32 (Abramowitz-Stegun 7.1.26, |abs err| < 3.4e-7):
//   gelu(x) = max(x,0) - 0.5|x| t P(t) exp(-x^2/2),   t = 1 / (1 + p|x|/sqrt2)
// With ONE wave per SIMD nothing hides the latency of a dependent VALU chain (first hardware run of this kernel: one value at a
// time, 4-6 dependent instructions per MFMA gap, cost ~53 cycles per gap on top of the 32 of the MFMA).  So the 16 values of a
// slab are processed FOUR AT A TIME, one micro-operation per value in turn: consecutive instructions belong to different values
// and are independent; the same value comes round again four issue slots later.  A slab is 4 groups x 16 rows x 4 values = 256
// micro-operations, numbered idx = 64 group + 4 row + value; an iteration spreads them evenly over its 48 MFMA gaps.
struct GeluQuad {
    float x[4], a[4], t[4], w[4], p[4];
};
template <int IDX>  // every index is a constant expression: the arrays stay in registers
__device__ __forceinline__ void gelu_uop(const float (&acc)[16], GeluQuad& q, half8 (&pf)[2]) {
    constexpr int grp = IDX >> 6, row = (IDX >> 2) & 15, k = IDX & 3, v = 4 * grp + k;
    // 14 instructions per value (|x| is a source modifier; the 0.5 of 0.5|x| lives in the polynomial coefficients);
    // rows 14, 15 are empty so that a group stays 64 slots long
    if constexpr (row == 0) {
        q.x[k] = acc[v];
        q.t[k] = __builtin_fmaf(__builtin_fabsf(acc[v]), 0.3275911f * 0.70710678f, 1.0f);
    } else if constexpr (row == 1) q.t[k] = __builtin_amdgcn_rcpf(q.t[k]);
    else if constexpr (row == 2) q.w[k] = q.x[k] * -0.72134752f;  // -0.5 log2(e) x
    else if constexpr (row == 3) q.w[k] = q.w[k] * q.x[k];          // -0.5 log2(e) x^2
    else if constexpr (row == 4) q.p[k] = __builtin_fmaf(q.t[k], 0.5f * 1.061405429f, 0.5f * -1.453152027f);
    else if constexpr (row == 5) q.p[k] = __builtin_fmaf(q.p[k], q.t[k], 0.5f * 1.421413741f);
    else if constexpr (row == 6) q.p[k] = __builtin_fmaf(q.p[k], q.t[k], 0.5f * -0.284496736f);
    else if constexpr (row == 7) q.p[k] = __builtin_fmaf(q.p[k], q.t[k], 0.5f * 0.254829592f);
    else if constexpr (row == 8) q.a[k] = __builtin_fabsf(q.x[k]) * q.t[k];
    else if constexpr (row == 9) q.p[k] = q.p[k] * q.a[k];  // 0.5 |x| t P(t)
    else if constexpr (row == 10) q.w[k] = __builtin_amdgcn_exp2f(q.w[k]);
    else if constexpr (row == 11) q.a[k] = __builtin_amdgcn_fmed3f(q.x[k], 0.0f, __builtin_inff());  // max(x, 0)
    else if constexpr (row == 12) q.p[k] = __builtin_fmaf(-q.p[k], q.w[k], q.a[k]);
    else if constexpr (row == 13) pf[v >> 3][v & 7] = (_Float16)q.p[k];  // fp16 into the B fragment of the second product
}
template <int LO, int... E>
__device__ __forceinline__ void gelu_range(std::integer_sequence<int, E...>, const float (&acc)[16], GeluQuad& q, half8 (&pf)[2]) {
    (gelu_uop<LO + E>(acc, q, pf), ...);
}

// One iteration of the skewed pipeline.  FC1: first product of the slab in stage w1s (bias bs) -> accn;  GEL: GELU of acc[0..16)
// -> pfcur;  FC2: second product of the slab in stage w2s with pfprev -> o.  48 slots, slot i = MFMA i (24 of FC1 then 24 of
// FC2) followed by GELU stage i (value i / 3, stage i % 3) and the fragment read of slot i + 4.
struct M3Ctx {
    const unsigned char* w1s;
    const unsigned char* w2s;
    int a1[8], b20, b21;
};
template <int SLOT>
__device__ __forceinline__ half8 m3_frag(const M3Ctx& c) {
    if constexpr (SLOT < 24) return *(const half8*)(c.w1s + c.a1[SLOT & 7] + 256 * (SLOT >> 3));
    else {
        constexpr int n = SLOT - 24, u = n / ML_NJ, j = n % ML_NJ;
        return *(const half8*)(c.w2s + (u ? c.b21 : c.b20) + 2048 * j);
    }
}
template <bool FC1, bool FC2, int SLOT>
constexpr bool m3_live() { return SLOT < 24 ? FC1 : (SLOT < 48 ? FC2 : false); }

template <bool FC1, bool FC2, bool GEL, int I>
__device__ __forceinline__ void m3_slot(const M3Ctx& c, const half8 (&xf)[ML_KS], float16v (&accn)[2], const float (&acc)[16],
                                        const half8 (&pfprev)[2], half8 (&pfcur)[2], float16v (&o)[ML_NJ], half8 (&ring)[4], GeluQuad& gq) {
    if constexpr (m3_live<FC1, FC2, I>()) {
        if constexpr (I < 24) {
            // two accumulators in turn: an instruction issued between two MFMAs on the SAME accumulator costs ~43 cycles
            // (MI355X_MICROARCH.md, per-instruction constants) -- and every gap here carries GELU micro-operations
            accn[I & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[I & 3], xf[I], accn[I & 1], 0, 0, 0);
        } else {
            constexpr int n = I - 24, u = n / ML_NJ, j = n % ML_NJ;
            o[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[I & 3], pfprev[u], o[j], 0, 0, 0);
        }
        if constexpr (m3_live<FC1, FC2, I + 4>()) ring[I & 3] = m3_frag<I + 4>(c);
    }
    if constexpr (GEL) {
        constexpr int lo = (256 * I) / 48, hi = (256 * (I + 1)) / 48;
        gelu_range<lo>(std::make_integer_sequence<int, hi - lo>{}, acc, gq, pfcur);
    }
    __builtin_amdgcn_sched_barrier(0);
}
template <bool FC1, bool FC2, bool GEL, int... I>
__device__ __forceinline__ void m3_slots(std::integer_sequence<int, I...>, const M3Ctx& c, const half8 (&xf)[ML_KS], float16v (&accn)[2],
                                         const float (&acc)[16], const half8 (&pfprev)[2], half8 (&pfcur)[2], float16v (&o)[ML_NJ], half8 (&ring)[4],
                                         GeluQuad& gq) {
    (m3_slot<FC1, FC2, GEL, I>(c, xf, accn, acc, pfprev, pfcur, o, ring, gq), ...);
}

// One iteration of the skewed pipeline.  FC1: first product of the slab in stage w1s (bias bs) -> accn;  GEL: GELU of acc[0..16)
// -> pfcur;  FC2: second product of the slab in stage w2s with pfprev -> o.  48 slots, slot i = MFMA i (24 of FC1 then 24 of
// FC2), the fragment read of slot i + 4 and GELU micro-operations [256 i / 48, 256 (i + 1) / 48).
template <bool FC1, bool FC2, bool GEL>
__device__ __forceinline__ void m3_iteration(const unsigned char* w1s, const unsigned char* w2s, const int (&a1)[8], int b20, int b21,
                                             const float* bs, const half8 (&xf)[ML_KS], float16v (&accn)[2], const float (&acc)[16],
                                             const half8 (&pfprev)[2], half8 (&pfcur)[2], float16v (&o)[ML_NJ]) {
    M3Ctx c;
    c.w1s = w1s;
    c.w2s = w2s;
#pragma unroll
    for (int i = 0; i < 8; ++i) c.a1[i] = a1[i];
    c.b20 = b20;
    c.b21 = b21;
    if (FC1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4v bv = *(const float4v*)(bs + 8 * q);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                accn[0][4 * q + i] = bv[i];  // bias in one of the two partial sums
                accn[1][4 * q + i] = 0.0f;
            }
        }
    }
    half8 ring[4];
    constexpr int first = FC1 ? 0 : 24;
    if constexpr (m3_live<FC1, FC2, first>()) {
        ring[0] = m3_frag<first>(c);
        ring[1] = m3_frag<first + 1>(c);
        ring[2] = m3_frag<first + 2>(c);
        ring[3] = m3_frag<first + 3>(c);
    }
    GeluQuad gq;
    m3_slots<FC1, FC2, GEL>(std::make_integer_sequence<int, 48>{}, c, xf, accn, acc, pfprev, pfcur, o, ring, gq);
}

// w1:  [F][384] fp16 (nn.Linear weight; slab s = rows 32s .. 32s+31, contiguous)
// w2p: [F/32][384][32] fp16 with the k permutation of leann_amd/encoder.py: fused_mlp_k_permutation
// ABL: ablation bits for on-hardware diagnosis (LEANN_MI355X_ABLATE; 0 = the product kernel): 1 = no weight DMA after the
// prologue (stale LDS), 2 = no counted wait / barrier per slab, 4 = no GELU stages (second product on stale fragments).
// Results are wrong by construction with any bit set; only the timing is of interest.
template <int ABL>
__global__ __launch_bounds__(256) LM_ONE_WAVE_PER_SIMD void k_mlp_fused_h384_v3(
    const __half* __restrict__ x, const __half* __restrict__ w1, const float* __restrict__ b1, const __half* __restrict__ w2p,
    const float* __restrict__ b2, const __half* __restrict__ gamma, const __half* __restrict__ beta, __half* __restrict__ out, int T,
    int F, float eps) {
    extern __shared__ __align__(16) unsigned char smem[];
    float* b1s = (float*)(smem + M3_B1_OFF);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r31 = lane & 31, g = lane >> 5;
    const int token = blockIdx.x * 128 + wv * 32 + r31;
    const bool valid = token < T;
    const int nslab = F >> 5;  // >= 4 (host-checked)
    const unsigned char* g1 = (const unsigned char*)w1;
    const unsigned char* g2 = (const unsigned char*)w2p;

    // ---- prologue: W1 slabs 0..2 and W2 slab 0 in flight; x^T fragments and b1 meanwhile ----
    int w1off[6];
    m3_w1_offsets(tid, w1off);
    m3_issue_w1(g1, smem + M3_W1_OFF, tid, w1off);
    m3_issue_w1(g1 + M3_SLAB, smem + M3_W1_OFF + M3_SLAB, tid, w1off);
    m3_issue_w1(g1 + 2 * M3_SLAB, smem + M3_W1_OFF + 2 * M3_SLAB, tid, w1off);
    m3_issue_w2(g2, smem + M3_W2_OFF, tid);
    half8 xf[ML_KS];
    {
        const _Float16* xr = (const _Float16*)x + (int64_t)(valid ? token : 0) * ML_H + 8 * g;
#pragma unroll
        for (int ks = 0; ks < ML_KS; ++ks) {
            half8 v = *(const half8*)(xr + 16 * ks);
            const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            xf[ks] = valid ? v : z;
        }
    }
    for (int i = tid; i < F; i += 256) b1s[i] = b1[i];

    // fragment addresses.  W1: row r31, chunk c = 2 ks + g at position (c & ~15) | ((c ^ r31) & 15): the low part depends on ks & 7
    int a1[8];
#pragma unroll
    for (int k7 = 0; k7 < 8; ++k7) a1[k7] = r31 * 768 + ((((2 * k7 + g) ^ r31) & 15) << 4);
    // W2: row 32 j + r31, chunk (2u + g) ^ ((r31 >> 2) & 3); u = 1 flips bit 5 of the byte offset
    const int b20 = r31 * 64 + ((g ^ ((r31 >> 2) & 3)) << 4), b21 = b20 ^ 32;

    float16v o[ML_NJ];
#pragma unroll
    for (int j = 0; j < ML_NJ; ++j) o[j] = (float16v){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float16v accn[2];  // first product: even / odd k-steps
    float acc[16];
    half8 pfa[2], pfb[2];  // GELU outputs of the slab being activated / of the previous slab
    if (ABL & 4) pfa[0] = pfa[1] = pfb[0] = pfb[1] = xf[0];
    M3_WAIT_VM(0);
    __syncthreads();  // b1s written, every wave's DMA pieces landed (nothing is in flight: a plain barrier is fine here)

    // first product of slab 0, nothing to overlap it with
    m3_iteration<true, false, false>(smem + M3_W1_OFF, nullptr, a1, b20, b21, b1s + 4 * g, xf, accn, acc, pfb, pfa, o);

    // iteration s: FC1 of slab s+1 (stage (s+1) % 3), GELU of slab s, FC2 of slab s-1 (stage (s-1) % 3).
    // At its top: W1(s+1) and W2(s-1) must have landed; issued after them, one iteration ago: W1(s+2), W2(s).
    // Then W1(s+3) and W2(s+1) are issued into the stages W1(s) / W2(s-2) occupied -- idle once every wave passed the barrier.
    auto top = [&](int s) {  // everything an iteration does before its 48 slots
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = accn[0][r] + accn[1][r];
        if (s > 0 && !(ABL & 2)) {
            const int pend = (s + 2 < nslab ? 6 : 0) + 6;  // pieces of W1(s+2), W2(s) issued at the top of s-1
            if (pend == 12) M3_WAIT_VM(12);
            else M3_WAIT_VM(6);
            M3_BARRIER();
        }
        if (ABL & 1) return;
        if (s + 3 < nslab) m3_issue_w1(g1 + (int64_t)(s + 3) * M3_SLAB, smem + M3_W1_OFF + ((s + 3) % M3_STAGES) * M3_SLAB, tid, w1off);
        if (s + 1 < nslab) m3_issue_w2(g2 + (int64_t)(s + 1) * M3_SLAB, smem + M3_W2_OFF + ((s + 1) % M3_STAGES) * M3_SLAB, tid);
    };
    auto w1_stage = [&](int s) { return (const unsigned char*)smem + M3_W1_OFF + ((s + 1) % M3_STAGES) * M3_SLAB; };
    auto w2_stage = [&](int s) { return (const unsigned char*)smem + M3_W2_OFF + ((s + 2) % M3_STAGES) * M3_SLAB; };  // (s - 1) mod 3
    // s = 0: no second product yet
    top(0);
    m3_iteration<true, false, !(ABL & 4)>(w1_stage(0), nullptr, a1, b20, b21, b1s + 32 + 4 * g, xf, accn, acc, pfb, pfa, o);
    pfb[0] = pfa[0];
    pfb[1] = pfa[1];
    // steady state: one basic block per iteration
    for (int s = 1; s + 1 < nslab; ++s) {
        top(s);
        m3_iteration<true, true, !(ABL & 4)>(w1_stage(s), w2_stage(s), a1, b20, b21, b1s + 32 * (s + 1) + 4 * g, xf, accn, acc, pfb, pfa, o);
        pfb[0] = pfa[0];
        pfb[1] = pfa[1];
    }
    // s = nslab - 1: no first product left
    top(nslab - 1);
    m3_iteration<false, true, !(ABL & 4)>(nullptr, w2_stage(nslab - 1), a1, b20, b21, nullptr, xf, accn, acc, pfb, pfa, o);
    pfb[0] = pfa[0];
    pfb[1] = pfa[1];
    // second product of the last slab (its W2 slab was waited for at the top of the last iteration: pend covered it)
    M3_WAIT_VM(0);
    M3_BARRIER();
    m3_iteration<false, true, false>(nullptr, smem + M3_W2_OFF + ((nslab - 1) % M3_STAGES) * M3_SLAB, a1, b20, b21, nullptr, xf, accn, acc, pfb,
                                     pfa, o);
    // the epilogue's ~150 read-only loads must not be hoisted above the slab loop (they would be spilled): an opaque copy
    // of the lane's half index ties their addresses to this point of the program
    int g_e = g;
    LM_KEEP_LOCAL(g_e);
    mlp_epilogue(o, x, b2, gamma, beta, out, token, valid, g_e, eps);
}

}  // namespace lm

int lm_mlp_fused_v3_launch(const void* d_x, const void* d_w1, const float* d_b1, const void* d_w2p, const float* d_b2, const void* d_gamma,
                           const void* d_beta, void* d_out, int64_t tokens, int32_t ffn, float eps, void* stream) {
    using namespace lm;
    const size_t shmem = (size_t)M3_B1_OFF + (size_t)ffn * 4;
    if (ffn < 128 || shmem > 160 * 1024) return 1;  // not applicable: the caller takes variant 2
    dim3 grid((unsigned)((tokens + 127) / 128)), block(256);
    const char* ab = getenv("LEANN_MI355X_ABLATE");
    const int abl = ab ? atoi(ab) : 0;
#define M3_GO(A)                                                                                                                     \
    case A:                                                                                                                           \
        LM_HIP(hipFuncSetAttribute((const void*)k_mlp_fused_h384_v3<A>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));     \
        hipLaunchKernelGGL(k_mlp_fused_h384_v3<A>, grid, block, shmem, (hipStream_t)stream, (const __half*)d_x, (const __half*)d_w1, \
                           d_b1, (const __half*)d_w2p, d_b2, (const __half*)d_gamma, (const __half*)d_beta, (__half*)d_out,            \
                           (int)tokens, ffn, eps);                                                                                      \
        break
    switch (abl) {
        M3_GO(0); M3_GO(1); M3_GO(2); M3_GO(3); M3_GO(4); M3_GO(7);
        default: LM_FAIL(LM_EINVAL, "LEANN_MI355X_ABLATE must be 0, 1, 2, 3, 4 or 7");
    }
#undef M3_GO
    LM_HIP(hipGetLastError());
    return LM_OK;
}
