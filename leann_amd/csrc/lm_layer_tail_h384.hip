// lm_layer_tail_h384.hip -- the second half of a BERT layer (hidden 384) in ONE kernel, fourth generation of the fused feed-forward block:
//
//     x = LayerNorm(resid + attn W_o^T + b_o) * gamma1 + beta1
//     y = LayerNorm(x + GELU(x W1^T + b1) W2^T + b2) * gamma + beta                     attn, resid, y: [T, 384] fp16
//
// Role in the reference: the attention output projection and the FFN inside compute_embeddings' BERT forward
// (leann/embedding_compute.py:229-239).  This is the dominant kernel of the selective-recompute search (60 % of a step).
//
// Geometry as before (measured alternatives are in DESIGN.md): one 256-thread workgroup = 4 waves, one per SIMD, 128 tokens;
// everything transposed (out^T = W x^T: weights are the MFMA A operand streamed L2 -> LDS by LDS-DMA, the token block is the B
// operand and lives in registers), the 384-wide activations and the 1536-wide intermediate never leave the registers.
//
// WHY A FOURTH GENERATION.  The third (lm_mlp_fused_v3.hip) sat at 33 % of the dense fp16 peak for two rounds.  Its steady-state
// loop is ISSUE bound, not matrix-pipe bound: ~470 instructions per 48 MFMAs (1536 matrix-pipe cycles) -- 50 s_waitcnt, 32
// v_accvgpr_read + 16 v_add + 32 v_mov for the two partial sums of the first product, ~40 address / stage-rotation instructions,
// 36 for the twelve DMA pieces -- on ONE wave per SIMD, where every issue slot beyond ~5 per MFMA is matrix-pipe idle time
// (MI355X_MICROARCH.md, per-instruction constants).  This generation removes instructions instead of re-ordering them:
//   * the two products ALTERNATE, MFMA by MFMA (even slots: first product of slab s+1, odd slots: second product of slab s-1).
//     Two consecutive MFMAs never share an accumulator, so the first product needs ONE accumulator chain (started from the bias
//     vector as the C operand) instead of two partial sums: no zero / bias moves, no sum;
//   * the chain's accumulator and the GELU output fragments are double buffered BY NAME and the three-stage weight rings are
//     addressed with compile-time stage numbers (the loop is unrolled six times = lcm(2, 3)): no copies, no stage arithmetic --
//     a fragment read is one ds_read_b128 with base register + immediate;
//   * the weights are stored as ready-made LDS images (host side: lm_layer_tail_pack_h384), so the DMA is a linear copy:
//     one M0 write per FOUR 1 KB pieces (global_load_lds_dwordx4 with instruction offsets 0 / 1024 / 2048 / 3072, the offset
//     applies to the global and to the LDS address alike), waves 0, 1 stream W1, waves 2, 3 stream W2: 12 loads + 3 M0 writes
//     per wave and iteration; no per-piece address arithmetic; the tail end re-reads the last slab instead of branching;
//   * GELU micro-operations as before (8.5 per value, one transcendental), three behind each MFMA.
// Budget of an iteration: 48 MFMA + 48 ds_read + 136 GELU + 16 accumulator reads + 21 DMA + waits.
//
// Envelope: ffn a multiple of 192 (the six-fold unrolled pipeline), 192 <= ffn <= 1728; other shapes take the general GEMM path
// (lm_gemm_f16).  LDS map: [0, 72 K) W1 ring, [72 K, 144 K) W2 ring, then b1, b2, b_o, gamma / beta of both norms (fp32).
#include <cstdlib>
#include <cstring>
#include <utility>

#include "lm_h384_stream.h"

namespace lm {

typedef _Float16 t4_half2 __attribute__((ext_vector_type(2)));

constexpr int T4_W1_OFF = 0;
constexpr int T4_W2_OFF = 3 * T4_SLAB;
constexpr int T4_B1_OFF = 6 * T4_SLAB;

// the product instance of the kernel (the diagnosis library carries the other schedule variants: LEANN_MI355X_TAIL4)
#ifndef LM_T4_DM
#define LM_T4_DM 2
#endif
#ifndef LM_T4_RD
#define LM_T4_RD 4
#endif
#ifndef LM_T4_GF
#define LM_T4_GF 1
#endif
#ifndef LM_T4_WO_RING
#define LM_T4_WO_RING 2  // stages of the W_o ring in the prologue: 2 (product) or 6 (A/B; see the prologue comment)
#endif
#ifndef LM_T4_WM
#define LM_T4_WM 0
#endif

// ---- GELU ------------------------------------------------------------------------------------------------------------------------
// exact (erf) GELU in scalar fp32, 8.5 instructions per value, ONE transcendental:
//     gelu(x) = max(x, 0) - |x| * 0.5 erfc(|x| / sqrt2),      0.5 erfc(u / sqrt2) = 2^(-1 - u q(u)),
//   q = degree-4 polynomial fitted (weighted minimax on the error of the RESULT, u in [0, 9]) to -log2(erfc(u / sqrt2)) / u; |error|
//   < 1e-6 absolute over all x in fp32 (fp16 output: <= 1 ulp for x > -3, absolute < 6e-7 below).  With ONE wave per SIMD nothing
//   hides the latency of a dependent VALU chain, so the 16 values of a slab are processed FOUR AT A TIME, one micro-operation per
//   value in turn: a slab is 4 groups x 9 rows x 4 values micro-operations, numbered idx = 36 group + 4 row + value, spread evenly
//   over the 48 MFMA gaps of an iteration.  FORM 1 = plain C (the compiler picks the instructions; also the host-emulated build),
//   FORM 4 = the same arithmetic as `asm volatile` VOP3 micro-operations with source modifiers (pins the interleaved order).
struct T4Gelu {
    float x[4], a[4], w[4], p[4];
};
template <int FORM, int IDX>  // every index is a constant expression: the arrays stay in registers
__device__ __forceinline__ void t4_gelu_uop(const float16v& acc, T4Gelu& q, half8 (&pf)[2]) {
    constexpr int R = 9, NF = 4;
    constexpr int grp = IDX / (NF * R), row = (IDX % (NF * R)) / NF, k = IDX % NF, v = NF * grp + k;
#ifndef LM_EMULATED_DEVICE
    if constexpr (FORM == 4) {
        if constexpr (row == 0) {
            q.x[k] = acc[v];
            asm volatile("v_fma_f32 %0, |%1|, %2, %3" : "=v"(q.p[k]) : "v"(q.x[k]), "s"(-0.0004881171917077154f), "v"(0.007198805455118418f));
        } else if constexpr (row == 1) asm volatile("v_fma_f32 %0, %1, |%2|, %3" : "=v"(q.p[k]) : "v"(q.p[k]), "v"(q.x[k]), "s"(-0.052146803587675095f));
        else if constexpr (row == 2) asm volatile("v_fma_f32 %0, %1, |%2|, %3" : "=v"(q.p[k]) : "v"(q.p[k]), "v"(q.x[k]), "s"(-0.4595957100391388f));
        else if constexpr (row == 3) asm volatile("v_fma_f32 %0, %1, |%2|, %3" : "=v"(q.p[k]) : "v"(q.p[k]), "v"(q.x[k]), "s"(-1.1510006189346313f));
        else if constexpr (row == 4) asm volatile("v_fma_f32 %0, %1, |%2|, -1.0" : "=v"(q.w[k]) : "v"(q.p[k]), "v"(q.x[k]));
        else if constexpr (row == 5) asm volatile("v_exp_f32 %0, %1" : "=v"(q.w[k]) : "v"(q.w[k]));
        else if constexpr (row == 6) asm volatile("v_max_f32 %0, 0, %1" : "=v"(q.a[k]) : "v"(q.x[k]));
        else if constexpr (row == 7) asm volatile("v_fma_f32 %0, -|%1|, %2, %3" : "=v"(q.p[k]) : "v"(q.x[k]), "v"(q.w[k]), "v"(q.a[k]));
        else if constexpr (row == 8 && (k & 1) == 0) {
            unsigned r;
            asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(q.p[k]), "v"(q.p[k + 1]));
            u32x4 u = __builtin_bit_cast(u32x4, pf[v >> 3]);
            u[(v & 7) >> 1] = r;
            pf[v >> 3] = __builtin_bit_cast(half8, u);
        }
        return;
    }
#endif
    if constexpr (row == 0) {
        q.x[k] = acc[v];
        q.p[k] = __builtin_fmaf(__builtin_fabsf(q.x[k]), -0.0004881171917077154f, 0.007198805455118418f);
    } else if constexpr (row == 1) q.p[k] = __builtin_fmaf(q.p[k], __builtin_fabsf(q.x[k]), -0.052146803587675095f);
    else if constexpr (row == 2) q.p[k] = __builtin_fmaf(q.p[k], __builtin_fabsf(q.x[k]), -0.4595957100391388f);
    else if constexpr (row == 3) q.p[k] = __builtin_fmaf(q.p[k], __builtin_fabsf(q.x[k]), -1.1510006189346313f);
    else if constexpr (row == 4) q.w[k] = __builtin_fmaf(q.p[k], __builtin_fabsf(q.x[k]), -1.0f);  // -1 - u q(u)
    else if constexpr (row == 5) q.w[k] = __builtin_amdgcn_exp2f(q.w[k]);
    else if constexpr (row == 6) {  // max(x, 0)
#ifdef LM_EMULATED_DEVICE
        q.a[k] = __builtin_amdgcn_fmed3f(q.x[k], 0.0f, __builtin_inff());
#else
        asm("v_max_f32 %0, 0, %1" : "=v"(q.a[k]) : "v"(q.x[k]));  // one instruction: the C forms canonicalise x (an accumulator read) first
#endif
    } else if constexpr (row == 7) q.p[k] = __builtin_fmaf(-__builtin_fabsf(q.x[k]), q.w[k], q.a[k]);
    else if constexpr ((k & 1) == 0) {  // row 8: values k, k + 1 -> two fp16 of the B fragment of the second product
        const float2v pr = {q.p[k], q.p[k + 1]};
        const t4_half2 h = __builtin_convertvector(pr, t4_half2);
        pf[v >> 3][v & 7] = h[0];
        pf[v >> 3][(v & 7) + 1] = h[1];
    }
}
template <int FORM, int LO, int... E>
__device__ __forceinline__ void t4_gelu_range(std::integer_sequence<int, E...>, const float16v& acc, T4Gelu& q, half8 (&pf)[2]) {
    (t4_gelu_uop<FORM, LO + E>(acc, q, pf), ...);
}

// ---- one iteration of the skewed pipeline ------------------------------------------------------------------------------------------
// Iteration s (ST = s % 3, P = s & 1):   FC1 of slab s + 1   |   GELU of slab s   |   FC2 of slab s - 1
//   FC1: accn[P ^ 1] = b1(slab s + 1) + W1(stage (ST + 1) % 3) x^T            24 MFMAs, EVEN slots
//   GEL: pf[P]       = GELU(accn[P])                                          144 micro-operation slots (136 used) over the 48 gaps
//   FC2: o          += W2(stage (ST + 2) % 3) pf[P ^ 1]                        24 MFMAs, ODD slots (u-major: the two MFMAs of a tile
//                                                                              are 24 slots apart)
//   DMA: W1(s + 3) -> W1 stage ST (waves 0, 1), W2(s + 1) -> W2 stage (ST + 1) % 3 (waves 2, 3).
// The fragment stream is CONTINUOUS over the iterations: slot i reads the fragment of slot i + RD, which from slot 48 - RD on belongs
// to iteration s + 1 (stages (ST + 2) % 3 of W1, ST of W2; the bias vector of slab s + 2 comes with it) -- no iteration starts with an
// empty ring.  ONE barrier per iteration, in front of slot 48 - RD:
//   * read after write: the next iteration's slabs W1(s + 2), W2(s) were requested during iteration s - 1; each wave has waited for
//     its own pieces (vmcnt = the pieces it has requested in THIS iteration so far), the barrier makes that true of every wave;
//   * write after read: the requests of iteration s + 1 go to the stages iteration s reads.  Every wave has ISSUED its last read of
//     them (slot 47 - RD) when it arrives at the barrier, and the first request of iteration s + 1 is RD + 3 slots (> 300 cycles)
//     behind the barrier plus an L2 round trip away from landing: an LDS read that was issued before the barrier has long returned.
//     (The emulation's reads are synchronous, so there the barrier orders them outright.)
struct T4Addr {  // per-lane LDS addresses of the fragment reads (ring base + position inside a slab image); see the kernel
    const unsigned char* a1[8];
    const unsigned char* b2[2];
};
struct T4Dma {  // per-wave DMA description: its half (12 KB) of its matrix' slabs
    unsigned char* stg[3];  // destination (its half) by ST
    unsigned voff[3];       // 16 lane + 4096 k
};
template <bool FC1, bool FC2, int I>
constexpr bool t4_live() { return (I & 1) ? FC2 : FC1; }

template <int ST, int I>
__device__ __forceinline__ half8 t4_frag(const T4Addr& c) {
    if constexpr ((I & 1) == 0) {
        constexpr int ks = I >> 1;
        return *(const half8*)(c.a1[ks & 7] + ((ST + 1) % 3) * T4_SLAB + 256 * (ks >> 3));
    } else {
        constexpr int n = I >> 1, u = n / ML_NJ, j = n % ML_NJ;
        return *(const half8*)(c.b2[u] + ((ST + 2) % 3) * T4_SLAB + 2048 * j);
    }
}

// DM: where the iteration's twelve DMA pieces are issued.  -1 none; 0 all behind slot 0 (three groups of four, as generation 3);
// 1 three groups of four behind slots 3 / 19 / 35; 2 single pieces behind slots 3, 7, ..., 47
template <int DM, int ST, int I>
__device__ __forceinline__ void t4_slot_dma(const T4Dma& d, const unsigned char* src) {
    if constexpr (DM == 0) {
        if constexpr (I == 0) {
            t4_dma_group<4>(src, d.voff[0], d.stg[ST]);
            t4_dma_group<4>(src, d.voff[1], d.stg[ST] + 4096);
            t4_dma_group<4>(src, d.voff[2], d.stg[ST] + 8192);
        }
    } else if constexpr (DM == 1) {
        if constexpr (I % 16 == 3) t4_dma_group<4>(src, d.voff[I / 16], d.stg[ST] + 4096 * (I / 16));
    } else if constexpr (DM == 2) {
        if constexpr (I % 4 == 3) t4_dma_piece<(I / 4) % 4>(src, d.voff[I / 16], d.stg[ST] + 4096 * (I / 16));
    }
}
template <int DM>
constexpr int t4_pieces_before(int slot) {  // pieces a wave has requested in this iteration when it arrives at `slot`
    return DM == 0 ? 12 : (DM == 1 ? 4 * ((slot + 12) / 16) : (DM == 2 ? slot / 4 : 0));
}
// Carried from iteration to iteration: the fragment ring and the bias vector of the NEXT first product.
template <int RD>
struct T4Carry {
    half8 ring[RD];
    float16v biasv;
};

// NFC1 / NFC2: does the NEXT iteration have a first / second product (its fragments and bias are read here).
// WM: 1 = one counted LDS wait per four slots (needs RD = 8: the four fragments of slots i .. i + 3 are then older than the four reads
// behind them), 0 = the compiler's per-MFMA waits.
template <bool FC1, bool FC2, bool NFC1, bool NFC2, int GEL, int ST, int P, int RD, int DM, int WM, int I>
__device__ __forceinline__ void t4_slot(const T4Addr& c, const float* bs_next, const half8 (&xf)[ML_KS], float16v (&accn)[2], half8 (&pf)[2][2],
                                        float16v (&o)[ML_NJ], T4Carry<RD>& cy, T4Gelu& gq, const T4Dma& d, const unsigned char* src) {
    if constexpr (I == 48 - RD && (NFC1 || NFC2)) {  // the iteration's barrier: see above
        t4_wait_vm<t4_pieces_before<DM>(48 - RD)>();
        T4_BARRIER();
    }
    if constexpr (WM == 1 && RD == 8 && FC1 && FC2 && I % 4 == 0) {
        // outstanding behind the fragment of slot I + 3: the four reads of slots I - 4 .. I - 1 (+ the four bias reads of slots 40 .. 43)
        if constexpr (I == 44 && NFC1) t4_wait_lgkm<8>();
        else if constexpr (I == 44 && !NFC1 && !NFC2) t4_wait_lgkm<0>();
        else t4_wait_lgkm<4>();
        __builtin_amdgcn_sched_barrier(0);  // a register-only MFMA would otherwise be hoisted over the wait (and get a wait of its own)
    }
    if constexpr (t4_live<FC1, FC2, I>()) {
        if constexpr ((I & 1) == 0) {
            constexpr int ks = I >> 1;
            if constexpr (ks == 0) accn[P ^ 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cy.ring[I % RD], xf[0], cy.biasv, 0, 0, 0);
            else accn[P ^ 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cy.ring[I % RD], xf[ks], accn[P ^ 1], 0, 0, 0);
        } else {
            constexpr int n = I >> 1, u = n / ML_NJ, j = n % ML_NJ;
            o[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cy.ring[I % RD], pf[P ^ 1][u], o[j], 0, 0, 0);
        }
    }
    if constexpr (I + RD < 48) {
        if constexpr (t4_live<FC1, FC2, I + RD>()) cy.ring[I % RD] = t4_frag<ST, I + RD>(c);
    } else {
        if constexpr (t4_live<NFC1, NFC2, I + RD - 48>()) cy.ring[I % RD] = t4_frag<(ST + 1) % 3, I + RD - 48>(c);
    }
    if constexpr (NFC1 && I >= 40 && I < 44) {  // bias vector of the next first product (accumulator register 4 q + i <-> hidden unit 8 q + 4 g + i)
        constexpr int q = I - 40;
        const float4v bv = *(const float4v*)(bs_next + 8 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) cy.biasv[4 * q + i] = bv[i];
    }
    if constexpr (GEL != 0) {
        constexpr int lo = (144 * I) / 48, hi = (144 * (I + 1)) / 48;
        t4_gelu_range<GEL, lo>(std::make_integer_sequence<int, hi - lo>{}, accn[P], gq, pf[P]);
    }
    t4_slot_dma<DM, ST, I>(d, src);
    __builtin_amdgcn_sched_barrier(0);
}
template <bool FC1, bool FC2, bool NFC1, bool NFC2, int GEL, int ST, int P, int RD, int DM, int WM, int... I>
__device__ __forceinline__ void t4_slots(std::integer_sequence<int, I...>, const T4Addr& c, const float* bs_next, const half8 (&xf)[ML_KS],
                                         float16v (&accn)[2], half8 (&pf)[2][2], float16v (&o)[ML_NJ], T4Carry<RD>& cy, T4Gelu& gq, const T4Dma& d,
                                         const unsigned char* src) {
    (t4_slot<FC1, FC2, NFC1, NFC2, GEL, ST, P, RD, DM, WM, I>(c, bs_next, xf, accn, pf, o, cy, gq, d, src), ...);
}
// the ring and the bias vector in front of the very first iteration
template <bool FC1, bool FC2, int ST, int RD, int... I>
__device__ __forceinline__ void t4_ring_fill(std::integer_sequence<int, I...>, const T4Addr& c, T4Carry<RD>& cy) {
    (([&] {
         if constexpr (t4_live<FC1, FC2, I>()) cy.ring[I % RD] = t4_frag<ST, I>(c);
     }()),
     ...);
}

// bs_next = this lane's bias slice of the NEXT iteration's FC1 slab: b1s + 32 (s + 2) + 4 g
template <bool FC1, bool FC2, bool NFC1, bool NFC2, int GEL, int ST, int P, int RD, int DM, int WM>
__device__ __forceinline__ void t4_iteration(const T4Addr& c, const float* bs_next, const half8 (&xf)[ML_KS], float16v (&accn)[2],
                                             half8 (&pf)[2][2], float16v (&o)[ML_NJ], T4Carry<RD>& cy, const T4Dma& d, const unsigned char* src) {
    T4Gelu gq;
    t4_slots<FC1, FC2, NFC1, NFC2, GEL, ST, P, RD, DM, WM>(std::make_integer_sequence<int, 48>{}, c, bs_next, xf, accn, pf, o, cy, gq, d, src);
}

// ---- attention output projection: one 32-wide k slab (24 dense MFMAs: tile j of o, k-steps u = 0, 1) ---------------------------------
// (Round 6 measured two changes to this phase -- 1.5 k cycles per slab against a 0.77 k matrix-pipe floor at one wave per SIMD -- and kept neither: the
// fragment ring 8 deep instead of 4, and on top of it the ring CONTINUOUS over the twelve slabs with the slab's barrier moved in front of slot 16 (no refill
// bubble per slab, the barrier under fragments in flight): 688.8 / 691.8 / 689.5 us per 262 k tokens for continuous + 8 / 8 / this form, three alternating
// processes each (profiles/r6_kbench_layer_tail_outprojection_ring_depth_and_continuous_ring_equal.jsonl).  Neither the LDS latency of the fragment
// reads nor the per-slab refill is what this phase waits for.)
__device__ __forceinline__ void t4_outproj_slab(const unsigned char* b20, const unsigned char* b21, const half8& a0, const half8& a1,
                                                float16v (&o)[ML_NJ]) {
    half8 ring[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) ring[n] = *(const half8*)(b20 + 2048 * n);
#pragma unroll
    for (int n = 0; n < 2 * ML_NJ; ++n) {
        const int j = n % ML_NJ;
        o[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[n & 3], n < ML_NJ ? a0 : a1, o[j], 0, 0, 0);
        if (n + 4 < 2 * ML_NJ) ring[n & 3] = *(const half8*)((n + 4 < ML_NJ ? b20 : b21) + 2048 * ((n + 4) % ML_NJ));
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- first LayerNorm, inside the registers -----------------------------------------------------------------------------------------
//     x = LayerNorm(o + residual) * gamma1 + beta1,       o = attn W_o^T + b_o in the accumulators
// written as fp16 straight into the B-fragment registers of the first product, in ACCUMULATOR order: fragment 2 j + u, element e <-
// register 8 u + e of tile j = feature 32 j + 16 u + 8 (e >> 2) + 4 g + (e & 3).  W1's columns are packed in that k order
// (leann_amd/encoder.py: pack_w1_acc_order), so x never leaves the registers.  The residual (the layer's input rows, fragments in
// natural order) is brought into accumulator order with lane swaps: lane (token, g) owns features 16 ks + 8 g + e and needs
// 32 j + 8 q + 4 g + i, i.e. half of every fragment register pair trades places with the partner lane (token, g ^ 1).
__device__ __forceinline__ void t4_ln1(float16v (&o)[ML_NJ], const half8 (&rf)[ML_KS], half8 (&xf)[ML_KS], const float* gam_s,
                                       const float* bet_s, int g, float eps) {
    // ONE pass for both moments (sum and sum of squares of v = o + residual; var = E[v^2] - mean^2 in fp32 over 384 values of order 1),
    // then y = (v rstd - mean rstd) gamma + beta as two packed fused multiply-adds per pair; gamma / beta are fp32 in LDS (no conversions).
    // 4.75 instructions per value against the 8.25 of the two-pass form with fp16 parameters: this phase runs with the matrix pipe idle.
    __builtin_amdgcn_sched_barrier(0);
    float2v sa = {0.f, 0.f}, sb = {0.f, 0.f}, qa = {0.f, 0.f}, qb = {0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < ML_KS; ++ks) {
        u32x4 d = __builtin_bit_cast(u32x4, rf[ks]);
        uint32_t a0 = d[0], b0 = d[2], a1 = d[1], b1 = d[3];
        lane32_swap(a0, b0);  // a: features of the even q (i = 0, 1), b: of the odd q
        lane32_swap(a1, b1);  // the same for i = 2, 3
        const int j = ks >> 1, qe = 2 * (ks & 1);
        const t4_half2 ea = __builtin_bit_cast(t4_half2, a0), eb = __builtin_bit_cast(t4_half2, a1);
        const t4_half2 oa = __builtin_bit_cast(t4_half2, b0), ob = __builtin_bit_cast(t4_half2, b1);
        float2v v0 = (float2v){o[j][4 * qe], o[j][4 * qe + 1]} + (float2v){(float)ea[0], (float)ea[1]};
        float2v v1 = (float2v){o[j][4 * qe + 2], o[j][4 * qe + 3]} + (float2v){(float)eb[0], (float)eb[1]};
        float2v v2 = (float2v){o[j][4 * qe + 4], o[j][4 * qe + 5]} + (float2v){(float)oa[0], (float)oa[1]};
        float2v v3 = (float2v){o[j][4 * qe + 6], o[j][4 * qe + 7]} + (float2v){(float)ob[0], (float)ob[1]};
        o[j][4 * qe] = v0[0];
        o[j][4 * qe + 1] = v0[1];
        o[j][4 * qe + 2] = v1[0];
        o[j][4 * qe + 3] = v1[1];
        o[j][4 * qe + 4] = v2[0];
        o[j][4 * qe + 5] = v2[1];
        o[j][4 * qe + 6] = v3[0];
        o[j][4 * qe + 7] = v3[1];
        sa += v0 + v2;
        sb += v1 + v3;
        qa = __builtin_elementwise_fma(v0, v0, qa);
        qb = __builtin_elementwise_fma(v1, v1, qb);
        qa = __builtin_elementwise_fma(v2, v2, qa);
        qb = __builtin_elementwise_fma(v3, v3, qb);
    }
    float sum = (sa[0] + sa[1]) + (sb[0] + sb[1]), sq = (qa[0] + qa[1]) + (qb[0] + qb[1]);
    sum += __shfl_xor(sum, 32);
    sq += __shfl_xor(sq, 32);
    const float mean = sum * (1.0f / ML_H);
    const float var = __builtin_fmaxf(__builtin_fmaf(-mean, mean, sq * (1.0f / ML_H)), 0.0f);
    const float rstd = rsqrtf(var + eps);
    const float2v rs = {rstd, rstd}, nmr = {-mean * rstd, -mean * rstd};
    // gamma / beta of tile j + 1 are read while tile j is normalised (left to itself the compiler emits read, read, wait, use)
    float4v gv[4], bv[4], gn[4], bn[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        gv[q] = *(const float4v*)(gam_s + 8 * q + 4 * g);
        bv[q] = *(const float4v*)(bet_s + 8 * q + 4 * g);
    }
#pragma unroll
    for (int j = 0; j < ML_NJ; ++j) {
        if (j + 1 < ML_NJ) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                gn[q] = *(const float4v*)(gam_s + 32 * (j + 1) + 8 * q + 4 * g);
                bn[q] = *(const float4v*)(bet_s + 32 * (j + 1) + 8 * q + 4 * g);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            half8 h;
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int q = 2 * u + qq;
                const float2v n0 = __builtin_elementwise_fma((float2v){o[j][4 * q], o[j][4 * q + 1]}, rs, nmr);
                const float2v n1 = __builtin_elementwise_fma((float2v){o[j][4 * q + 2], o[j][4 * q + 3]}, rs, nmr);
                const float2v y0 = __builtin_elementwise_fma(n0, (float2v){gv[q][0], gv[q][1]}, (float2v){bv[q][0], bv[q][1]});
                const float2v y1 = __builtin_elementwise_fma(n1, (float2v){gv[q][2], gv[q][3]}, (float2v){bv[q][2], bv[q][3]});
                const t4_half2 h0 = __builtin_convertvector(y0, t4_half2), h1 = __builtin_convertvector(y1, t4_half2);
                h[4 * qq] = h0[0];
                h[4 * qq + 1] = h0[1];
                h[4 * qq + 2] = h1[0];
                h[4 * qq + 3] = h1[1];
            }
            xf[2 * j + u] = h;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            gv[q] = gn[q];
            bv[q] = bn[q];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// ---- epilogue:  y = LayerNorm(o + x) * gamma + beta  (b2 is already in the accumulators), written as fp16 ------------------------------
//   * the residual x is the first LayerNorm's output the wave still holds as B fragments, already in ACCUMULATOR order (fragment
//     2 j + u, element e = the value that belongs to register 8 u + e of tile j): no lane traffic;
//   * gamma / beta are read from an LDS copy (two addresses per instruction: a broadcast);
//   * the fp16 results go through the wave's own 24 KB of the (now idle) weight stages -- [32 tokens][48 chunks of 16 B], chunk c
//     of row r at position (c & ~15) | ((c ^ r) & 15): conflict-free ds_write_b64 in, ds_read_b128 out -- and leave as 24 fully
//     coalesced 1 KB stores per wave (a wave's 32 token rows are one contiguous 24 KB block of the output).
__device__ __forceinline__ void t4_epilogue(float16v (&o)[ML_NJ], const half8 (&xf)[ML_KS], const float* gam_s, const float* bet_s,
                                            unsigned char* tile, __half* __restrict__ out, int64_t token0, int T, int r31, int g, int lane,
                                            float eps) {
    __builtin_amdgcn_sched_barrier(0);
    float2v sa = {0.f, 0.f}, sb = {0.f, 0.f}, qa = {0.f, 0.f}, qb = {0.f, 0.f};  // one pass for both moments, as t4_ln1
#pragma unroll
    for (int ks = 0; ks < ML_KS; ++ks) {
        const int j = ks >> 1, r0 = 8 * (ks & 1);
        const half8 d = xf[ks];
        float2v v0 = (float2v){o[j][r0], o[j][r0 + 1]} + (float2v){(float)d[0], (float)d[1]};
        float2v v1 = (float2v){o[j][r0 + 2], o[j][r0 + 3]} + (float2v){(float)d[2], (float)d[3]};
        float2v v2 = (float2v){o[j][r0 + 4], o[j][r0 + 5]} + (float2v){(float)d[4], (float)d[5]};
        float2v v3 = (float2v){o[j][r0 + 6], o[j][r0 + 7]} + (float2v){(float)d[6], (float)d[7]};
        o[j][r0] = v0[0];
        o[j][r0 + 1] = v0[1];
        o[j][r0 + 2] = v1[0];
        o[j][r0 + 3] = v1[1];
        o[j][r0 + 4] = v2[0];
        o[j][r0 + 5] = v2[1];
        o[j][r0 + 6] = v3[0];
        o[j][r0 + 7] = v3[1];
        sa += v0 + v2;
        sb += v1 + v3;
        qa = __builtin_elementwise_fma(v0, v0, qa);
        qb = __builtin_elementwise_fma(v1, v1, qb);
        qa = __builtin_elementwise_fma(v2, v2, qa);
        qb = __builtin_elementwise_fma(v3, v3, qb);
    }
    float sum = (sa[0] + sa[1]) + (sb[0] + sb[1]), sq = (qa[0] + qa[1]) + (qb[0] + qb[1]);
    sum += __shfl_xor(sum, 32);
    sq += __shfl_xor(sq, 32);
    const float mean = sum * (1.0f / ML_H);
    const float var = __builtin_fmaxf(__builtin_fmaf(-mean, mean, sq * (1.0f / ML_H)), 0.0f);
    const float rstd = rsqrtf(var + eps);
    const float2v rs = {rstd, rstd}, nmr = {-mean * rstd, -mean * rstd};
    unsigned char* trow = tile + r31 * 768 + 8 * g;
    float4v gv[4], bv[4], gn[4], bn[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        gv[q] = *(const float4v*)(gam_s + 8 * q + 4 * g);
        bv[q] = *(const float4v*)(bet_s + 8 * q + 4 * g);
    }
#pragma unroll
    for (int j = 0; j < ML_NJ; ++j) {
        if (j + 1 < ML_NJ) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                gn[q] = *(const float4v*)(gam_s + 32 * (j + 1) + 8 * q + 4 * g);
                bn[q] = *(const float4v*)(bet_s + 32 * (j + 1) + 8 * q + 4 * g);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = 4 * j + q;
            const float2v n0 = __builtin_elementwise_fma((float2v){o[j][4 * q], o[j][4 * q + 1]}, rs, nmr);
            const float2v n1 = __builtin_elementwise_fma((float2v){o[j][4 * q + 2], o[j][4 * q + 3]}, rs, nmr);
            const float2v y0 = __builtin_elementwise_fma(n0, (float2v){gv[q][0], gv[q][1]}, (float2v){bv[q][0], bv[q][1]});
            const float2v y1 = __builtin_elementwise_fma(n1, (float2v){gv[q][2], gv[q][3]}, (float2v){bv[q][2], bv[q][3]});
            const t4_half2 h0 = __builtin_convertvector(y0, t4_half2), h1 = __builtin_convertvector(y1, t4_half2);
            const half4 y = {h0[0], h0[1], h1[0], h1[1]};
            *(half4*)(trow + ((c & ~15) | ((c ^ r31) & 15)) * 16) = y;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            gv[q] = gn[q];
            bv[q] = bn[q];
        }
    }
    LM_WAVE_SYNC();  // the rows were written by other lanes of this wave (lock-step on the GPU: program order is enough)
    // the wave's tile back out in linear order: chunk L = 64 i + lane = (row L / 48, chunk L % 48) is bytes [16 L, 16 L + 16) of the
    // wave's 24 KB of output
    unsigned char* obase = (unsigned char*)out + token0 * (ML_H * 2);
    const int rows_valid = (int)((int64_t)T - token0 < 32 ? (int64_t)T - token0 : 32);
#pragma unroll
    for (int b = 0; b < 3; ++b) {  // eight reads in flight, then their eight stores
        u32x4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int L = 64 * (8 * b + i) + lane, row = L / 48, c = L - 48 * row;
            v[i] = *(const u32x4*)(tile + row * 768 + ((c & ~15) | ((c ^ row) & 15)) * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int L = 64 * (8 * b + i) + lane;
#if defined(LM_T4_NT) && LM_T4_NT && !defined(LM_EMULATED_DEVICE)
            if (L < 48 * rows_valid) __builtin_nontemporal_store(v[i], (u32x4*)(obase + 16 * L));
#else
            if (L < 48 * rows_valid) *(u32x4*)(obase + 16 * L) = v[i];
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// attn: [T][384] fp16 (attention output); wo_img / w1_img / w2_img: the LDS images written by lm_layer_tail_pack_h384
struct T4Pre {
    const __half* attn;
    const __half* wo_img;
    const float* bo;
    const __half* gamma1;
    const __half* beta1;
    float eps1;
    int stagger;  // first-round workgroups (blockIdx < 256) start ((37 b) & 255) / 256 * stagger x 1024 cycles late: see the kernel
};

// ABL & 64 (diagnosis builds only): ten s_memtime stamps per workgroup (wave 0), written over the first 80 bytes of the workgroup's
// first output row when it is done -- where the cycles of a workgroup go (scripts/kbench.cpp "tail4stamps").
#ifdef LM_EMULATED_DEVICE
#define T4_STAMP(i) ((void)0)
#else
#define T4_STAMP(i)                              \
    if constexpr ((ABL & 64) != 0) {             \
        __builtin_amdgcn_sched_barrier(0);       \
        stamp[i] = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);       \
    }
#endif

// ABL: 0 product, 64 stamps.  DM: DMA placement (t4_slot_dma).  RD: fragment read-ahead in slots (4 or 8).  GF: GELU form (1 C, 4 asm).
// WM: LDS wait form (t4_slot).
template <int ABL, int DM, int RD, int GF, int WM>
__global__ __launch_bounds__(256) LM_ONE_WAVE_PER_SIMD void k_layer_tail_h384(
    const __half* __restrict__ resid, T4Pre pre, const __half* __restrict__ w1_img, const float* __restrict__ b1,
    const __half* __restrict__ w2_img, const float* __restrict__ b2, const __half* __restrict__ gamma, const __half* __restrict__ beta,
    __half* __restrict__ out, int T, int F, float eps) {
    extern __shared__ __align__(16) unsigned char smem[];
    [[maybe_unused]] unsigned long long stamp[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#ifndef LM_EMULATED_DEVICE
    // Every workgroup of a launch does the same work in the same time, so the 256 resident workgroups stay in lock-step for the
    // whole launch: all read their 2 x 98 KB of rows at the same moment (HBM at its limit) and then leave HBM idle for the 150,000
    // cycles of their MFMA loops.  Starting the FIRST round's workgroups spread over `stagger` x 1024 cycles shifts the CUs'
    // phases against each other for the rest of the launch, which turns the bursts into a steady trickle.
    if (pre.stagger > 0 && blockIdx.x < 256) {
        const int n = (int)(((blockIdx.x * 37u) & 255u) * (unsigned)pre.stagger) >> 8;
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(16);
    }
#endif
    T4_STAMP(0);
    float* b1s = (float*)(smem + T4_B1_OFF);
    float* b2s = b1s + F;  // behind b1: b2, gamma, beta, b_o, gamma1, beta1 -- 384 floats each (the LayerNorm parameters converted once per workgroup)
    float* gam_s = b2s + ML_H;
    float* bet_s = gam_s + ML_H;
    float* bos = bet_s + ML_H;
    float* gam1_s = bos + ML_H;
    float* bet1_s = gam1_s + ML_H;
    const int tid = threadIdx.x, lane = tid & 63;
#ifdef LM_EMULATED_DEVICE
    const int wv = tid >> 6;
#else
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave uniform, and known to be
#endif
    const int r31 = lane & 31, g = lane >> 5;
    const int token = blockIdx.x * 128 + wv * 32 + r31;
    const bool valid = token < T;
    const int nslab = F >> 5;  // a multiple of 6, >= 6 (host-checked)
    const unsigned char* g1 = (const unsigned char*)w1_img;
    const unsigned char* g2 = (const unsigned char*)w2_img;
    const unsigned char* go = (const unsigned char*)pre.wo_img;
    const unsigned voff0 = (unsigned)lane * 16u;

    // ---- prologue + attention output projection.  Six 24 KB stages (0 .. 5; 0 .. 2 = the W1 ring, 3 .. 5 = the W2 ring of the feed-forward loop).
    //      LM_T4_WO_RING == 2 (the PRODUCT form): stages 0 .. 3 are row tiles -- wave w owns tile w: its attention rows, then, re-filled by its
    //      own DMA as soon as those are in registers, its residual rows -- and stages 4, 5 a two-stage ring for the twelve W_o slabs (slab s in
    //      stage 4 + (s & 1); slabs 0, 1 arrive with the prologue, slab s + 1 at the top of slab s >= 1: ONE slab of lead, vmcnt(0) per slab).
    //      LM_T4_WO_RING == 6: the attention rows leave stage w before the projection starts and all six stages carry W_o slabs (slab s in
    //      stage OST[s % 6], FIVE slabs of lead, counted vmcnt(24)); the residual rows follow in the refill slots of slabs 7 .. 10, a quarter
    //      per wave.  Same time as the two-stage form on most MI355X boxes (673 vs 675 us per 262k tokens), 1085 vs 731 us on the others
    //      (profiles/r4_session4_layer_tail_bisect_same_source_two_speeds.jsonl: the deep DMA queue of one workgroup's prologue delays the
    //      feed-forward weight stream of its neighbours) -- kept for A/B only. ----
    constexpr int WOR = LM_T4_WO_RING;
    static_assert(WOR == 2 || WOR == 6, "LM_T4_WO_RING: 2 or 6");
    unsigned char* mytile = smem + wv * T4_SLAB;
    const int tok0c = (int)blockIdx.x * 128 + wv * 32 < T ? (int)blockIdx.x * 128 + wv * 32 : T - 1;  // wave uniform
    const int rows_valid = T - tok0c < 32 ? T - tok0c : 32;
    t4_copy_quarter(go, smem + 4 * T4_SLAB, wv, voff0);
    t4_copy_quarter(go + T4_SLAB, smem + 5 * T4_SLAB, wv, voff0);
    t4_issue_rows<0, 24>((const unsigned char*)pre.attn + (int64_t)tok0c * (ML_H * 2), rows_valid, mytile, lane);
    for (int i = tid; i < F; i += 256) b1s[i] = b1[i];
    for (int i = tid; i < ML_H; i += 256) {
        b2s[i] = b2[i];
        gam_s[i] = (float)((const _Float16*)gamma)[i];
        bet_s[i] = (float)((const _Float16*)beta)[i];
        bos[i] = pre.bo[i];
        gam1_s[i] = (float)((const _Float16*)pre.gamma1)[i];
        bet1_s[i] = (float)((const _Float16*)pre.beta1)[i];
    }

    // fragment addresses inside a slab image.  W1 (and row tiles): row r31, chunk c = 2 ks + g at position (c & ~15) | ((c ^ r31) & 15):
    // the low part depends on ks & 7.  W2 / W_o: row 32 j + r31, chunk (2 u + g) ^ ((r31 >> 2) & 3); u = 1 flips bit 5 of the byte offset
    int a1[8];
#pragma unroll
    for (int k7 = 0; k7 < 8; ++k7) a1[k7] = r31 * 768 + ((((2 * k7 + g) ^ r31) & 15) << 4);
    const int b20 = r31 * 64 + ((g ^ ((r31 >> 2) & 3)) << 4), b21 = b20 ^ 32;
    T4Addr ad;
#pragma unroll
    for (int k7 = 0; k7 < 8; ++k7) ad.a1[k7] = smem + T4_W1_OFF + a1[k7];
    ad.b2[0] = smem + T4_W2_OFF + b20;
    ad.b2[1] = smem + T4_W2_OFF + b21;
    // DMA role in the feed-forward loop: waves 0, 1 the two halves of the W1 slabs, waves 2, 3 of the W2 slabs
    const int mrole = wv >> 1, hrole = wv & 1;
    T4Dma dm;
#pragma unroll
    for (int st = 0; st < 3; ++st)
        dm.stg[st] = smem + (mrole ? T4_W2_OFF + ((st + 1) % 3) * T4_SLAB : T4_W1_OFF + st * T4_SLAB) + 12288 * hrole;
#pragma unroll
    for (int k = 0; k < 3; ++k) dm.voff[k] = voff0 + 4096u * k;
    const unsigned char* dbase = (mrole ? g2 : g1) + 12288 * hrole;
    const int doff = mrole ? 1 : 3;
    auto dsrc = [&](int s) {  // source of iteration s: slab s + 3 of W1 / s + 1 of W2; past the end the last slab again (never read)
        const int sl = s + doff < nslab ? s + doff : nslab - 1;
        return dbase + (int64_t)sl * T4_SLAB;
    };

    float16v o[ML_NJ];  // out-projection, then the second product
    float16v accn[2];   // first product: slab parity
    half8 pf[2][2];     // GELU outputs: slab parity x k-step
    half8 xf[ML_KS];    // the attention-output fragments first, the first LayerNorm's output (accumulator order) afterwards
    half8 rf[ML_KS];    // the residual rows (the layer's input), natural fragment order
    T4_WAIT_VM(0);
    __syncthreads();  // LDS fills written, every wave's DMA pieces landed (nothing is in flight: a plain barrier is fine here)
    T4_STAMP(1);
    {
        // attention rows: tile -> B fragments (the reads of a W1 fragment: conflict free)
        const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < ML_KS; ++ks) {
            const half8 v = *(const half8*)(mytile + a1[ks & 7] + 256 * (ks >> 3));
            xf[ks] = valid ? v : z;
        }
    }
    T4_WAIT_LGKM0();
    if constexpr (WOR == 6) {
        T4_BARRIER();  // every wave has its rows in registers: stages 0 .. 3 join the W_o ring
        t4_copy_quarter(go + 2 * T4_SLAB, smem + 0 * T4_SLAB, wv, voff0);
        t4_copy_quarter(go + 3 * T4_SLAB, smem + 1 * T4_SLAB, wv, voff0);
        t4_copy_quarter(go + 4 * T4_SLAB, smem + 2 * T4_SLAB, wv, voff0);
        t4_copy_quarter(go + 5 * T4_SLAB, smem + 3 * T4_SLAB, wv, voff0);
    } else {
        LM_WAVE_SYNC();  // the tile is re-filled by this wave's own DMA: program order on the GPU
        t4_issue_rows<0, 24>((const unsigned char*)resid + (int64_t)tok0c * (ML_H * 2), rows_valid, mytile, lane);
    }
#pragma unroll
    for (int j = 0; j < ML_NJ; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4v bv = *(const float4v*)(bos + 32 * j + 8 * q + 4 * g);
#pragma unroll
            for (int i = 0; i < 4; ++i) o[j][4 * q + i] = bv[i];
        }
    constexpr int OST[6] = {4, 5, 0, 1, 2, 3};  // WOR == 6 (RST[r] = {4, 5, 0, 1}: where the residual rows of wave r land -- the stage slab r + 6 frees)
#pragma unroll
    for (int s = 0; s < ML_H / 32; ++s) {
        if constexpr (WOR == 6) {
            if (s > 0) {
                if (s > 1) t4_wait_vm<24>();  // slab s has landed: the four requests behind it (6 pieces each) may be in flight
                T4_BARRIER();                 // ... for every wave; and every wave is done with slab s - 1, whose stage is refilled:
                unsigned char* freed = smem + OST[(s - 1) % 6] * T4_SLAB;
                if (s + 5 < ML_H / 32) t4_copy_quarter(go + (int64_t)(s + 5) * T4_SLAB, freed, wv, voff0);
                else if (s - 7 < 4) {  // s = 7 .. 10: the residual rows of wave r = s - 7 into stage RST[r] (= the stage just freed), a quarter per wave
                    const int r = s - 7;
                    const int tr = (int)blockIdx.x * 128 + r * 32 < T ? (int)blockIdx.x * 128 + r * 32 : T - 1;
                    const int rv = T - tr < 32 ? T - tr : 32;
                    const unsigned char* rrows = (const unsigned char*)resid + (int64_t)tr * (ML_H * 2);
                    if (wv == 0) t4_issue_rows<0, 6>(rrows, rv, freed, lane);
                    else if (wv == 1) t4_issue_rows<6, 6>(rrows, rv, freed, lane);
                    else if (wv == 2) t4_issue_rows<12, 6>(rrows, rv, freed, lane);
                    else t4_issue_rows<18, 6>(rrows, rv, freed, lane);
                }
            }
        } else if (s > 0) {
            // top of slab s >= 1: slab s has landed (s >= 2: vmcnt(0) -- it is the youngest request; slabs 0, 1 came with the prologue), every
            // wave is done with slab s - 1 (barrier), whose stage takes slab s + 1
            if (s > 1) T4_WAIT_VM(0);
            T4_BARRIER();
            if (s + 1 < ML_H / 32) t4_copy_quarter(go + (int64_t)(s + 1) * T4_SLAB, smem + (4 + ((s + 1) & 1)) * T4_SLAB, wv, voff0);
        }
        const unsigned char* stg = smem + (WOR == 6 ? OST[s % 6] : 4 + (s & 1)) * T4_SLAB;
        t4_outproj_slab(stg + b20, stg + b21, xf[2 * s], xf[2 * s + 1], o);
    }
    T4_STAMP(8);
    if constexpr (WOR == 6) {
        T4_WAIT_VM(0);
        T4_BARRIER();  // the residual rows have landed (every wave's share of every tile)
    }
    {
        // residual rows: the wave's tile -> fragments (WOR == 2: its own DMA, older than every W_o slab waited for above; WOR == 6: stage
        // RST[wv]).  Then all six stages are idle once every wave is here: the feed-forward block's first weights (W1 slabs 0..2, W2 slab 0)
        // arrive under the LayerNorm
        const unsigned char* rtile = WOR == 6 ? smem + (wv < 2 ? 4 + wv : wv - 2) * T4_SLAB : mytile;
        const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < ML_KS; ++ks) {
            const half8 v = *(const half8*)(rtile + a1[ks & 7] + 256 * (ks >> 3));
            rf[ks] = valid ? v : z;
        }
    }
    T4_WAIT_LGKM0();
    T4_BARRIER();
    t4_copy_quarter(g1, smem + T4_W1_OFF, wv, voff0);
    t4_copy_quarter(g1 + T4_SLAB, smem + T4_W1_OFF + T4_SLAB, wv, voff0);
    t4_copy_quarter(g1 + 2 * T4_SLAB, smem + T4_W1_OFF + 2 * T4_SLAB, wv, voff0);
    t4_copy_quarter(g2, smem + T4_W2_OFF, wv, voff0);
    t4_ln1(o, rf, xf, gam1_s, bet1_s, g, pre.eps1);
    T4_WAIT_VM(0);
    T4_BARRIER();
    T4_STAMP(9);
#pragma unroll
    for (int j = 0; j < ML_NJ; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4v bv = *(const float4v*)(b2s + 32 * j + 8 * q + 4 * g);
#pragma unroll
            for (int i = 0; i < 4; ++i) o[j][4 * q + i] = bv[i];
        }

    // ---- feed-forward block.  Iteration s: FC1 of slab s + 1 (W1 stage (s + 1) % 3), GELU of slab s, FC2 of slab s - 1 (W2 stage
    //      (s - 1) % 3); at its top W1(s + 1) and W2(s - 1) must have landed -- they were requested during iteration s - 2 --
    //      while the twelve pieces of iteration s - 1 may still be in flight: vmcnt(12), then the barrier that also frees the stages
    //      this iteration's requests go to. ----
    const float* bl = b1s + 4 * g;
    T4Carry<RD> cy;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4v bv = *(const float4v*)(bl + 8 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) cy.biasv[4 * q + i] = bv[i];
    }
    t4_ring_fill<true, false, 2, RD>(std::make_integer_sequence<int, RD>{}, ad, cy);
    // FC1 of slab 0 alone ("iteration -1": ST = 2, P = 1), nothing to overlap it with
    t4_iteration<true, false, true, false, 0, 2, 1, RD, -1, WM>(ad, bl + 32, xf, accn, pf, o, cy, dm, nullptr);
    T4_STAMP(2);
    t4_iteration<true, false, true, true, GF, 0, 0, RD, DM, WM>(ad, bl + 64, xf, accn, pf, o, cy, dm, dsrc(0));
    T4_STAMP(3);
#define T4_IT(ST, P, S) t4_iteration<true, true, true, true, GF, ST, P, RD, DM, WM>(ad, bl + 32 * ((S) + 2), xf, accn, pf, o, cy, dm, dsrc(S))
    {  // steady state s = 1 .. nslab - 2: (nslab - 6) / 6 blocks of six (compile-time stage / parity numbers), then four more
        int s = 1;
        for (; s + 5 <= nslab - 3; s += 6) {
            if (s == 19) { T4_STAMP(4); }
            T4_IT(1, 1, s);
            T4_IT(2, 0, s + 1);
            T4_IT(0, 1, s + 2);
            T4_IT(1, 0, s + 3);
            T4_IT(2, 1, s + 4);
            T4_IT(0, 0, s + 5);
        }
        T4_IT(1, 1, s);
        T4_IT(2, 0, s + 1);
        T4_IT(0, 1, s + 2);
        // s = nslab - 2: the next iteration has no first product
        t4_iteration<true, true, false, true, GF, 1, 0, RD, DM, WM>(ad, nullptr, xf, accn, pf, o, cy, dm, dsrc(s + 3));
    }
#undef T4_IT
    T4_STAMP(5);
    // s = nslab - 1 (= 5 mod 6): no first product left, nothing to request; its barrier has every request behind it (vmcnt(0))
    t4_iteration<false, true, false, true, GF, 2, 1, RD, -1, WM>(ad, nullptr, xf, accn, pf, o, cy, dm, nullptr);
    // second product of the last slab ("iteration nslab": ST = 0, P = 0)
    t4_iteration<false, true, false, false, 0, 0, 0, RD, -1, WM>(ad, nullptr, xf, accn, pf, o, cy, dm, nullptr);
    T4_STAMP(6);
    __syncthreads();  // every wave is done with the weight stages: they become the output staging tiles
    t4_epilogue(o, xf, gam_s, bet_s, smem + wv * T4_SLAB, out, (int64_t)blockIdx.x * 128 + wv * 32, T, r31, g, lane, eps);
#ifndef LM_EMULATED_DEVICE
    if constexpr ((ABL & 64) != 0) {
        __builtin_amdgcn_s_waitcnt(0);
        stamp[7] = __builtin_amdgcn_s_memtime();
        __syncthreads();  // every wave's rows are stored: the stamps go on top
        if (tid == 0) {
            unsigned long long* dst = (unsigned long long*)(out + (int64_t)blockIdx.x * 128 * ML_H);
#pragma unroll
            for (int i = 0; i < 10; ++i) dst[i] = stamp[i];
        }
    }
#endif
}

// ---- weight images ----------------------------------------------------------------------------------------------------------------
// 16-byte chunk L of an image <- chunk src(L) of the packed matrix.  KIND 0 (W1, accumulator k order, [F][384]): slabs of 32 rows x
// 48 chunks, position pos of row r holds chunk (pos & ~15) | ((pos ^ r) & 15).  KIND 1 (W2 / W_o slabs [n][384][32]): rows of 4
// chunks, position pos of row r holds chunk pos ^ ((r >> 2) & 3).  These are the XOR swizzles that make the fragment reads
// (ds_read_b128, lane = row) bank-conflict free; generation 3 applied them to the DMA source offsets on every piece.
__global__ void k_tail_image(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t nchunks, int kind) {
    const int64_t L = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (L >= nchunks) return;
    int64_t s;
    if (kind == 0) {
        const int64_t row = L / 48;
        const int pos = (int)(L - 48 * row), r = (int)(row & 31);
        s = row * 48 + ((pos & ~15) | ((pos ^ r) & 15));
    } else {
        const int64_t row = L >> 2;
        const int pos = (int)(L & 3);
        s = row * 4 + (pos ^ (int)((row >> 2) & 3));
    }
    dst[L] = src[s];
}

}  // namespace lm

extern "C" int lm_layer_tail_pack_h384(const void* d_wo_slabs, const void* d_w1_acc, const void* d_w2_slabs, int32_t ffn, void* d_wo_img,
                                       void* d_w1_img, void* d_w2_img, void* stream) {
    using namespace lm;
    if (!d_wo_slabs || !d_w1_acc || !d_w2_slabs || !d_wo_img || !d_w1_img || !d_w2_img || ffn <= 0 || ffn % 32)
        LM_FAIL(LM_EINVAL, "lm_layer_tail_pack_h384: bad arguments");
    const int64_t nwo = (int64_t)ML_H * ML_H / 8, nw = (int64_t)ffn * ML_H / 8;
    hipLaunchKernelGGL(k_tail_image, dim3((unsigned)((nwo + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint4*)d_wo_slabs, (uint4*)d_wo_img, nwo, 1);
    hipLaunchKernelGGL(k_tail_image, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint4*)d_w1_acc, (uint4*)d_w1_img, nw, 0);
    hipLaunchKernelGGL(k_tail_image, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint4*)d_w2_slabs, (uint4*)d_w2_img, nw, 1);
    LM_HIP(hipGetLastError());
    return LM_OK;
}

// W [n_out][384] (nn.Linear layout, device) -> the image lm_qkv_h384_f16 streams: 32-row slabs in the W1 form above
extern "C" int lm_qkv_pack_h384(const void* d_w, int32_t n_out, void* d_w_img, void* stream) {
    using namespace lm;
    if (!d_w || !d_w_img || n_out <= 0 || n_out % 32) LM_FAIL(LM_EINVAL, "lm_qkv_pack_h384: bad arguments");
    const int64_t nw = (int64_t)n_out * ML_H / 8;
    hipLaunchKernelGGL(k_tail_image, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint4*)d_w, (uint4*)d_w_img, nw, 0);
    LM_HIP(hipGetLastError());
    return LM_OK;
}

extern "C" int lm_layer_tail_h384_f16(const void* d_attn, const void* d_resid, const void* d_wo_img, const float* d_bo, const void* d_gamma1,
                                      const void* d_beta1, float eps1, const void* d_w1_img, const float* d_b1, const void* d_w2_img,
                                      const float* d_b2, const void* d_gamma, const void* d_beta, void* d_out, int64_t tokens, int32_t ffn,
                                      float eps, void* stream) {
    using namespace lm;
    if (tokens == 0) return LM_OK;
    if (!d_attn || !d_resid || !d_wo_img || !d_bo || !d_gamma1 || !d_beta1 || !d_w1_img || !d_b1 || !d_w2_img || !d_b2 || !d_gamma || !d_beta ||
        !d_out || tokens < 0 || tokens > 0x7fffffff)
        LM_FAIL(LM_EINVAL, "bad fused layer-tail arguments");
    const size_t shmem = (size_t)T4_B1_OFF + (size_t)(ffn > 0 ? ffn : 0) * 4 + ML_H * 24;  // + b2, b_o, gamma, beta, gamma1, beta1 (fp32)
    if (ffn < 192 || ffn % 192 || shmem > 160 * 1024) LM_FAIL(LM_EINVAL, "fused layer tail: ffn must be a multiple of 192 in [192, 1728]");
    dim3 grid((unsigned)((tokens + 127) / 128)), block(256);
    // environment switches are read once per process (the launch path of a B = 1 search runs this ~600 times per query)
    static const int stagger_env = [] { const char* sg = getenv("LEANN_MI355X_STAGGER"); return sg ? atoi(sg) : 40; }();  // spread of the first round's start times, x 1024 cycles (0 = off)
    const T4Pre pre = {(const __half*)d_attn, (const __half*)d_wo_img, d_bo, (const __half*)d_gamma1, (const __half*)d_beta1, eps1,
                       grid.x >= 512 ? stagger_env : 0};  // a launch of fewer than two rounds has no lock-step to break: no start delay (small-batch latency)
    KtScope kt(LM_KT_LAYER_TAIL, stream, (double)tokens * (4.0 * ffn * ML_H + 2.0 * ML_H * ML_H));
#define T4_GO(A, DM, RD, GF, WM)                                                                                                            \
    {                                                                                                                                       \
        static DynLdsAttr attr; /* per device; the attribute only ever needs to grow */                                                     \
        LM_HIP(ensure_dyn_lds(attr, (const void*)k_layer_tail_h384<A, DM, RD, GF, WM>, shmem));                                             \
        hipLaunchKernelGGL((k_layer_tail_h384<A, DM, RD, GF, WM>), grid, block, shmem, (hipStream_t)stream, (const __half*)d_resid, pre,     \
                           (const __half*)d_w1_img, d_b1, (const __half*)d_w2_img, d_b2, (const __half*)d_gamma, (const __half*)d_beta,      \
                           (__half*)d_out, (int)tokens, ffn, eps);                                                                          \
    }
#ifdef LM_DIAG  // diagnosis builds (stamps, schedule variants) exist only in the -DLM_DIAG library that scripts/build_kbench.sh makes
    // LEANN_MI355X_TAIL4 = 10000 stamps + 1000 DM + 100 RD/4 + 10 GF + WM; unset / 0 = the product instance
    const char* venv = getenv("LEANN_MI355X_TAIL4");  // per call: scripts/kbench.cpp switches variants inside one process
    const int var = venv ? atoi(venv) : 0;
#define T4_CASE(DM, RDQ, GF, WM)                                              \
    case 1000 * DM + 100 * RDQ + 10 * GF + WM: T4_GO(0, DM, 4 * RDQ, GF, WM) break; \
    case 10000 + 1000 * DM + 100 * RDQ + 10 * GF + WM: T4_GO(64, DM, 4 * RDQ, GF, WM) break;
    switch (var) {
        case 0: T4_GO(0, LM_T4_DM, LM_T4_RD, LM_T4_GF, LM_T4_WM) break;
        T4_CASE(0, 1, 1, 0) T4_CASE(1, 1, 1, 0) T4_CASE(2, 1, 1, 0)
        T4_CASE(0, 2, 1, 0) T4_CASE(1, 2, 1, 0) T4_CASE(2, 2, 1, 0)
        T4_CASE(0, 2, 1, 1) T4_CASE(1, 2, 1, 1) T4_CASE(2, 2, 1, 1)
        T4_CASE(2, 1, 4, 0) T4_CASE(2, 2, 4, 1) T4_CASE(1, 2, 4, 1)
        default: LM_FAIL(LM_EINVAL, "LEANN_MI355X_TAIL4: unknown variant");
    }
#undef T4_CASE
#else
    T4_GO(0, LM_T4_DM, LM_T4_RD, LM_T4_GF, LM_T4_WM)
#endif
#undef T4_GO
    LM_HIP(hipGetLastError());
    return LM_OK;
}
