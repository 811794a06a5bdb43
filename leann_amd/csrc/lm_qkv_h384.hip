// lm_qkv_h384.hip -- linear layer with 384 input features as a WEIGHT-STREAMING kernel with TWO waves per SIMD:
//
//     out[T][N] = x W^T + b            x [T][384] fp16, W the nn.Linear weight [N][384] as an LDS image, N % 128 == 0 (QKV: 1152)
//
// Role in the reference: the QKV projection inside compute_embeddings' BERT forward (leann/embedding_compute.py:229-239); 18 % of a
// selective-recompute search step.
//
// Why a second form next to lm_gemm_ws_h384.hip (weight-stationary: a workgroup keeps a 192 x 384 weight block in LDS and streams
// token tiles past it).  That kernel reads x SIX times (once per feature block) as fragment-shaped loads -- 32 rows x 32 B per
// instruction -- and the CU's vector-memory pipe, not the matrix pipe, bounds it: 24 % of the dense fp16 peak on the MI355X
// (profiles/r2_kbench_gemm_ws_phase_stamps.jsonl).  Here the loop nest is the layer tail's (lm_layer_tail_h384.hip):
//   * x is read ONCE: a wave keeps the x^T fragments of its 32 tokens in registers (96) for the whole kernel;
//   * W streams L2 -> LDS as 24 KB slabs (32 output features x 384 k, ready-made images: linear LDS-DMA, one M0 write per group)
//     through a four-stage ring, requested three slabs ahead; the ring is shared by EIGHT waves = 256 tokens per workgroup, so a slab
//     costs a wave 3 DMA pieces and half the L2 traffic per token of a four-wave workgroup;
//   * a wave needs ~210 registers (no 384-wide output block to hold: a slab's 32 x 32 result leaves at once), so two waves fit a SIMD:
//     what one wave loses to its barrier, its LDS round trips, its DMA issue and its epilogue, the other one fills -- the lever the
//     512-register layer tail cannot have;
//   * a slab's product runs as TWO accumulator chains (even / odd k-steps, bias in the first) so that consecutive MFMAs never share an
//     accumulator; the sum, the fp16 conversion and the store of slab s happen in the MFMA gaps of slab s + 1, through a per-wave
//     4 KB LDS tile ([32 tokens][64 features], XOR swizzled) so that the results leave as full 128-byte lines.
// LDS: [0, 96 K) the ring, [96 K, 128 K) eight output tiles, then the bias vector (4 N bytes).
#include <cstdlib>
#include <cstring>
#include <utility>

#include "lm_h384_stream.h"

namespace lm {

typedef _Float16 qk_half2 __attribute__((ext_vector_type(2)));

constexpr int QK_STAGES = 4;
constexpr int QK_TILE_OFF = QK_STAGES * T4_SLAB;  // 98304
constexpr int QK_TILE = 4096;                     // per wave: [32 tokens][64 features] fp16 (two slabs)
constexpr int QK_BIAS_OFF = QK_TILE_OFF + 8 * QK_TILE;

struct QkAddr {
    const unsigned char* a1[2][8];  // fragment addresses: ring stages 0, 1 / stages 2, 3 (a ds_read offset is 16 bits)
};
template <int ST, int KS>
__device__ __forceinline__ half8 qk_frag(const QkAddr& c) {
    return *(const half8*)(c.a1[ST >> 1][KS & 7] + (ST & 1) * T4_SLAB + 256 * (KS >> 3));
}

template <int RD>
struct QkCarry {
    half8 ring[RD];
    float16v biasv;  // bias of the NEXT slab (accumulator register 4 q + i <-> feature 8 q + 4 g + i of the slab)
};

// Epilogue of the PREVIOUS slab, one micro-step per call (STEP 0 .. 7), spread over the slots of the current slab:
//   0..3  registers 4 q .. 4 q + 3 of (A + B) -> four fp16 -> the wave's tile, row r31, chunk (4 c + q) ^ (r31 & 7) (c = slab parity), +8 g
//   4     (odd slabs only) the tile is complete: 64 features x 32 tokens
//   5, 6  (odd slabs) two of the four 16-byte pieces per lane each: tile -> registers -> global, 8 token rows x 128 B per instruction
template <int STEP, bool ODD>
__device__ __forceinline__ void qk_epilogue_step(const float16v& a, const float16v& b, unsigned char* tile, __half* __restrict__ out, int64_t tok0,
                                                 int rows_valid, int N, int feat0, int r31, int g, int lane) {
    if constexpr (STEP < 4) {
        constexpr int q = STEP;
        const float2v v0 = (float2v){a[4 * q], a[4 * q + 1]} + (float2v){b[4 * q], b[4 * q + 1]};
        const float2v v1 = (float2v){a[4 * q + 2], a[4 * q + 3]} + (float2v){b[4 * q + 2], b[4 * q + 3]};
        const qk_half2 h0 = __builtin_convertvector(v0, qk_half2), h1 = __builtin_convertvector(v1, qk_half2);
        const half4 y = {h0[0], h0[1], h1[0], h1[1]};
        *(half4*)(tile + r31 * 128 + ((((ODD ? 4 : 0) + q) ^ (r31 & 7)) << 4) + 8 * g) = y;
    } else if constexpr (ODD && STEP == 4) {
        LM_WAVE_SYNC();  // the rows were written by other lanes of this wave (lock-step on the GPU: program order is enough)
    } else if constexpr (ODD && (STEP == 5 || STEP == 6)) {
        if (rows_valid > 0) {  // wave uniform
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int id = 64 * (2 * (STEP - 5) + jj) + lane, c16 = id & 7;
                const int row = (id >> 3) < rows_valid ? (id >> 3) : rows_valid - 1;  // past the end: the last valid row once more (same bytes): the
                const u32x4 v = *(const u32x4*)(tile + row * 128 + ((c16 ^ (row & 7)) << 4));  // number of stores is what the counted waits assume
                // N > 0: row major [T][N].  N < 0 (= -T): HEAD major [N / 32 heads][T][32] -- the 64 features of this tile are two heads, a head's
                // 32 token rows x 64 B are one contiguous 2 KB run (the attention kernel then reads a (sequence, head)'s K, V, Q rows as contiguous
                // blocks instead of 64 B out of every 2 N)
                const int64_t off = N > 0 ? ((tok0 + row) * N + feat0) * 2 + 16 * c16
                                          : (((int64_t)(feat0 >> 5) + (c16 >> 2)) * (int64_t)(-N) + tok0 + row) * 64 + 16 * (c16 & 3);
                *(u32x4*)((unsigned char*)out + off) = v;
            }
        }
    }
}

// One slab: 24 MFMAs (slot I = k-step I; even k-steps chain A, odd chain B), the fragment stream continuous over the slabs (slot I
// reads the fragment of slot I + RD; from slot 24 - RD on that is the next slab's stage), the previous slab's epilogue in the gaps,
// the DMA of slab s + 3 behind slot 1, ONE barrier in front of slot 24 - RD (read after write: slab s + 1 was requested two slabs ago,
// every wave has waited for its own pieces; write after read: the stage the next slab's request goes to was last read in slot
// 23 - RD of this slab -- lm_layer_tail_h384.hip has the argument in full).
template <int ST, int P, bool NEXT, bool PREV, bool DMA, int RD, int I>
__device__ __forceinline__ void qk_slot(const QkAddr& c, const float* bs_next, const half8 (&xf)[ML_KS], float16v (&acc)[2][2], QkCarry<RD>& cy,
                                        unsigned char* tile, __half* __restrict__ out, int64_t tok0, int rows_valid, int N, int feat0_prev, int r31, int g,
                                        int lane, const unsigned char* dsrc, unsigned voff, unsigned char* ddst) {
    if constexpr (I == 24 - RD && NEXT) {
        // Counted wait for the pieces of slab s + 1 (requested in slab s - 2).  Behind them in this wave's queue, in issue order: the four
        // stores of slab s - 2's slots (if it had them), the three pieces of slab s - 1 (slab s + 2), its stores, this slab's three pieces
        // (slab s + 3) and its stores.  A slab's slots carry stores when the slab before it completed a 64-feature tile, i.e. in EVEN
        // slabs: two of the three slabs s - 2, s - 1, s when s is even (P = 0), one when it is odd.  The stores are issued unconditionally
        // (rows past the end of the matrix repeat the last valid row: qk_epilogue_step), except by a wave with no valid row at all.
        if (rows_valid > 0) t4_wait_vm<DMA ? (P == 0 ? 14 : 10) : 0>();
        else t4_wait_vm<DMA ? 6 : 0>();
        T4_BARRIER();
    }
    if constexpr (I == 0) acc[P][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cy.ring[0], xf[0], cy.biasv, 0, 0, 0);
    else if constexpr (I == 1) {
        const float16v z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[P][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cy.ring[1 % RD], xf[1], z, 0, 0, 0);
    } else acc[P][I & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cy.ring[I % RD], xf[I], acc[P][I & 1], 0, 0, 0);
    if constexpr (I + RD < 24) cy.ring[I % RD] = qk_frag<ST, I + RD>(c);
    else if constexpr (NEXT) cy.ring[I % RD] = qk_frag<(ST + 1) % QK_STAGES, I + RD - 24>(c);
    if constexpr (NEXT && I >= 16 && I < 20) {  // bias vector of the next slab
        constexpr int q = I - 16;
        const float4v bv = *(const float4v*)(bs_next + 8 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) cy.biasv[4 * q + i] = bv[i];
    }
    if constexpr (DMA && I == 1) t4_dma_group<3>(dsrc, voff, ddst);
    if constexpr (PREV && I >= 3 && I <= 15 && (I & 1)) {  // slots 3, 5, ..., 15: epilogue steps 0 .. 6 of the previous slab (parity P ^ 1)
        constexpr bool ODD = (P ^ 1) != 0;
        qk_epilogue_step<(I - 3) / 2, ODD>(acc[P ^ 1][0], acc[P ^ 1][1], tile, out, tok0, rows_valid, N, feat0_prev, r31, g, lane);
    }
    __builtin_amdgcn_sched_barrier(0);
}
template <int ST, int P, bool NEXT, bool PREV, bool DMA, int RD, int... I>
__device__ __forceinline__ void qk_slots(std::integer_sequence<int, I...>, const QkAddr& c, const float* bs_next, const half8 (&xf)[ML_KS],
                                         float16v (&acc)[2][2], QkCarry<RD>& cy, unsigned char* tile, __half* __restrict__ out, int64_t tok0, int rows_valid,
                                         int N, int feat0_prev, int r31, int g, int lane, const unsigned char* dsrc, unsigned voff, unsigned char* ddst) {
    (qk_slot<ST, P, NEXT, PREV, DMA, RD, I>(c, bs_next, xf, acc, cy, tile, out, tok0, rows_valid, N, feat0_prev, r31, g, lane, dsrc, voff, ddst), ...);
}

// grid: ceil(T / 256) workgroups of 512 threads.  w_img: lm_layer_tail_pack_h384's KIND 0 image of W [N][384] (lm_qkv_pack_h384).
template <int RD>
__global__ __launch_bounds__(512) LM_TWO_WAVES_PER_SIMD void k_qkv_h384(const __half* __restrict__ x, const __half* __restrict__ w_img,
                                                                         const float* __restrict__ bias, __half* __restrict__ out, int T, int Nfeat,
                                                                         int head_major) {
    extern __shared__ __align__(16) unsigned char smem[];
    float* bss = (float*)(smem + QK_BIAS_OFF);
    const int N = Nfeat;                       // output features (slabs, bias)
    const int Nst = head_major ? -T : Nfeat;   // what the epilogue's stores take: the row stride, or minus the token count for the head-major layout
    const int tid = threadIdx.x, lane = tid & 63;
#ifdef LM_EMULATED_DEVICE
    const int wv = tid >> 6;
#else
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    const int r31 = lane & 31, g = lane >> 5;
    const int64_t tok0 = (int64_t)blockIdx.x * 256 + wv * 32;
    const int rows_valid = (int)(T - tok0 < 32 ? (T - tok0 > 0 ? T - tok0 : 0) : 32);  // wave uniform
    const int nslab = N >> 5;  // a multiple of 4 (host-checked)
    const unsigned char* gw = (const unsigned char*)w_img;
    const unsigned voff0 = (unsigned)lane * 16u;
    unsigned char* tile = smem + QK_TILE_OFF + wv * QK_TILE;
    // this wave's share of a slab's DMA: pieces 3 wv .. 3 wv + 2
    auto dma_slab = [&](int s, int st) {
        const int sl = s < nslab ? s : nslab - 1;  // past the end: the last slab again (never read)
        t4_dma_group<3>(gw + (int64_t)sl * T4_SLAB + 3072 * wv, voff0, smem + st * T4_SLAB + 3072 * wv);
    };
    dma_slab(0, 0);
    dma_slab(1, 1);
    dma_slab(2, 2);
    for (int i = tid; i < N; i += 512) bss[i] = bias[i];
    // x^T fragments straight from memory, natural k order: lane (token r31, g) holds features 16 ks + 8 g .. + 7 (read ONCE per kernel)
    half8 xf[ML_KS];
    {
        const bool valid = r31 < rows_valid;
        const _Float16* xr = (const _Float16*)x + (tok0 + (valid ? r31 : 0)) * ML_H + 8 * g;
        const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < ML_KS; ++ks) {
            const half8 v = valid && tok0 < T ? *(const half8*)(xr + 16 * ks) : z;
            xf[ks] = v;
        }
    }
    QkAddr ad;
#pragma unroll
    for (int k7 = 0; k7 < 8; ++k7) {
        const int a = r31 * 768 + ((((2 * k7 + g) ^ r31) & 15) << 4);
        ad.a1[0][k7] = smem + a;
        ad.a1[1][k7] = smem + 2 * T4_SLAB + a;
    }
    T4_WAIT_VM(0);
    __syncthreads();  // bias written, slabs 0..2 landed (nothing is in flight: a plain barrier is fine here)
    const float* bl = bss + 4 * g;
    float16v acc[2][2];
    QkCarry<RD> cy;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4v bv = *(const float4v*)(bl + 8 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) cy.biasv[4 * q + i] = bv[i];
    }
#pragma unroll
    for (int i = 0; i < RD; ++i) cy.ring[i] = *(const half8*)(ad.a1[0][i & 7] + 256 * (i >> 3));  // slab 0 = stage 0, k-steps 0 .. RD - 1
    // slab s: stage s % 4, parity s & 1; its DMA is slab s + 3 -> stage (s + 3) % 4 (last read during slab s - 1)
#define QK_SLAB(ST, P, NEXT, PREV, S)                                                                                                          \
    qk_slots<ST, P, NEXT, PREV, NEXT, RD>(std::make_integer_sequence<int, 24>{}, ad, bl + 32 * ((S) + 1), xf, acc, cy, tile, out, tok0, rows_valid, Nst, \
                                          32 * ((S) - 2), r31, g, lane, gw + (int64_t)((S) + 3 < nslab ? (S) + 3 : nslab - 1) * T4_SLAB + 3072 * wv, voff0, \
                                          smem + (((ST) + 3) % QK_STAGES) * T4_SLAB + 3072 * wv)
    // (the epilogue of an ODD previous slab stores features [32 (S - 2), 32 S): feat0_prev = 32 (S - 2) of the pair it completes)
    QK_SLAB(0, 0, true, false, 0);
    for (int s = 1; s + 3 < nslab; s += 4) {
        QK_SLAB(1, 1, true, true, s);
        QK_SLAB(2, 0, true, true, s + 1);
        QK_SLAB(3, 1, true, true, s + 2);
        QK_SLAB(0, 0, true, true, s + 3);
    }
    {
        const int s = nslab - 3;
        QK_SLAB(1, 1, true, true, s);
        QK_SLAB(2, 0, true, true, s + 1);
        // the last slab: nothing follows it in the ring; its barrier-free
        qk_slots<3, 1, false, true, false, RD>(std::make_integer_sequence<int, 24>{}, ad, nullptr, xf, acc, cy, tile, out, tok0, rows_valid, Nst, 32 * (s + 2 - 2),
                                               r31, g, lane, nullptr, voff0, nullptr);
    }
#undef QK_SLAB
    // epilogue of the last slab (odd: completes the last 64-feature tile)
    {
        const int f0 = 32 * (nslab - 2);
        qk_epilogue_step<0, true>(acc[1][0], acc[1][1], tile, out, tok0, rows_valid, Nst, f0, r31, g, lane);
        qk_epilogue_step<1, true>(acc[1][0], acc[1][1], tile, out, tok0, rows_valid, Nst, f0, r31, g, lane);
        qk_epilogue_step<2, true>(acc[1][0], acc[1][1], tile, out, tok0, rows_valid, Nst, f0, r31, g, lane);
        qk_epilogue_step<3, true>(acc[1][0], acc[1][1], tile, out, tok0, rows_valid, Nst, f0, r31, g, lane);
        qk_epilogue_step<4, true>(acc[1][0], acc[1][1], tile, out, tok0, rows_valid, Nst, f0, r31, g, lane);
        qk_epilogue_step<5, true>(acc[1][0], acc[1][1], tile, out, tok0, rows_valid, Nst, f0, r31, g, lane);
        qk_epilogue_step<6, true>(acc[1][0], acc[1][1], tile, out, tok0, rows_valid, Nst, f0, r31, g, lane);
    }
    T4_WAIT_VM(0);  // the requests past the end (re-reads of the last slab) land before the workgroup's LDS is handed on
}

}  // namespace lm

// (lm_qkv_pack_h384 -- W [n_out][384] -> the image this kernel streams -- lives next to the image kernel: lm_layer_tail_h384.hip)

// head_major != 0: the output as [n_out / 32 heads][tokens][32] (see the kernel's epilogue) -- the one-call forward's layout for the attention kernel
int lm_qkv_h384_launch(const void* d_x, const void* d_w_img, const float* d_bias, int32_t n_out, void* d_out, int64_t tokens, int32_t head_major, void* stream) {
    using namespace lm;
    if (tokens == 0) return LM_OK;
    if (!d_x || !d_w_img || !d_bias || !d_out || tokens < 0 || tokens > 0x7fffffff) LM_FAIL(LM_EINVAL, "bad linear arguments");
    if (n_out < 256 || n_out % 128 || n_out > 6144) LM_FAIL(LM_EINVAL, "lm_qkv_h384_f16: n_out must be a multiple of 128 in [256, 6144]");
    const size_t shmem = (size_t)QK_BIAS_OFF + (size_t)n_out * 4;
    KtScope kt(LM_KT_QKV, stream, 2.0 * (double)tokens * n_out * ML_H);
    static DynLdsAttr attr;
    LM_HIP(ensure_dyn_lds(attr, (const void*)k_qkv_h384<4>, shmem));
    hipLaunchKernelGGL(k_qkv_h384<4>, dim3((unsigned)((tokens + 255) / 256)), dim3(512), shmem, (hipStream_t)stream, (const __half*)d_x, (const __half*)d_w_img,
                       d_bias, (__half*)d_out, (int)tokens, n_out, (int)(head_major != 0));
    LM_HIP(hipGetLastError());
    return LM_OK;
}

extern "C" int lm_qkv_h384_f16(const void* d_x, const void* d_w_img, const float* d_bias, int32_t n_out, void* d_out, int64_t tokens, void* stream) {
    return lm_qkv_h384_launch(d_x, d_w_img, d_bias, n_out, d_out, tokens, 0, stream);
}
