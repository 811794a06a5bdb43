// lm_qkv_attn_h384.hip -- the QKV projection FUSED INTO self-attention for hidden 384 = 12 heads x 32 (MiniLM-L6, bge-small), packed
// variable-length sequences of 1..256 tokens:
//
//     a[T][384] = concat_h softmax(Q_h K_h^T / sqrt(32)) V_h,      [Q_h | K_h | V_h] = x W_h^T + b_h,      x [T][384] fp16
//
// Role in the reference: the first half of every layer of compute_embeddings' BERT forward (leann/embedding_compute.py:229-239).
//
// Why (round 6; VERDICT r5 "next round" item 2).  As two kernels -- lm_qkv_h384.hip, lm_attn_v3.hip -- this half of a
// layer was 39 % of a search step and bound by an intermediate that existed only to be re-read: per 262 k tokens the projection wrote
// 604 MB of Q / K / V that attention read back, against 201 MB of x in and 201 MB of output out (profiles/r4_pmc_qkv_kernels.json,
// r5_pmc_attention_v3_hbm_traffic.json): 1.2 of the pair's 1.6 GB.  A stand-alone attention kernel on that layout cannot pass 0.22 of
// the MFMA peak (its HBM floor) and eleven variants of generation 3 all landed within noise of each other.  Here Q, K and V never
// leave the CU:
//   * ONE 512-thread workgroup (eight waves, two per SIMD) per sequence; wave w owns the sequence's token rows 32 w .. 32 w + 31: it keeps
//     their x^T fragments in registers for the whole kernel (96 VGPRs, x read from memory ONCE, as in lm_qkv_h384.hip) and is the query
//     block of those rows in every head;
//   * heads in a loop.  Per head three 24 KB weight slabs (the rows of W_q, W_k, W_v of that head: slabs h, 12 + h, 24 + h of the SAME
//     image lm_qkv_pack_h384 writes for the stand-alone kernel) stream L2 -> LDS by DMA through a three-stage ring, stage = kind, each
//     requested one head ahead (the moment its stage falls free) -- the attention phase in between hides the landing;
//   * a slab is 24 MFMAs (32x32x16 f16) per wave in two accumulator chains, bias in the first: out^T = W x^T, so a lane ends up with 16
//     of its token's 32 head features.  Q: x softmax scale x log2 e, ONE rounding to fp16 (the stand-alone pair rounded twice), and it IS
//     the score MFMA's B operand -- no movement at all (the k-slot order of that operand is whatever the accumulator layout gives; K is
//     stored so that its A operand uses the same order).  K -> LDS [256][32] fp16 in the swizzled 16-byte chunks generation 3 reads
//     conflict-free; V -> LDS row major, read TRANSPOSED by ds_read_b64_tr_b16: both exactly the layouts lm_attn_v3.hip stages by DMA;
//   * then generation 3's tile loop unchanged (scores swapped, running maximum as the score MFMA's C operand, deferred rescaling,
//     arithmetic masking of the last tile) over the sequence's key tiles, and the head's 32 output columns leave as 64 B per token row.
// Waves whose rows lie past the sequence's end (lengths are ~N(180, 50): six of eight blocks on average) only move their share of the weight
// stream and meet the barriers.  Four barriers per head: head start (the stage-0 slab has landed; every wave is done with the previous
// head's K / V), one per slab boundary (next slab landed / this stage free), one between the V epilogue and the tile loop.
// Counted waits: a wave's vector-memory operations are, in order, 3 DMA pieces per slab request and 4 output stores per head (active
// waves only).  Every wait below is written as "at most N YOUNGER operations may be outstanding" with N counting DMA pieces only, so that
// it holds with or without the stores (they are the youngest operations whenever a wait is reached, or older than what is waited for).
// LDS: [0, 72 K) ring, [72 K, 88 K) K, [88 K, 104 K) V, then the 1152 biases.
#include <cstdlib>
#include <utility>

#include "lm_h384_stream.h"

namespace lm {

typedef _Float16 qa_half2 __attribute__((ext_vector_type(2)));

constexpr int QA_K_OFF = 3 * T4_SLAB;          // 73728
constexpr int QA_V_OFF = QA_K_OFF + 256 * 64;  // 90112
constexpr int QA_BIAS_OFF = QA_V_OFF + 256 * 64;
constexpr int QA_LDS = QA_BIAS_OFF + 1152 * 4;  // 111104
#ifndef LM_QA_RD
#define LM_QA_RD 8
#endif
constexpr int QA_RD = LM_QA_RD;                 // fragment reads in flight (round 6: 4 -> 8: a wave alone in its slabs -- generation 2 -- has nobody to hide the LDS latency behind)
constexpr float QA_THR = 8.0f;                  // deferred rescaling threshold (lm_attn_v3.hip: A3_THR)

__device__ __forceinline__ half4 qa_lds_read_tr16(const unsigned char* p) {
#ifdef LM_EMULATED_DEVICE
    return emul::ds_read_tr16_b64<half4>(p);
#else
    typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
    typedef __attribute__((address_space(3))) fp16x4 lds_fp16x4;
    return __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4*)p));
#endif
}
__device__ __forceinline__ float qa_max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }  // (never asm on MFMA results: lm_attn_v3.hip)

#if defined(LM_DIAG) && !defined(LM_EMULATED_DEVICE)
// diagnosis library only (kbench `fusedqa` with KBENCH_QA_STAMPS): ABL bit 4 = the kernel with s_memtime stamps, one record of phase sums per wave
constexpr int QA_STAMP_WAVES = 1 << 15, QA_STAMP_WORDS = 16;
__device__ unsigned long long g_qa_stamps[QA_STAMP_WAVES * QA_STAMP_WORDS];
#define QA_NOW() ((ABL & 16) ? __builtin_amdgcn_s_memtime() : 0ull)
#define QA_STAMPS 1
#else
#define QA_NOW() 0ull
#define QA_STAMPS 0
#endif

struct QaAddr {
    const unsigned char* a[2][8];  // W fragment addresses: ring stages 0, 1 / stage 2 (a ds_read offset is 16 bits)
};
template <int ST, int KS>
__device__ __forceinline__ half8 qa_frag(const QaAddr& c) {
    return *(const half8*)(c.a[ST >> 1][KS & 7] + (ST & 1) * T4_SLAB + 256 * (KS >> 3));
}

struct QaCarry {
    half8 ring[QA_RD];
    float16v biasv;  // accumulator register 4 q + i <-> feature 8 q + 4 g + i of the slab
};
__device__ __forceinline__ void qa_load_bias(QaCarry& cy, const float* bl) {  // bl = bias of the slab + 4 g
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4v bv = *(const float4v*)(bl + 8 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) cy.biasv[4 * q + i] = bv[i];
    }
}

// One slot of a slab: MFMA I (even k-steps chain 0 with the bias as C, odd k-steps chain 1), the fragment of slot I + RD, and
//   slot 1        this wave's three DMA pieces of the slab `dsrc` -> the stage that fell free at the last barrier
//   slot 24 - RD  (NEXT) counted wait + barrier: the next slab's stage has landed for every wave, and every wave has issued its last
//                 read of this slab's stage (fragment 23 is read in slot 23 - RD); from here on the fragment ring reads the next stage
//   slots 16..19  (NEXT) the next slab's bias vector
// ABL (diagnosis library only, kbench fusedqa with LEANN_MI355X_QA_ABLATE: timing runs whose RESULTS ARE GARBAGE -- what each ingredient of the slab loop
// costs): bit 0 skip the tile loop, bit 1 skip the slab barriers and their counted waits, bit 2 issue no DMA, bit 3 no fragment reads (MFMAs on stale registers)
template <int ST, bool NEXT, bool DMA, int ABL, int I>
__device__ __forceinline__ void qa_slot(const QaAddr& c, const float* bl_next, const half8 (&xf)[ML_KS], float16v (&acc)[2], QaCarry& cy,
                                        const unsigned char* dsrc, unsigned voff, unsigned char* ddst) {
    if constexpr (I == 24 - QA_RD && NEXT && !(ABL & 2)) {
        t4_wait_vm<3>();  // younger than the awaited slab's pieces: at most one other request (3 pieces) -- and stores, which are then older or youngest
        T4_BARRIER();
    }
    if constexpr (I == 0) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cy.ring[0], xf[0], cy.biasv, 0, 0, 0);
    else if constexpr (I == 1) {
        const float16v z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cy.ring[1 % QA_RD], xf[1], z, 0, 0, 0);
    } else acc[I & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cy.ring[I % QA_RD], xf[I], acc[I & 1], 0, 0, 0);
    if constexpr (ABL & 8) {
    } else if constexpr (I + QA_RD < 24) cy.ring[I % QA_RD] = qa_frag<ST, I + QA_RD>(c);
    else if constexpr (NEXT) cy.ring[I % QA_RD] = qa_frag<(ST + 1) % 3, I + QA_RD - 24>(c);
    if constexpr (NEXT && I >= 16 && I < 20) {
        constexpr int q = I - 16;
        const float4v bv = *(const float4v*)(bl_next + 8 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) cy.biasv[4 * q + i] = bv[i];
    }
    if constexpr (DMA && I == 1 && !(ABL & 4)) t4_dma_group<3>(dsrc, voff, ddst);
    __builtin_amdgcn_sched_barrier(0);
}
template <int ST, bool NEXT, bool DMA, int ABL, int... I>
__device__ __forceinline__ void qa_slab(std::integer_sequence<int, I...>, const QaAddr& c, const float* bl_next, const half8 (&xf)[ML_KS], float16v (&acc)[2],
                                        QaCarry& cy, const unsigned char* dsrc, unsigned voff, unsigned char* ddst) {
    (qa_slot<ST, NEXT, DMA, ABL, I>(c, bl_next, xf, acc, cy, dsrc, voff, ddst), ...);
}

// grid: one workgroup per sequence.  w_img: lm_qkv_pack_h384's image of the nn.Linear weight [1152][384] (rows: W_q | W_k | W_v, head h = rows
// 32 h .. 32 h + 31 of each); bias [1152] fp32; out [T][384] fp16.
template <int ABL>
__global__ __launch_bounds__(512) LM_TWO_WAVES_PER_SIMD void k_qkv_attn_h384(const __half* __restrict__ x, const __half* __restrict__ w_img,
                                                                              const float* __restrict__ bias, const int32_t* __restrict__ cu,
                                                                              __half* __restrict__ out, float scale_log2e) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int H = ML_H, HEADS = 12;
    const int seq = blockIdx.x;
    const int tok0 = cu[seq];
    const int len = cu[seq + 1] - tok0;
    if (len <= 0) return;  // (workgroup uniform)
    [[maybe_unused]] const unsigned long long ts0 = QA_NOW();
    [[maybe_unused]] unsigned long long tp[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float* bss = (float*)(smem + QA_BIAS_OFF);
    unsigned char* Ks = smem + QA_K_OFF;  // [256][64 B], 16-byte chunk c of row r at position c ^ ((r >> 2) & 3)
    unsigned char* Vs = smem + QA_V_OFF;  // [256][64 B], row major
    const int tid = threadIdx.x, lane = tid & 63;
#ifdef LM_EMULATED_DEVICE
    const int wv = tid >> 6;
#else
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    const int r31 = lane & 31, g = lane >> 5;
    const int nt = (len + 31) >> 5;   // 32-row blocks of this sequence = key tiles = active waves
    const bool active = wv < nt;      // wave uniform
    const int row = 32 * wv + r31;    // this lane's token row within the sequence
    const unsigned char* gw = (const unsigned char*)w_img;
    const unsigned voff0 = (unsigned)lane * 16u;
    // this wave's share of a slab's DMA: pieces 3 wv .. 3 wv + 2 of slab `slab` -> stage `st`
    auto dma_slab = [&](int slab, int st) { t4_dma_group<3>(gw + (int64_t)slab * T4_SLAB + 3072 * wv, voff0, smem + st * T4_SLAB + 3072 * wv); };
    dma_slab(0, 0);
    dma_slab(HEADS, 1);
    dma_slab(2 * HEADS, 2);
    for (int i = tid; i < 3 * H; i += 512) bss[i] = bias[i];
    // x^T fragments straight from memory: lane (token r31, g) holds features 16 ks + 8 g .. + 7 of its row (zero past the sequence end)
    half8 xf[ML_KS];
    {
        const bool valid = row < len;
        const _Float16* xr = (const _Float16*)x + (int64_t)(tok0 + (valid ? row : 0)) * H + 8 * g;
        const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < ML_KS; ++ks) {
            const half8 v = *(const half8*)(xr + 16 * ks);
            xf[ks] = valid ? v : z;
        }
    }
    QaAddr ad;
#pragma unroll
    for (int k7 = 0; k7 < 8; ++k7) {
        const int a = r31 * 768 + ((((2 * k7 + g) ^ r31) & 15) << 4);
        ad.a[0][k7] = smem + a;
        ad.a[1][k7] = smem + 2 * T4_SLAB + a;
    }
    // attention-side per-lane LDS offsets (lm_attn_v3.hip): K fragment (row r31 of a tile, chunk 2 ks + g at its swizzled position), V
    // transposing read (16-lane group: rows 4 g + (lane % 16) / 4 of an 8-key half step, 8-byte piece lane % 4 of column half (lane / 16) % 2)
    const int sw = (r31 >> 2) & 3;
    const unsigned char* kf0 = Ks + r31 * 64 + ((g ^ sw) << 4);
    const unsigned char* kf1 = Ks + r31 * 64 + (((2 + g) ^ sw) << 4);
    const unsigned char* vf = Vs + g * 256 + ((lane >> 2) & 3) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
    // where this lane's projected K chunks / V pieces go
    unsigned char* kw0 = Ks + row * 64 + ((g ^ sw) << 4);
    unsigned char* kw1 = Ks + row * 64 + (((2 + g) ^ sw) << 4);
    unsigned char* vw = Vs + row * 64 + 8 * g;
    const int tail = len & 31;
    const float lenv = (float)(tail - 4 * g);
    const float* bl = bss + 4 * g;
    float16v acc[2];
    QaCarry cy;
    [[maybe_unused]] const unsigned long long ts1a = QA_NOW();
    T4_WAIT_VM(0);
    [[maybe_unused]] const unsigned long long ts1b = QA_NOW();
    __syncthreads();  // biases written, the first head's three slabs landed (nothing is in flight: a plain barrier)
    [[maybe_unused]] const unsigned long long ts1 = QA_NOW();

    for (int h = 0; h < HEADS; ++h) {
        const int hn = h + 1 < HEADS ? h + 1 : HEADS - 1;  // the head whose slabs are requested during this one (past the end: the last head's again, never read)
        [[maybe_unused]] unsigned long long t_a = QA_NOW(), t_b;
#define QA_LAP(i) do { if constexpr ((ABL & 16) != 0) { t_b = QA_NOW(); tp[i] += t_b - t_a; t_a = t_b; } } while (0)
        // ---- head start: the Q slab (stage 0, requested during the previous head's K slab) has landed; every wave has left the previous head's
        //      tile loop (K / V and stage 2 are free).  Younger than its pieces: the K and V requests (6 pieces) and this wave's stores. ----
        if (h > 0) {
            t4_wait_vm<6>();
            T4_BARRIER();
        }
        QA_LAP(0);  // head-start wait + barrier
        if (active) {
            qa_load_bias(cy, bl + 32 * h);
#pragma unroll
            for (int i = 0; i < QA_RD; ++i) cy.ring[i] = *(const half8*)(ad.a[0][i & 7] + 256 * (i >> 3));
        }
        half8 qf0, qf1;
        // ---- Q slab (stage 0) ----
        if (active) {
            qa_slab<0, true, false, ABL>(std::make_integer_sequence<int, 24>{}, ad, bl + 32 * (HEADS + h), xf, acc, cy, nullptr, voff0, nullptr);
            QA_LAP(1);  // Q slab (fragment-ring preload, 24 MFMAs, the barrier in front of slot 20)
            // Q x (softmax scale x log2 e): ONE rounding, and the result is the score MFMA's B operand as it stands (k-slot e of k-step ks of lane
            // g <-> head feature 16 ks + 8 (e >> 2) + 4 g + (e & 3); K's chunks below carry the same order)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                qf0[e] = (_Float16)((acc[0][e] + acc[1][e]) * scale_log2e);
                qf1[e] = (_Float16)((acc[0][8 + e] + acc[1][8 + e]) * scale_log2e);
            }
        } else if constexpr (!(ABL & 2)) {
            t4_wait_vm<3>();
            T4_BARRIER();
        }
        QA_LAP(2);  // Q epilogue (inactive waves: their wait + barrier)
        // ---- K slab (stage 1); stage 0 fell free at the barrier above: request the next head's Q slab ----
        if (active) {
            qa_slab<1, true, true, ABL>(std::make_integer_sequence<int, 24>{}, ad, bl + 32 * (2 * HEADS + h), xf, acc, cy, gw + (int64_t)hn * T4_SLAB + 3072 * wv, voff0,
                                   smem + 3072 * wv);
            QA_LAP(3);  // K slab
            half8 k0, k1;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                k0[e] = (_Float16)(acc[0][e] + acc[1][e]);
                k1[e] = (_Float16)(acc[0][8 + e] + acc[1][8 + e]);
            }
            *(half8*)kw0 = k0;
            *(half8*)kw1 = k1;
        } else {
            if constexpr (!(ABL & 4)) dma_slab(hn, 0);
            if constexpr (!(ABL & 2)) {
                t4_wait_vm<3>();
                T4_BARRIER();
            }
        }
        QA_LAP(4);  // K epilogue
        // ---- V slab (stage 2, nothing follows it in the ring); stage 1 fell free: request the next head's K slab ----
        if (active) {
            qa_slab<2, false, true, ABL>(std::make_integer_sequence<int, 24>{}, ad, nullptr, xf, acc, cy, gw + (int64_t)(HEADS + hn) * T4_SLAB + 3072 * wv, voff0,
                                    smem + T4_SLAB + 3072 * wv);
            QA_LAP(5);  // V slab
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                half4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = (_Float16)(acc[0][4 * q + i] + acc[1][4 * q + i]);
                *(half4*)(vw + 16 * q) = v;  // head features 8 q + 4 g .. + 3 of this token: row major
            }
        } else {
            if constexpr (!(ABL & 4)) dma_slab(HEADS + hn, 1);
        }
        T4_WAIT_LGKM0();
        QA_LAP(6);  // V epilogue + LDS writes retired
        T4_BARRIER();  // K and V of every block are in LDS; every wave is done with stage 2
        QA_LAP(7);  // the barrier in front of the tile loop
        if constexpr (!(ABL & 4)) dma_slab(2 * HEADS + hn, 2);
        if (!active) continue;
        if constexpr (ABL & 1) {  // (timing run without the tile loop: the projected Q still has a consumer)
            if (row < len) *(half8*)((_Float16*)out + (int64_t)(tok0 + row) * H + h * 32 + 8 * g) = qf0 + qf1;
            continue;
        }

        // ---- generation 3's tile loop (lm_attn_v3.hip) for this wave's 32 query rows against the sequence's nt key tiles ----
        float16v cm, o;  // cm = -m (running reference of the row, log2 units): the C operand of the score MFMAs
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            cm[r] = 0.f;
            o[r] = 0.f;
        }
        float2v l2 = {0.f, 0.f}, l2b = {0.f, 0.f};
        float16v s;
        for (int t = 0; t < nt; ++t) {
            const bool more = t + 1 < nt;
            {
                const half8 k0 = *(const half8*)(kf0 + 2048 * t), k1 = *(const half8*)(kf1 + 2048 * t);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, qf0, cm, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, qf1, s, 0, 0, 0);
            }
            const unsigned char* vt = vf + 2048 * t;
            const half4 va0 = qa_lds_read_tr16(vt), vb0 = qa_lds_read_tr16(vt + 512), va1 = qa_lds_read_tr16(vt + 1024), vb1 = qa_lds_read_tr16(vt + 1536);
            if (!more && tail) {  // keys past the sequence end (last tile only)
                float lb = (lenv - 0.5f) * 1.0e30f;
                LM_KEEP_LOCAL(lb);
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_fmed3f(s[r], lb - (float)((r & 3) + 8 * (r >> 2)) * 1.0e30f, -3.0e38f);
            }
            float tm;
            {
                const float m0 = qa_max3(s[0], s[1], s[2]), m1 = qa_max3(s[3], s[4], s[5]), m2 = qa_max3(s[6], s[7], s[8]);
                const float m3 = qa_max3(s[9], s[10], s[11]), m4 = qa_max3(s[12], s[13], s[14]);
                tm = __builtin_fmaxf(qa_max3(m0, m1, m2), qa_max3(m3, m4, s[15]));
                uint32_t a = __builtin_bit_cast(uint32_t, tm), b = a;
                lane32_swap(a, b);
                tm = __builtin_fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
            }
            const bool first = t == 0;
            if (first || __ballot(tm > QA_THR) != 0) {  // wave uniform
                const float delta = first ? tm : fmaxf(tm, 0.f);
                if (!first) {
                    const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[r] *= alpha;
                    l2 *= (float2v){alpha, alpha};
                    l2b *= (float2v){alpha, alpha};
                }
                const float2v d2 = {delta, delta};
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float2v a = (float2v){s[r], s[r + 1]} - d2, b = (float2v){cm[r], cm[r + 1]} - d2;
                    s[r] = a[0];
                    s[r + 1] = a[1];
                    cm[r] = b[0];
                    cm[r + 1] = b[1];
                }
            }
            half8 p0, p1;
#pragma unroll
            for (int r = 0; r < 8; r += 2) {
                const float2v e0 = {__builtin_amdgcn_exp2f(s[r]), __builtin_amdgcn_exp2f(s[r + 1])};
                const float2v e1 = {__builtin_amdgcn_exp2f(s[8 + r]), __builtin_amdgcn_exp2f(s[9 + r])};
                l2 += e0;
                l2b += e1;
                p0[r] = (_Float16)e0[0];
                p0[r + 1] = (_Float16)e0[1];
                p1[r] = (_Float16)e1[0];
                p1[r + 1] = (_Float16)e1[1];
            }
            const half8 v0 = {va0[0], va0[1], va0[2], va0[3], vb0[0], vb0[1], vb0[2], vb0[3]};
            const half8 v1 = {va1[0], va1[1], va1[2], va1[3], vb1[0], vb1[1], vb1[2], vb1[3]};
            o = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, p0, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1, p1, o, 0, 0, 0);
        }
        QA_LAP(8);  // tile loop
        float l = (l2[0] + l2[1]) + (l2b[0] + l2b[1]);
        {
            uint32_t a = __builtin_bit_cast(uint32_t, l), b = a;
            lane32_swap(a, b);
            l = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
        }
        const float inv = 1.0f / l;
        if (row < len) {  // lane (q = r31, g) holds d = (r & 3) + 8 (r >> 2) + 4 g
            _Float16* orow = (_Float16*)out + (int64_t)(tok0 + row) * H + h * 32;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float2v i2 = {inv, inv};
                const float2v a = (float2v){o[4 * r4], o[4 * r4 + 1]} * i2, b = (float2v){o[4 * r4 + 2], o[4 * r4 + 3]} * i2;
                const half4 w = {(_Float16)a[0], (_Float16)a[1], (_Float16)b[0], (_Float16)b[1]};
                *(half4*)(orow + 8 * r4 + 4 * g) = w;
            }
        }
        QA_LAP(9);  // normalisation + output stores issued
    }
#undef QA_LAP
#if QA_STAMPS
    if constexpr ((ABL & 16) != 0) {
        if (lane == 0) {
            const unsigned long long te = __builtin_amdgcn_s_memtime();
            unsigned long long* rec = g_qa_stamps + (size_t)(((unsigned)blockIdx.x * 8u + (unsigned)wv) & (QA_STAMP_WAVES - 1)) * QA_STAMP_WORDS;
            rec[0] = ts1a - ts0;  // entry -> requests issued (x loads, first three slabs, bias copy)
            rec[1] = ts1b - ts1a; // -> own loads / DMA pieces landed
            rec[2] = ts1 - ts1b;  // -> prologue barrier passed
            for (int i = 0; i < 10; ++i) rec[3 + i] = tp[i];
            rec[13] = te - ts0;   // lifetime
            rec[14] = (unsigned long long)len;
            rec[15] = active ? 1ull : 2ull;
        }
    }
#endif
    T4_WAIT_VM(0);  // the requests past the end (re-reads of the last head's slabs) land before the workgroup's LDS is handed on
}


// (Measured and deleted in round 6: GENERATION 2 of this kernel -- the two waves of a SIMD half a head apart: waves 0-3 run head h's tile loop while waves 4-7
// project head h + 1, then the roles swap; K / V double buffered by head parity, three workgroup barriers per slot, the attending group's tile loop cut in three
// parts to meet the projecting group's slab boundaries; bit-identical results.  The idea: one MFMA-bound and one VALU-bound wave per SIMD at any time.  On the
// chip: 439-446 us against 425-436 us for this kernel at length 256, 526-553 against 492-536 at N(180, 50) lengths.  Stamps per wave, slot part and group
// (profiles/r6_kbench_fused_qkv_attention_gen2_stamps_and_ring_depth.jsonl): beside a projecting partner a tile takes ~990 cycles -- against 590 for a wave
// alone on its SIMD and 765 beside a partner that is in its tile loop too -- whichever of the two waves is older, and s_setprio 2 / 3 on the attending wave moves it
// by 7 % at most (r6_kbench_fused_qkv_attention_gen2_priority_ab_lost.jsonl): the projecting group then idles half of every slot at the barriers.  MFMA-dense and
// VALU-dense streams of two waves do NOT add up on one SIMD of this chip the way the instruction mix suggests; MI355X_MICROARCH.md's "moving work between the two
// waves of a SIMD is zero- or negative-sum" held.)

}  // namespace lm

#if defined(LM_DIAG) && !defined(LM_HOST_EMULATION) && !defined(LM_EMULATED_DEVICE)
extern "C" int lm_qa_stamps_read(unsigned long long* out, int64_t max_words, int reset) {  // out: [waves][16] records (lm::QA_STAMP_WORDS); zeroed slots = unused
    const size_t total = (size_t)lm::QA_STAMP_WAVES * lm::QA_STAMP_WORDS;
    const size_t n = total < (size_t)max_words ? total : (size_t)max_words;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(lm::g_qa_stamps), n * sizeof(unsigned long long)) != hipSuccess) return LM_EHIP;
    if (reset) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(lm::g_qa_stamps)) != hipSuccess || hipMemset(p, 0, total * sizeof(unsigned long long)) != hipSuccess) return LM_EHIP;
    }
    return LM_OK;
}
#endif

#ifndef LM_HOST_EMULATION
int lm_qkv_attn_h384_launch(const void* d_x, const void* d_wqkv_img, const float* d_bqkv, const int32_t* d_cu_seqlens, int32_t n_seqs, int32_t max_len,
                            int64_t total_tokens, void* d_out, void* stream) {
    using namespace lm;
    if (n_seqs == 0 || total_tokens == 0) return LM_OK;
    if (!d_x || !d_wqkv_img || !d_bqkv || !d_cu_seqlens || !d_out || n_seqs < 0 || total_tokens < 0) LM_FAIL(LM_EINVAL, "lm_qkv_attn_h384_f16: bad arguments");
    if (max_len <= 0 || max_len > 256) LM_FAIL(LM_EINVAL, "lm_qkv_attn_h384_f16 supports sequence lengths 1..256");
    const float scale_log2e = 1.4426950408889634f / sqrtf(32.0f);
    kt_attn_work(d_cu_seqlens, n_seqs, ML_H, stream, LM_KT_QKV_ATTN);  // attention's flops depend on the lengths (device memory): summed there, onto THIS kernel's slot
    KtScope kt(LM_KT_QKV_ATTN, stream, 2.0 * (double)total_tokens * 3 * ML_H * ML_H);
#if defined(LM_DIAG)
    if (const char* ab = getenv("LEANN_MI355X_QA_ABLATE")) {  // diagnosis library only: timing runs, garbage results (see qa_slot)
        const int v = atoi(ab);
#define QA_ABL(n)                                                                                                                                          \
    if (v == n) {                                                                                                                                          \
        static DynLdsAttr a##n;                                                                                                                            \
        LM_HIP(ensure_dyn_lds(a##n, (const void*)k_qkv_attn_h384<n>, (size_t)QA_LDS));                                                                     \
        hipLaunchKernelGGL(k_qkv_attn_h384<n>, dim3((unsigned)n_seqs), dim3(512), (size_t)QA_LDS, (hipStream_t)stream, (const __half*)d_x,                 \
                           (const __half*)d_wqkv_img, d_bqkv, d_cu_seqlens, (__half*)d_out, scale_log2e);                                                  \
        LM_HIP(hipGetLastError());                                                                                                                         \
        return LM_OK;                                                                                                                                      \
    }
        QA_ABL(1) QA_ABL(3) QA_ABL(7) QA_ABL(15) QA_ABL(2) QA_ABL(9) QA_ABL(16)
#undef QA_ABL
    }
#endif
    static DynLdsAttr attr;
    LM_HIP(ensure_dyn_lds(attr, (const void*)k_qkv_attn_h384<0>, (size_t)QA_LDS));
    hipLaunchKernelGGL(k_qkv_attn_h384<0>, dim3((unsigned)n_seqs), dim3(512), (size_t)QA_LDS, (hipStream_t)stream, (const __half*)d_x, (const __half*)d_wqkv_img,
                       d_bqkv, d_cu_seqlens, (__half*)d_out, scale_log2e);
    LM_HIP(hipGetLastError());
    return LM_OK;
}

extern "C" int lm_qkv_attn_h384_f16(const void* d_x, const void* d_wqkv_img, const float* d_bqkv, const int32_t* d_cu_seqlens, int32_t n_seqs, int32_t max_len,
                                    int64_t total_tokens, void* d_out, void* stream) {
    return lm_qkv_attn_h384_launch(d_x, d_wqkv_img, d_bqkv, d_cu_seqlens, n_seqs, max_len, total_tokens, d_out, stream);
}
#endif
