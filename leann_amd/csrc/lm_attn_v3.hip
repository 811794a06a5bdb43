// lm_attn_v3.hip -- fused self-attention for packed variable-length sequences, fp16 in / out, head_dim 32 (MiniLM-L6 / bge-small:
// hidden 384 = 12 x 32, lengths <= 256): generation 3 of the head_dim-32 kernel (round 5).  Head_dim 64 stays on lm_attn_v2.hip.
// Role in the reference: part of compute_embeddings' BERT forward (leann/embedding_compute.py:229-239).
//
// What the counters said about generation 2 (profiles/r5_pmc_sq_attention_v2_pass_*.json, 262k tokens, 305 us): matrix pipe busy 14 %
// of the time, VALU ~50 %, 2.4 waves resident per SIMD on average (waves whose query blocks are done leave, the workgroup's LDS stays),
// 969 VALU instructions per wave at 4.75 cycles each -- and scripts/vopbench.cpp (profiles/r5_vopbench_instruction_costs.jsonl) what
// they cost: ONE wave issues a VALU instruction every ~7 cycles whatever it is, a SIMD needs four resident waves to reach the pipe's
// own rates (v_fma_f32 2.1, v_max3 / v_cvt_pk / v_pk_* 3.1, v_exp_f32 4.5 cycles), v_cndmask on VCC costs ~10.  So the kernel was bound
// by the NUMBER of VALU instructions at low occupancy, and its HBM floor (604 MB of Q / K / V in, 201 MB out: ~140 us at the measured
// copy rates) sits at half of its time.  Generation 3 is that kernel on a diet:
//   * K and V rows go global -> LDS by DMA (global_load_lds_dwordx4, 64 B per key row, 16 rows per wave instruction): no staging
//     registers, no LDS store instructions, no transposing writes.  K's 16-byte chunks are XOR-swizzled by the SOURCE offsets (chunk
//     c of row r at position c ^ ((r >> 2) & 3)): conflict-free ds_read_b128 fragments.  V stays row-major and is read TRANSPOSED by
//     ds_read_b64_tr_b16 (gfx950: a 16-lane group reads a [4 keys][16 columns] block and every lane receives one column's four
//     keys -- lane mapping measured by vopbench's probe): the V^T operand of O^T = V^T P^T with two reads per k-step;
//   * the softmax scale (1 / sqrt(32) x log2 e) is folded into Q once per query block (one rounding: f16(q x c)), and the running
//     row maximum enters the score MFMA as its C operand: a 16-register tuple holds -m, S' = K Q'^T + (-m) leaves the matrix pipe
//     as the argument of exp2 -- no multiply / subtract per score, and no zeroing of an accumulator per tile either;
//   * the maximum is DEFERRED (cdna_hip_programming.md T13): a tile whose scores stay below m + 8 (in log2 units) keeps m, so the
//     common path never rescales O; the rescale path (first tile of a query block, and tiles that exceed the threshold -- a
//     wave-uniform branch) scales O and the row sum, and shifts the tile's scores, the NEXT tile's scores and the -m tuple;
//   * keys past the sequence end exist in the last tile only; they are pushed to -1e30 with one v_med3 per score against a per-lane
//     limit computed arithmetically (no compare / select pairs: v_cndmask on VCC is the most expensive instruction of the set);
//   * the row maximum is a TREE of v_max3 (fmaxf: see a3_max3 below for why not asm), the row sum runs in two independent packed
//     chains, the rescale and the epilogue are packed-f32;
//   * both query blocks of a wave are requested from global memory before the first is used; lane halves are exchanged by
//     v_permlane32_swap instead of ds_bpermute.
// ONE score tuple: issuing tile t + 1's score MFMAs ahead of tile t's softmax (same registers: 278 vs 271 us; a second tuple: 288 us
// and a wave per SIMD less) did not pay -- the variants measured and dropped are recorded where they would have lived, below.
// Per 32 x 32 score tile and wave: 4 MFMAs and ~55 VALU instructions (16 v_exp_f32, 6 v_max3, 8 v_cvt_pk, 8 v_pk_add_f32, ...)
// against ~110 in generation 2; a wave executes 889 VALU instructions against 969 (the per-block prologue / epilogue does not shrink).
// LDS per workgroup 64 B x 2 x padded length = 32 KB at 256 tokens: four workgroups per CU.  DESIGN.md 6.1a has the measurement record.
#include <hip/hip_fp16.h>

#include <algorithm>
#include <cstdlib>

#include "lm_h384_stream.h"

namespace lm {

typedef _Float16 half2v __attribute__((ext_vector_type(2)));

// ds_read_b64_tr_b16: four halfwords, transposed inside a group of 16 lanes (header comment)
__device__ __forceinline__ half4 a3_lds_read_tr16(const unsigned char* p) {
#ifdef LM_EMULATED_DEVICE
    return emul::ds_read_tr16_b64<half4>(p);
#else
    typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
    typedef __attribute__((address_space(3))) fp16x4 lds_fp16x4;
    return __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4*)p));
#endif
}

#if defined(LM_DIAG) && !defined(LM_EMULATED_DEVICE)
// diagnosis library only (scripts/build_kbench.sh; kbench `a3stamps`): VAR 2 = issue order 0 with s_memtime stamps, per-wave phase sums
constexpr int A3_STAMP_WAVES = 1 << 17, A3_STAMP_WORDS = 12;
__device__ unsigned long long g_a3_stamps[A3_STAMP_WAVES * A3_STAMP_WORDS];  // one record per wave (a shared counter serialises the launch: ~12 ns per atomic)
#define A3_STAMPS 1
#define A3_NOW() ((VAR & 4) ? __builtin_amdgcn_s_memtime() : 0ull)
#else
#define A3_STAMPS 0
#define A3_NOW() 0ull
#endif

// Row maximum helpers.  NOT inline asm: an `asm("v_max3_f32 ...")` on the score MFMA's result registers is invisible to the compiler's hazard
// recogniser -- a VALU read of an 8-pass MFMA's destination needs 11 wait states, the compiler inserts the s_nop for instructions it knows and
// none for an asm block -- and the tree then reads registers the matrix pipe has not written yet.  Round 5's last-but-one build did exactly
// that (to save the four v_max_f32 x, x, x canonicalisations per tile that fmaxf costs): the kernel passed every tolerance test (a stale
// maximum only moves the deferred-rescale reference) but was not bit-reproducible from launch to launch; the full GPU suite's
// "same bits from both launch paths" tests caught it (profiles/r5_session17_pytest_gpu_9_failed_asm_max_hazard.log).  Final: 262-274 us (session 19).
__device__ __forceinline__ float a3_max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ float a3_max(float a, float b) { return __builtin_fmaxf(a, b); }

constexpr float A3_THR = 8.0f;  // a tile may exceed the running maximum by 2^8 before O is rescaled (P <= 256: exact in fp16's range)

// NT = number of 32-key tiles the launch's longest sequence needs (max_len <= 32 NT), 1..8.  VAR: 0 = the kernel; 4 = the same with phase stamps
// (diagnosis library only: kbench a3stamps)
template <int NT, int VAR>
__global__ __launch_bounds__(256, 4) void k_attn_varlen_hd32_v3(const __half* __restrict__ qkv, const int32_t* __restrict__ cu,
                                                                __half* __restrict__ out, int heads, float scale_log2e, int n_units, int64_t hm_tokens) {
    extern __shared__ __align__(16) unsigned char smem[];
    // XCD-aware unit order (as generation 2): the twelve heads of a sequence run on one XCD at about the same time, so the two 64-byte
    // halves of a 128-byte line of the [T][3H] activations (neighbouring heads) meet in one L2.  n_units < 0: plain order (A/B).
    const int per = gridDim.x >> 3;
    const int unit = n_units < 0 ? (int)blockIdx.x : (int)((blockIdx.x & 7) * per + (blockIdx.x >> 3));
    if (unit >= (n_units < 0 ? -n_units : n_units)) return;
    const int seq = unit / heads, h = unit - seq * heads;
    const int tok0 = cu[seq];
    const int len = cu[seq + 1] - tok0;
    if (len <= 0) return;
    [[maybe_unused]] const unsigned long long ts0 = A3_NOW();
    const int H = heads * 32;
    // qkv layout: hm_tokens == 0: [tokens][3 H] (a head's Q / K / V row = 64 B out of every 6 H); hm_tokens > 0: head major
    // [3 x heads][hm_tokens][32] (lm_qkv_h384_launch): a (sequence, head)'s rows are one contiguous block per operand
    const int rsb = hm_tokens ? 64 : 6 * H;  // bytes from one token's row to the next
    constexpr int Tp = 32 * NT;
    unsigned char* Ks = smem;            // [Tp][64 B], 16-byte chunks swizzled
    unsigned char* Vs = smem + Tp * 64;  // [Tp][64 B], row major
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r31 = lane & 31, g = lane >> 5;
    const int nt = (len + 31) >> 5;  // key tiles = query blocks of this sequence
    const int64_t plane = hm_tokens ? (int64_t)heads * hm_tokens * 64 : (int64_t)2 * H;  // bytes from Q to K to V of the same token and head
    const unsigned char* base = (const unsigned char*)qkv + (hm_tokens ? ((int64_t)h * hm_tokens + tok0) * 64 : (int64_t)tok0 * rsb + h * 64);

    // ---- K and V rows -> LDS by DMA: instruction j covers rows 16 j .. 16 j + 15 (lane l: row 16 j + l / 4, chunk position l % 4);
    //      rows past the sequence end repeat the last row (finite values; their scores are masked, their P is exactly 0).
    //      (Measured equal and dropped: staging through registers -- global_load_dwordx4 x NT per thread, then ds_write_b128; the stamps
    //      show requests issued in 2.5 k instead of 4.9 k cycles and landed 1.9 k later: the same ~7.3 k from entry to the barrier.)
    {
        const unsigned char* kbase = base + plane;      // K of this head
        const unsigned char* vbase = base + 2 * plane;  // V of this head
        const int rl = lane >> 2, pos = lane & 3;
#pragma unroll
        for (int i = 0; i < (NT + 1) / 2; ++i) {
            const int j = wv + 4 * i;
            if (j < 2 * nt) {
                const int row = 16 * j + rl;
                const int rowc = row < len ? row : len - 1;
                const unsigned ro = (unsigned)(rowc * rsb);
                lm_dma16_sv(kbase, ro + (unsigned)((pos ^ ((row >> 2) & 3)) << 4), Ks + 1024 * j);
                lm_dma16_sv(vbase, ro + (unsigned)(pos << 4), Vs + 1024 * j);
            }
        }
    }
    // ---- both query blocks of this wave (blocks wv and wv + 4), requested now: lane (q row r31, g) holds head dims 16 ks + 8 g .. + 8 ----
    half8 qraw[2][2];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        const int qrow = 32 * (wv + 4 * qi) + r31;
        const unsigned char* qp = base + (int64_t)(qrow < len ? qrow : len - 1) * rsb + g * 16;  // (clamp + select: no load under a lane mask)
        const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        const half8 q0 = *(const half8*)qp, q1 = *(const half8*)(qp + 32);
        qraw[qi][0] = qrow < len ? q0 : z;
        qraw[qi][1] = qrow < len ? q1 : z;
    }
    [[maybe_unused]] const unsigned long long ts1 = A3_NOW();
    T4_WAIT_VM(0);  // this wave's DMA pieces have landed ...
    [[maybe_unused]] const unsigned long long ts2 = A3_NOW();
    __syncthreads();  // ... and everybody else's
    [[maybe_unused]] const unsigned long long ts3 = A3_NOW();
    [[maybe_unused]] unsigned long long tacc_setup = 0, tacc_loop = 0, tacc_epi = 0, nqb_done = 0;

    // per-lane LDS offsets: K fragment (row r31 of a tile, chunk 2 ks + g at its swizzled position), V transposing read (16-lane group:
    // rows 4 g + (lane % 16) / 4 of an 8-key half step, 8-byte piece (lane % 4) of column half (lane / 16) % 2)
    const int sw = (r31 >> 2) & 3;
    const unsigned char* kf0 = Ks + r31 * 64 + ((g ^ sw) << 4);
    const unsigned char* kf1 = Ks + r31 * 64 + (((2 + g) ^ sw) << 4);
    const unsigned char* vf = Vs + g * 256 + ((lane >> 2) & 3) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
    // masking of the last tile: key offset kc = (r & 3) + 8 (r >> 2) (+ 4 g) is valid iff kc < lenv
    const int tail = len & 31;
    const float lenv = (float)(tail - 4 * g);

#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
        const int qb = wv + 4 * qi;
        if (qb >= nt) break;
        const int qrow = 32 * qb + r31;
        [[maybe_unused]] const unsigned long long tq0 = A3_NOW();
        // Q x (softmax scale x log2 e), ONE rounding: f16(float(q) x c).  (Measured in round 5's last session: the same in packed fp16 with the constant
        // split into two halfs -- fl(q c_hi), then fma(q, c_lo, .): 16 packed instructions instead of ~45 -- makes the kernel 3 % faster and its error
        // 1.7 x larger (two roundings of the score MFMA's operand: 1.0e-3 instead of 5.8e-4 against fp64 on kbench's data); not taken.)
        half8 qf0, qf1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            qf0[e] = (_Float16)((float)qraw[qi][0][e] * scale_log2e);
            qf1[e] = (_Float16)((float)qraw[qi][1][e] * scale_log2e);
        }
        float16v cm, o;  // cm = -m (running reference of the row, log2 units) in every register: the C operand of the score MFMAs
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            cm[r] = 0.f;
            o[r] = 0.f;
        }
        float2v l2 = {0.f, 0.f}, l2b = {0.f, 0.f};  // row sum of this lane's keys (four partial sums: two independent accumulation chains)

        // one tile of the online softmax: keys 32 t .. 32 t + 31 against the wave's 32 query rows.  s holds the tile's scores S'^T (keys x q:
        // lane (q = r31, g), register r <-> key 32 t + (r & 3) + 8 (r >> 2) + 4 g).  (Measured and dropped in round 5: the next tile's score MFMAs
        // issued behind this tile's exponentials into the same registers -- 278 vs 271 us; two score tuples -- 288 us, a wave per SIMD less.)
        // (Measured and dropped as well: row sums on the matrix pipe -- l^T += 1 P^T with an all-ones A operand instead of eight v_pk_add_f32 per
        // tile: 284-297 vs 270-283 us.)
        float16v s;
        {
            const half8 k0 = *(const half8*)kf0, k1 = *(const half8*)kf1;
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, qf0, cm, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, qf1, s, 0, 0, 0);
        }
        [[maybe_unused]] const unsigned long long tq1 = A3_NOW();
        for (int t = 0; t < nt; ++t) {
            const bool more = t + 1 < nt;
            if (t > 0) {
                const half8 k0 = *(const half8*)(kf0 + 2048 * t), k1 = *(const half8*)(kf1 + 2048 * t);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, qf0, cm, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, qf1, s, 0, 0, 0);
            }
            // V^T fragments of this tile: k-step u, slots 0..3 = keys 16 u + 4 g + {0..3}, slots 4..7 = keys 16 u + 8 + 4 g + {0..3}
            const unsigned char* vt = vf + 2048 * t;
            const half4 va0 = a3_lds_read_tr16(vt), vb0 = a3_lds_read_tr16(vt + 512), va1 = a3_lds_read_tr16(vt + 1024), vb1 = a3_lds_read_tr16(vt + 1536);
            if (!more && tail) {  // keys past the sequence end (last tile only): s = min(s, (lenv - kc - 0.5) x 1e30), kc = the register's key offset
                float lb = (lenv - 0.5f) * 1.0e30f;
                LM_KEEP_LOCAL(lb);  // (the sixteen limits are recomputed here: hoisted out of the loops they would hold sixteen registers)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_fmed3f(s[r], lb - (float)((r & 3) + 8 * (r >> 2)) * 1.0e30f, -3.0e38f);
            }
            // row maximum as a TREE of v_max3 (depth 3, not a chain of eight): a wave alone issues one VALU instruction per ~7 cycles when they are
            // independent and waits ~12 per link of a dependent chain -- the stamps put a tile at 733 cycles for ~62 instructions
            float tm;
            {
                const float m0 = a3_max3(s[0], s[1], s[2]), m1 = a3_max3(s[3], s[4], s[5]), m2 = a3_max3(s[6], s[7], s[8]);
                const float m3 = a3_max3(s[9], s[10], s[11]), m4 = a3_max3(s[12], s[13], s[14]);
                tm = a3_max(a3_max3(m0, m1, m2), a3_max3(m3, m4, s[15]));
                uint32_t a = __builtin_bit_cast(uint32_t, tm), b = a;
                lane32_swap(a, b);
                tm = a3_max(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));  // both lanes of the row: its maximum over the tile's 32 keys
            }
            const bool first = t == 0;
            if (first || __ballot(tm > A3_THR) != 0) {  // wave uniform
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
                for (int r = 0; r < 16; r += 2) {  // (v_pk_add_f32)
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
            // O^T += V^T P^T : A = V^T (m = d), B = P^T (n = q); k-slots <-> the keys the lane's P registers belong to
            const half8 v0 = {va0[0], va0[1], va0[2], va0[3], vb0[0], vb0[1], vb0[2], vb0[3]};
            const half8 v1 = {va1[0], va1[1], va1[2], va1[3], vb1[0], vb1[1], vb1[2], vb1[3]};
            o = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, p0, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1, p1, o, 0, 0, 0);
        }
        [[maybe_unused]] const unsigned long long tq2 = A3_NOW();
        float l = (l2[0] + l2[1]) + (l2b[0] + l2b[1]);
        {
            uint32_t a = __builtin_bit_cast(uint32_t, l), b = a;
            lane32_swap(a, b);
            l = __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
        }
        const float inv = 1.0f / l;
        // lane (q = r31, g) holds d = (r & 3) + 8 (r >> 2) + 4 g
        if (qrow < len) {
            _Float16* orow = (_Float16*)out + (int64_t)(tok0 + qrow) * H + h * 32;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float2v i2 = {inv, inv};
                const float2v a = (float2v){o[4 * r4], o[4 * r4 + 1]} * i2, b = (float2v){o[4 * r4 + 2], o[4 * r4 + 3]} * i2;  // (v_pk_mul_f32)
                const half4 w = {(_Float16)a[0], (_Float16)a[1], (_Float16)b[0], (_Float16)b[1]};
                *(half4*)(orow + 8 * r4 + 4 * g) = w;
            }
        }
        if constexpr (A3_STAMPS && (VAR & 4) != 0) {
            const unsigned long long tq3 = A3_NOW();
            tacc_setup += tq1 - tq0;
            tacc_loop += tq2 - tq1;
            tacc_epi += tq3 - tq2;
            nqb_done += 1;
        }
    }
#if A3_STAMPS
    if constexpr ((VAR & 4) != 0) {
        if (lane == 0) {
            const unsigned long long te = __builtin_amdgcn_s_memtime();
            unsigned long long* rec = g_a3_stamps + (size_t)(((unsigned)blockIdx.x * 4u + (unsigned)wv) & (A3_STAMP_WAVES - 1)) * A3_STAMP_WORDS;
            rec[0] = ts1 - ts0;   // entry -> DMA pieces and Q loads issued
            rec[1] = ts2 - ts1;   // -> own DMA pieces landed
            rec[2] = ts3 - ts2;   // -> barrier passed
            rec[3] = tacc_setup;  // per query block: Q prescale, accumulator set-up, first score MFMAs issued
            rec[4] = tacc_loop;   // tile loops
            rec[5] = tacc_epi;    // normalisation + stores issued
            rec[6] = nqb_done;
            rec[7] = nqb_done * (unsigned long long)nt;  // tiles
            rec[8] = 1ull;
            rec[9] = te - ts0;    // wave lifetime from the first stamp on
            rec[10] = (unsigned long long)len;
            rec[11] = ts0;        // absolute start (launch shape)
        }
    }
#endif
}

// (Measured equal and deleted in round 5's last session: the RING form -- flash-attention's loop order at workgroup level: the 32-key K / V tiles
// stream ONCE per (sequence, head) through a four-stage LDS ring (16 KB), one DMA piece per wave and tile requested three tiles ahead, one counted
// vmcnt + one barrier per tile, each wave holding BOTH of its query blocks at once, the running reference entering the scores through a third MFMA
// k-step (v_mfma_f32_32x32x8_f16 with a ones column) instead of a 16-register C tuple: 271-285 us against 272-281 us, whole encoder 107.5 vs 107.1 ms
// (profiles/r5_kbench_attention_v3_ring_form_equal.jsonl).  With it every structural variant of this kernel -- staging by DMA or through registers,
// one or two score tuples, 4 or 5 waves per SIMD, row sums on either pipe, K / V per unit or per tile ring -- lands within box noise of 0.31 ns x
// the VALU instructions a wave executes (generation 2: 969, 300 us; generation 3: 889, 275 us): what is left to gain is instruction COUNT.)

}  // namespace lm

#if defined(LM_DIAG) && !defined(LM_HOST_EMULATION) && !defined(LM_EMULATED_DEVICE)
extern "C" int lm_attn_v3_stamps_read(unsigned long long* out, int64_t max_words, int reset) {  // out: [waves][12] records (lm::A3_STAMP_WORDS), zeroed slots = unused
    const size_t total = (size_t)lm::A3_STAMP_WAVES * lm::A3_STAMP_WORDS;
    const size_t n = std::min<size_t>(total, (size_t)max_words);
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(lm::g_a3_stamps), n * sizeof(unsigned long long)) != hipSuccess) return LM_EHIP;
    if (reset) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(lm::g_a3_stamps)) != hipSuccess || hipMemset(p, 0, total * sizeof(unsigned long long)) != hipSuccess) return LM_EHIP;
    }
    return LM_OK;
}
#endif

#ifndef LM_HOST_EMULATION
// launched by lm_attn_v2.hip's attn_v2_launch_hd<32> (the default of head_dim 32 since round 5; LEANN_MI355X_ATTN=2 = generation 2)
int lm_attn_v3_launch_hd32(const void* d_qkv, const int32_t* d_cu_seqlens, int32_t n_seqs, int32_t heads, int32_t max_len, void* d_out, int64_t total_tokens,
                           void* stream) {
    using namespace lm;
    const int nt = (max_len + 31) / 32;
    const size_t shmem = (size_t)32 * nt * 128;
    const float scale_log2e = 1.4426950408889634f / sqrtf(32.0f);
    const char* xo = getenv("LEANN_MI355X_ATTN_XCD");
    const int n_units = (xo && xo[0] == '0') ? -(n_seqs * heads) : n_seqs * heads;
    dim3 grid((unsigned)((n_seqs * heads + 7) / 8 * 8)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const __half* q = (const __half*)d_qkv;
    __half* o = (__half*)d_out;
    kt_attn_work(d_cu_seqlens, n_seqs, heads * 32, stream);  // the flops depend on the sequence lengths (device memory): summed there
    KtScope kt(LM_KT_ATTN, stream, 0.0);
    int var = 0;
#if A3_STAMPS
    const char* ve = getenv("LEANN_MI355X_ATTN3");
    if (ve && ve[0] == '4') var = 4;
#define A3_CASE4(n) CASEV(n, 4);
#else
#define A3_CASE4(n)
#endif
    switch (nt * 8 + var) {
#define CASEV(n, v) \
    case n * 8 + v: hipLaunchKernelGGL((k_attn_varlen_hd32_v3<n, v>), grid, block, shmem, st, q, d_cu_seqlens, o, heads, scale_log2e, n_units, total_tokens); break
#define CASEA(n) CASEV(n, 0); A3_CASE4(n)
        CASEA(1); CASEA(2); CASEA(3); CASEA(4); CASEA(5); CASEA(6); CASEA(7); CASEA(8);
#undef CASEA
#undef CASEV
#undef A3_CASE4
        default: LM_FAIL(LM_EINVAL, "lm_attn_varlen_hd32_f16 supports sequence lengths 1..256");
    }
    LM_HIP(hipGetLastError());
    return LM_OK;
}
#endif  // LM_HOST_EMULATION
