// lm_gemm_h384.hip -- linear layers with 384 input features, second generation ("64 tokens per wave"):
//
//   MODE 0   out[T][N] = x W^T + b                          N = 384 P   (QKV projection: P = 3)
//   MODE 1   out[T][384] = LayerNorm(res + x W^T + b)                   (attention output projection + LN1)
//
// Why a second generation (round-1 hardware results): lm_linear_h384.hip lost to hipBLASLt (22.36 vs 20.27 ms per
// 2048-chunk forward).  Its wave holds x^T of 32 tokens in registers and reads ONE 1 KB weight fragment from LDS per
// MFMA; four waves per CU do that side by side, and the weight slabs reach LDS through VGPRs (ds_write_b128: 13 cycles
// per KB on top of the array cycles).  Per 768-cycle slab that is 384 LDS cycles of fragment reads + ~300 of staging
// writes: LDS bound, one wave per SIMD, nothing to hide a stall behind.  This kernel
//   * gives a wave 64 tokens x 192 output features (12 accumulator tiles): every weight fragment read from LDS feeds
//     TWO MFMAs.  x^T of the first 32-token tile sits in registers (96, loaded straight from HBM), x^T of the second
//     tile is read from an LDS-resident copy -- 7 LDS reads per 12 MFMAs instead of 12 (both tiles in registers would
//     be 6, but 192 + 192 live registers is past what hipcc allocates without scratch: measured 788 B/lane);
//     the four waves of a workgroup are 2 token groups x 2 feature halves = 128 tokens x 384 features per pass;
//   * streams the weight slabs L2 -> LDS with global_load_lds_dwordx4 (no VGPR round trip, no ds_write, 48 fewer
//     registers) through FOUR stages: slab t is computed while t+1 is known to have landed (its first fragments are
//     prefetched across the slab boundary, so the matrix pipe does not wait for LDS after a barrier), t+2 is in flight
//     and t+3 is being issued; counted vmcnt + raw s_barrier (cdna_hip_programming.md section 5: a __syncthreads()
//     would drain the DMA queue).  The LDS image is lane-linear per DMA instruction, so the bank-conflict-free layout
//     is an XOR swizzle applied to the SOURCE address and to the fragment READ (methodology rule 21);
//   * widens the epilogue stores: lane pairs (l, l ^ 32) exchange half of their 4-feature groups so that every lane
//     stores 16 contiguous bytes and a wave store covers 32 rows x 32 B (guide T21).
// Weight slab = 32 input features x 384 rows (24 KB), W packed on the host as [P][12][384][32]
// (leann_amd/encoder.py: pack_w_linear_h384 -- same packing as the first generation).
// Role in the reference: the attention projections inside compute_embeddings' BERT forward
// (leann/embedding_compute.py:229-239).
#include <cstdlib>
#include <cstring>

#include "lm_h384_common.h"

namespace lm {

constexpr int G2_SLAB_BYTES = ML_H * 64;              // 24576: 384 rows x 32 halfs
constexpr int G2_STAGES = 4;
constexpr int G2_SLABS = ML_H / 32;                   // 12 slabs per pass (12 % G2_STAGES == 0: stage of slab s is s % 4)
constexpr int G2_LDS_W = G2_STAGES * G2_SLAB_BYTES;   // 98304
constexpr int G2_XROW = ML_H * 2;                     // 768 B per token row of the LDS-resident x tile
constexpr int G2_LDS_X = 2 * 32 * G2_XROW;            // 49152: second 32-token tile of both token groups
constexpr int G2_LDS_RED = 4 * 64 * 4;                // LayerNorm partial sums: [wave][token] floats
constexpr int G2_LDS_TOTAL = G2_LDS_W + G2_LDS_X + G2_LDS_RED;  // 148480

#define g2_dma16 lm_dma16  // lm_h384_common.h (inline assembly form: see there why)
#ifdef LM_EMULATED_DEVICE
#define G2_WAIT_VM(n) ((void)0)
#define G2_BARRIER() __syncthreads()
#else
#define G2_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define G2_BARRIER() __builtin_amdgcn_s_barrier()  // raw: fragment reads of the next slab stay in flight across it
#endif

// one weight slab (24 KB) -> stage: 6 DMA instructions per wave; LDS chunk L = 256 i + tid holds source chunk
// (row = L >> 2, pos = (L & 3) ^ ((row >> 2) & 3)) of the slab
__device__ __forceinline__ void g2_issue_slab(const unsigned char* slab, unsigned char* stage, int tid) {
    const int wv = tid >> 6;
    const unsigned char* src = slab + (tid >> 2) * 64 + (((tid & 3) ^ ((tid >> 4) & 3)) << 4);
#pragma unroll
    for (int i = 0; i < 6; ++i) g2_dma16(src + i * 4096, stage + (256 * i + 64 * wv) * 16);
}

// ABL: ablation bits for on-hardware diagnosis (LEANN_MI355X_ABLATE; 0 = the product kernel): 1 = no weight DMA in the main
// loop (stale LDS), 2 = no counted wait / barrier per slab, 4 = no epilogue stores.  Wrong results by construction.
template <int MODE, int ABL>
__global__ __launch_bounds__(256) LM_ONE_WAVE_PER_SIMD void k_gemm_h384(
    const __half* __restrict__ x, const __half* __restrict__ wp, const float* __restrict__ bias, const __half* __restrict__ res,
    const __half* __restrict__ gamma, const __half* __restrict__ beta, __half* __restrict__ out, int T, int P_arg, float eps) {
    // MODE 1 is a single pass by contract (n_out == 384): telling the compiler lets it drop the x fragments before the
    // LayerNorm epilogue instead of keeping them alive "for the next pass" (measured: 348 B of scratch per lane otherwise)
    const int P = MODE == 1 ? 1 : P_arg;
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r31 = lane & 31, g = lane >> 5;
    const int tg = wv >> 1, fh = wv & 1;                  // token group (64 tokens), feature half (192 features)
    const int tok0 = blockIdx.x * 128 + tg * 64;          // first token of this wave
    const int N = ML_H * P;
    const int nslab = G2_SLABS * P;
    const unsigned char* wbytes = (const unsigned char*)wp;
    unsigned char* xlds = smem + G2_LDS_W;

    // ---- prologue: weight slabs 0..2, the LDS copy of the second token tiles, the register copy of the first ----
    g2_issue_slab(wbytes, smem, tid);
    if (nslab > 1) g2_issue_slab(wbytes + G2_SLAB_BYTES, smem + G2_SLAB_BYTES, tid);
    if (nslab > 2) g2_issue_slab(wbytes + 2 * G2_SLAB_BYTES, smem + 2 * G2_SLAB_BYTES, tid);
    // x tile: LDS chunk L = 256 i + tid  <->  token group L / 1536, row (L % 1536) / 48, position pos = L % 48 holds source
    // chunk (pos & ~15) | ((pos ^ row) & 15) of that row (16-chunk XOR swizzle: rows are 768 B = 3 x 256 B apart)
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const int L = 256 * i + tid;
        const int gi = L / 1536, rem = L - 1536 * gi, row = rem / 48, pos = rem - 48 * row;
        const int c = (pos & ~15) | ((pos ^ row) & 15);
        int token = blockIdx.x * 128 + gi * 64 + 32 + row;
        token = token < T ? token : 0;  // rows past the end are never stored
        g2_dma16((const unsigned char*)x + (int64_t)token * G2_XROW + c * 16, xlds + (256 * i + 64 * wv) * 16);
    }
    // x^T fragments of the first tile (B operand): lane (n = token, g) holds x[token][16 ks + 8 g .. + 8]
    half8 xf[ML_KS];
    {
        const int token = tok0 + r31;
        const bool valid = token < T;
        const _Float16* xr = (const _Float16*)x + (int64_t)(valid ? token : 0) * ML_H + 8 * g;
#pragma unroll
        for (int ks = 0; ks < ML_KS; ++ks) {
            half8 v = *(const half8*)(xr + 16 * ks);
            const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            xf[ks] = valid ? v : z;
        }
    }
    G2_WAIT_VM(0);
    G2_BARRIER();  // slabs 0..2 and the x tile have landed for every wave

    // A fragment of (k-step u, feature tile j): row = 192 fh + 32 j + r31, chunk (2u + g) ^ ((row >> 2) & 3)
    const int swz = (r31 >> 2) & 3;
    const int a_off0 = (192 * fh + r31) * 64 + ((g ^ swz) << 4);  // u = 0
    const int a_off1 = a_off0 ^ 32;                              // u = 1: chunk ^ 2
    // B fragment of the second tile, k-step ks: row r31 of group tg, chunk c = 2 ks + g at position (c & ~15) | ((c ^ r31) & 15)
    const unsigned char* xrow = xlds + (tg * 32 + r31) * G2_XROW;
    float* red = (float*)(smem + G2_LDS_W + G2_LDS_X);

    // fragment stream of a pass: step m = 12 s + n (slab s, k-step u = n / 6, feature tile j = n % 6); A fragments are
    // read 4 steps ahead, the second tile's B fragment one k-step ahead -- also across slab (and pass) boundaries
    auto a_ptr = [&](int m) -> const half8* {
        const int s = (m / 12) % G2_SLABS, n = m % 12, u = n / 6, j = n % 6;
        return (const half8*)(smem + (s % G2_STAGES) * G2_SLAB_BYTES + (u ? a_off1 : a_off0) + 2048 * j);
    };
    auto b_ptr = [&](int kstep) -> const half8* {
        const int c = 2 * (kstep % ML_KS) + g;
        return (const half8*)(xrow + (((c & ~15) | ((c ^ r31) & 15)) << 4));
    };
    half8 ring[4], xb[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) ring[i] = *a_ptr(i);
    xb[0] = *b_ptr(0);

    for (int p = 0; p < P; ++p) {
        // accumulators start from the bias (register 4q + i of tile j <-> feature 192 fh + 32 j + 8 q + 4 g + i): no epilogue add
        float16v o[2][6];
        {
            int g_b = g;
            LM_KEEP_LOCAL(g_b);  // keep these 24 loads inside the pass (see the note at the epilogue)
            const float* bq = bias + ML_H * p + 192 * fh + 4 * g_b;
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4v bb = {0.f, 0.f, 0.f, 0.f};
                    if (MODE == 0) bb = *(const float4v*)(bq + 32 * j + 8 * q);
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[0][j][4 * q + i] = o[1][j][4 * q + i] = bb[i];
                }
        }
#pragma unroll
        for (int s = 0; s < G2_SLABS; ++s) {  // unrolled: xf is indexed by 2s + u, the stage of slab s is s % 4
            const int t = G2_SLABS * p + s;
            if (t > 0 && !(ABL & 2)) {
                // slab t+1 (its head is prefetched at the tail of this slab) must have landed; slab t+2 may be in flight
                if (t + 2 < nslab) G2_WAIT_VM(6);
                else G2_WAIT_VM(0);
                G2_BARRIER();  // ... for every wave; and every wave is done reading slab t-1, whose stage slab t+3 takes
            }
            if (!(ABL & 1) && t + 3 < nslab) g2_issue_slab(wbytes + (int64_t)(t + 3) * G2_SLAB_BYTES, smem + ((s + 3) % G2_STAGES) * G2_SLAB_BYTES, tid);
#pragma unroll
            for (int n = 0; n < 12; ++n) {
                const int m = 12 * s + n, u = n / 6, j = n % 6, ks = 2 * s + u;
                if (j == 0) xb[(ks + 1) & 1] = *b_ptr(ks + 1);  // next k-step's B fragment (wraps to k-step 0 of the next pass)
                o[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[n & 3], xf[ks], o[0][j], 0, 0, 0);
                o[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[n & 3], xb[ks & 1], o[1][j], 0, 0, 0);
                ring[n & 3] = *a_ptr(m + 4);  // step m + 4: at the tail of a slab this is the head of the next one
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- epilogue of the pass.  Lane (token r31 of tile nt, g), tile j, register 4q + i  <->
        //      feature 192 fh + 32 j + 8 q + 4 g + i ----
        // the epilogue's read-only loads are invariant in p as far as the compiler can see: hoisted out of the pass loop they
        // would occupy ~190 registers across the main loop (measured: 308 B of scratch per lane).  An opaque lane offset
        // keeps them inside the pass.
        int g_e = g;
        LM_KEEP_LOCAL(g_e);
        const float* bp = bias + ML_H * p + 192 * fh + 4 * g_e;
        if (MODE == 1) {
            // + bias + residual, then LayerNorm over the 384 features of a token: 96 values in this lane, its lane^32
            // partner and the two lanes of the other feature-half wave hold the rest (exchange through LDS)
            float mean[2], rstd[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int token = tok0 + 32 * nt + r31;
                const _Float16* rr = (const _Float16*)res + (int64_t)(token < T ? token : 0) * ML_H + 192 * fh + 4 * g_e;
                float2v sa = {0.f, 0.f}, sb = {0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 6; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int f0 = 32 * j + 8 * q;
                        half4 xr = *(const half4*)(rr + f0);
                        float4v bb = *(const float4v*)(bp + f0);
                        float2v v0 = (float2v){o[nt][j][4 * q], o[nt][j][4 * q + 1]} + ((float2v){(float)xr[0], (float)xr[1]} + (float2v){bb[0], bb[1]});
                        float2v v1 = (float2v){o[nt][j][4 * q + 2], o[nt][j][4 * q + 3]} + ((float2v){(float)xr[2], (float)xr[3]} + (float2v){bb[2], bb[3]});
                        o[nt][j][4 * q] = v0[0];
                        o[nt][j][4 * q + 1] = v0[1];
                        o[nt][j][4 * q + 2] = v1[0];
                        o[nt][j][4 * q + 3] = v1[1];
                        sa += v0;
                        sb += v1;
                    }
                float sum = (sa[0] + sa[1]) + (sb[0] + sb[1]);
                sum += __shfl_xor(sum, 32);
                if (g == 0) red[wv * 64 + 32 * nt + r31] = sum;
            }
            __syncthreads();
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                // fixed order (feature half 0 + feature half 1): both waves of a token group compute identical statistics
                const float tot = red[(2 * tg) * 64 + 32 * nt + r31] + red[(2 * tg + 1) * 64 + 32 * nt + r31];
                mean[nt] = tot * (1.0f / ML_H);
            }
            __syncthreads();
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const float2v nm = {-mean[nt], -mean[nt]};
                float2v qa = {0.f, 0.f}, qb = {0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 6; ++j)
#pragma unroll
                    for (int r = 0; r < 16; r += 4) {
                        float2v d0 = (float2v){o[nt][j][r], o[nt][j][r + 1]} + nm, d1 = (float2v){o[nt][j][r + 2], o[nt][j][r + 3]} + nm;
                        qa = __builtin_elementwise_fma(d0, d0, qa);
                        qb = __builtin_elementwise_fma(d1, d1, qb);
                    }
                float sq = (qa[0] + qa[1]) + (qb[0] + qb[1]);
                sq += __shfl_xor(sq, 32);
                if (g == 0) red[wv * 64 + 32 * nt + r31] = sq;
            }
            __syncthreads();
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const float tot = red[(2 * tg) * 64 + 32 * nt + r31] + red[(2 * tg + 1) * 64 + 32 * nt + r31];
                rstd[nt] = rsqrtf(tot * (1.0f / ML_H) + eps);
            }
            const _Float16* gm = (const _Float16*)gamma + 192 * fh + 4 * g_e;
            const _Float16* bt = (const _Float16*)beta + 192 * fh + 4 * g_e;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const float2v nm = {-mean[nt], -mean[nt]}, rs = {rstd[nt], rstd[nt]};
#pragma unroll
                for (int j = 0; j < 6; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int f0 = 32 * j + 8 * q;
                        half4 gv = *(const half4*)(gm + f0), bv = *(const half4*)(bt + f0);
                        float2v n0 = ((float2v){o[nt][j][4 * q], o[nt][j][4 * q + 1]} + nm) * rs;
                        float2v n1 = ((float2v){o[nt][j][4 * q + 2], o[nt][j][4 * q + 3]} + nm) * rs;
                        float2v y0 = __builtin_elementwise_fma(n0, (float2v){(float)gv[0], (float)gv[1]}, (float2v){(float)bv[0], (float)bv[1]});
                        float2v y1 = __builtin_elementwise_fma(n1, (float2v){(float)gv[2], (float)gv[3]}, (float2v){(float)bv[2], (float)bv[3]});
                        o[nt][j][4 * q] = y0[0];
                        o[nt][j][4 * q + 1] = y0[1];
                        o[nt][j][4 * q + 2] = y1[0];
                        o[nt][j][4 * q + 3] = y1[1];
                    }
            }
        }
        // ---- store.  Lanes l and l ^ 32 (same token, g = 0 / 1) hold features [8q, 8q+4) / [8q+4, 8q+8) of every group q:
        //      for a pair of groups (q, q+1) one v_permlane32_swap per dword gives the g = 0 lane features [8q, 8q+8) and the
        //      g = 1 lane [8q+8, 8q+16): 16 contiguous bytes per lane, 32 rows x 32 B per wave store (guide T21) ----
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int token = tok0 + 32 * nt + r31;
            _Float16* yr = (_Float16*)out + (int64_t)token * N + ML_H * p + 192 * fh + 8 * g;
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int qp = 0; qp < 2; ++qp) {
                    half4 h0, h1;  // groups q = 2 qp, 2 qp + 1
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        h0[i] = (_Float16)o[nt][j][8 * qp + i];
                        h1[i] = (_Float16)o[nt][j][8 * qp + 4 + i];
                    }
                    uint2 a = __builtin_bit_cast(uint2, h0), b = __builtin_bit_cast(uint2, h1);
                    lane32_swap(a.x, b.x);
                    lane32_swap(a.y, b.y);
                    const uint4 y = {a.x, a.y, b.x, b.y};
                    if (ABL & 4) asm volatile("" ::"v"(y.x), "v"(y.y), "v"(y.z), "v"(y.w));
                    else if (token < T) *(uint4*)(yr + 32 * j + 16 * qp) = y;
                }
        }
    }
}

}  // namespace lm

#ifndef LM_HOST_EMULATION
extern "C" int lm_gemm_h384_f16(const void* d_x, const void* d_wp, const float* d_bias, int32_t n_out, const void* d_residual,
                                const void* d_gamma, const void* d_beta, float eps, void* d_out, int64_t tokens, void* stream) {
    using namespace lm;
    if (tokens == 0) return LM_OK;
    if (!d_x || !d_wp || !d_bias || !d_out || tokens < 0 || tokens > 0x7fffffff) LM_FAIL(LM_EINVAL, "bad linear arguments");
    if (n_out <= 0 || n_out % ML_H) LM_FAIL(LM_EINVAL, "n_out must be a positive multiple of 384");
    const bool ln = d_residual != nullptr;
    if (ln && (n_out != ML_H || !d_gamma || !d_beta)) LM_FAIL(LM_EINVAL, "residual + LayerNorm mode needs n_out == 384, gamma and beta");
    dim3 grid((unsigned)((tokens + 127) / 128)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const __half *x = (const __half*)d_x, *w = (const __half*)d_wp, *r = (const __half*)d_residual;
    const __half *gm = (const __half*)d_gamma, *bt = (const __half*)d_beta;
    const char* ab = getenv("LEANN_MI355X_ABLATE");
    const int abl = ab ? atoi(ab) : 0;
    if (ln) {
        LM_HIP(hipFuncSetAttribute((const void*)k_gemm_h384<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS_TOTAL));
        hipLaunchKernelGGL((k_gemm_h384<1, 0>), grid, block, G2_LDS_TOTAL, st, x, w, d_bias, r, gm, bt, (__half*)d_out, (int)tokens, 1, eps);
    } else {
#define G2_GO(A)                                                                                                                          \
    case A:                                                                                                                                \
        LM_HIP(hipFuncSetAttribute((const void*)k_gemm_h384<0, A>, hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS_TOTAL));             \
        hipLaunchKernelGGL((k_gemm_h384<0, A>), grid, block, G2_LDS_TOTAL, st, x, w, d_bias, r, gm, bt, (__half*)d_out, (int)tokens,       \
                           n_out / ML_H, eps);                                                                                             \
        break
        switch (abl) {
            G2_GO(0); G2_GO(1); G2_GO(2); G2_GO(3); G2_GO(4); G2_GO(7);
            default: LM_FAIL(LM_EINVAL, "LEANN_MI355X_ABLATE must be 0, 1, 2, 3, 4 or 7");
        }
#undef G2_GO
    }
    LM_HIP(hipGetLastError());
    return LM_OK;
}
#endif  // LM_HOST_EMULATION
