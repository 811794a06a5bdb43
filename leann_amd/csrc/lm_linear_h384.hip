// lm_linear_h384.hip -- linear layers with 384 input features on the transposed MFMA layout of lm_mlp_fused.hip:
//
//   MODE 0   out[T][N] = x W^T + b                          N = 384 P   (QKV projection: P = 3)
//   MODE 1   out[T][384] = LayerNorm(res + x W^T + b)                   (attention output projection + LN1)
//
// STATUS: first generation of the 384-input linear kernels, validated on the MI355X (tests/test_gpu_encoder_kernels.py) and superseded by
// the weight-stationary form (lm_gemm_ws_h384.hip, the default); LEANN_MI355X_LINEAR=1 selects it for A/B runs.
//
// Why: with K = 384 the library GEMMs of the layer run at ~240 TFLOP/s (12 k-iterations per 256x256 tile: the
// prologue / epilogue of every tile is as long as its main loop) -- 1.3 ms of a 4.3 ms layer for 262k tokens
// (profiles/r1_final_bench_default_kernel_stats.csv, MT128x256x32: 2 calls per layer).  K = 384 is small enough
// to keep a wave's whole input slice in registers instead: x^T of 32 tokens = 24 B fragments (96 registers), so
// the main loop is MFMA + one LDS read per MFMA and nothing else.
//
// One 256-thread workgroup = 4 waves = 128 tokens; per pass a wave accumulates out^T [384 x 32 tokens] in 192
// registers:  out^T += W[384 rows][k slab] (A operand, LDS) . x^T[k slab] (B operand, registers).
// Weights stream through LDS in slabs of 32 input features (384 rows x 64 B = 24 KB, rows padded to 80 B),
// double buffered, 12 slabs per pass; W is packed on the host as [P][12][384][32] so that a slab is one
// contiguous 24 KB block (leann_amd/encoder.py: pack_w_linear_h384).  MODE 1 finishes with the shared
// bias + residual + LayerNorm epilogue (a lane pair holds all 384 features of its token).
// Role in the reference: the attention projections inside compute_embeddings' BERT forward
// (leann/embedding_compute.py:229-239).
#include <cstdlib>
#include <cstring>

#include "lm_h384_common.h"

namespace lm {

constexpr int LN_STRIDE = 40;                         // halfs per weight row in LDS (80 B)
constexpr int LN_BUF = ML_H * LN_STRIDE * 2;          // 30720 B per stage
constexpr int LN_CHUNKS = ML_H * 32 * 2 / 16;         // 1536 16-byte chunks per slab
constexpr int LN_NPRE = LN_CHUNKS / 256;              // 6 per thread
constexpr int LN_SLABS = ML_H / 32;                   // 12 slabs per pass
constexpr int LN_TILE_STRIDE = ML_H + 8;              // halfs per row of a wave's output staging tile (784 B)
constexpr int LN_TILE_BYTES = 4 * 32 * LN_TILE_STRIDE * 2;  // 100352 B for the four waves

template <int MODE, bool LDS_STORE>
__global__ __launch_bounds__(256) LM_ONE_WAVE_PER_SIMD void k_linear_h384(
    const __half* __restrict__ x, const __half* __restrict__ wp, const float* __restrict__ bias, const __half* __restrict__ res,
    const __half* __restrict__ gamma, const __half* __restrict__ beta, __half* __restrict__ out, int T, int P, float eps) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r31 = lane & 31, g = lane >> 5;
    const int token = blockIdx.x * 128 + wv * 32 + r31;
    const bool valid = token < T;
    const int N = ML_H * P;

    // x^T fragments (B operand): lane (n = token, g) holds x[token][16ks + 8g .. +8]
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
    int off[LN_NPRE];
#pragma unroll
    for (int i = 0; i < LN_NPRE; ++i) {
        const int c = tid + 256 * i;
        off[i] = (c >> 2) * (LN_STRIDE * 2) + (c & 3) * 16;
    }
    // Weight prefetch runs TWO slabs ahead in registers (a slab is only 24 MFMAs = 768 cycles, about one loaded L2
    // round trip): slab t+2 is loaded during slab t into pre[t & 1] and written to LDS at the end of slab t+1 -- into
    // the stage slab t was read from, which is idle by then.  LDS stays double buffered.
    u32x4 pre[2][LN_NPRE];
    const u32x4* gw = (const u32x4*)wp + tid;  // + slab * LN_CHUNKS + 256 i   (slab = 12 p + s)
    const int nslab = LN_SLABS * P;
#pragma unroll
    for (int i = 0; i < LN_NPRE; ++i) pre[0][i] = gw[256 * i];
#pragma unroll
    for (int i = 0; i < LN_NPRE; ++i) *(u32x4*)(smem + off[i]) = pre[0][i];
    if (nslab > 1) {
#pragma unroll
        for (int i = 0; i < LN_NPRE; ++i) pre[1][i] = gw[LN_CHUNKS + 256 * i];  // slab 1: stored at the end of slab 0
    }
    __syncthreads();

    const _Float16* Ws = (const _Float16*)smem + r31 * LN_STRIDE + 8 * g;  // A fragment base inside a stage (m = out row r31)
    for (int p = 0; p < P; ++p) {
        float16v o[ML_NJ];
#pragma unroll
        for (int j = 0; j < ML_NJ; ++j) o[j] = (float16v){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < LN_SLABS; ++s) {  // unrolled: xf is indexed by 2s + u
            // stage parity: slab t = 12p + s lives in stage t & 1 = s & 1 (12 is even)
            const _Float16* cur = Ws + (s & 1) * (LN_BUF / 2);
            unsigned char* nxt = smem + ((s + 1) & 1) * LN_BUF;
            const int t = LN_SLABS * p + s;
            const bool more = t + 1 < nslab, more2 = t + 2 < nslab;
            if (more2) {
                const u32x4* src = gw + (int64_t)(t + 2) * LN_CHUNKS;
#pragma unroll
                for (int i = 0; i < LN_NPRE; ++i) pre[s & 1][i] = src[256 * i];
            }
            // 24 products, a ring of 4 A fragments read 4 products ahead (source order = issue order)
            half8 ring[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) ring[i] = *(const half8*)(cur + 32 * i * LN_STRIDE);
#pragma unroll
            for (int n = 0; n < 2 * ML_NJ; ++n) {  // product n: k-step u = n / 12, tile j = n % 12
                const int u = n / ML_NJ, j = n % ML_NJ;
                o[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[n & 3], xf[2 * s + u], o[j], 0, 0, 0);
                if (n + 4 < 2 * ML_NJ) {
                    const int u2 = (n + 4) / ML_NJ, j2 = (n + 4) % ML_NJ;
                    ring[n & 3] = *(const half8*)(cur + 32 * j2 * LN_STRIDE + 16 * u2);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more) {  // slab t+1, loaded during slab t-1 (or in the prologue)
#pragma unroll
                for (int i = 0; i < LN_NPRE; ++i) *(u32x4*)(nxt + off[i]) = pre[(s + 1) & 1][i];
            }
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);  // the 12 slabs are one basic block: nothing may drift from one slab into another
        }
        if (MODE == 1) {
            // T > 0 always holds; the branch only gives the epilogue its own basic block.  Inside the unrolled slab
            // sequence its ~100 chain-free loads (read-only arguments) are placed at the TOP of the kernel by instruction
            // selection and spilled (820 B of scratch per lane); sched_barrier does not stop that.
            if (T > 0) mlp_epilogue(o, res, bias, gamma, beta, out, token, valid, g, eps);
        } else if (LDS_STORE) {
            // Coalesced output: the transposed accumulator layout gives every lane 4 consecutive columns of ITS token, i.e. a
            // wave store instruction would touch 32 rows with 16 bytes each.  Stage the wave's 32 x 384 tile in LDS (row major,
            // 784-byte rows: conflict-free 8-byte writes) and write it out as 24 fully contiguous 1 KB wave stores.
            // The staging area lies behind the two weight stages (the next pass's first slab is already in stage 0).
            _Float16* tile = (_Float16*)(smem + 2 * LN_BUF) + wv * (32 * LN_TILE_STRIDE);
            int g_e = g;
            LM_KEEP_LOCAL(g_e);
            const float* bp = bias + ML_H * p + 4 * g_e;
#pragma unroll
            for (int j = 0; j < ML_NJ; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int f0 = 32 * j + 8 * q;
                    float4v bb = *(const float4v*)(bp + f0);
                    half4 y;
#pragma unroll
                    for (int i = 0; i < 4; ++i) y[i] = (_Float16)(o[j][4 * q + i] + bb[i]);
                    *(half4*)(tile + r31 * LN_TILE_STRIDE + f0 + 4 * g_e) = y;
                    __builtin_amdgcn_sched_barrier(0);
                }
            LM_WAVE_SYNC();  // the tile is written and read by the same wave only
            const int tok0 = blockIdx.x * 128 + wv * 32;
            int lane_e = lane;
            LM_KEEP_LOCAL(lane_e);  // keep the 24 address computations here (hoisted to the pass header they are spilled)
#pragma unroll
            for (int i = 0; i < 24; ++i) {
                const int c = lane_e + 64 * i, row = c / 48, col = (c % 48) * 8;
                half8 v = *(const half8*)(tile + row * LN_TILE_STRIDE + col);
                if (tok0 + row < T) *(half8*)((_Float16*)out + (int64_t)(tok0 + row) * N + ML_H * p + col) = v;
                if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // at most 4 tile reads in flight (16 registers, not 96)
            }
            LM_WAVE_SYNC();  // the next pass overwrites the tile
        } else if (valid) {
            // lane (token r31, g), tile j, register 4q + i <-> output column 384p + 32j + 8q + 4g + i
            _Float16* yr = (_Float16*)out + (int64_t)token * N + ML_H * p + 4 * g;
            const float* bp = bias + ML_H * p + 4 * g;
#pragma unroll
            for (int j = 0; j < ML_NJ; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int f0 = 32 * j + 8 * q;
                    float4v bb = *(const float4v*)(bp + f0);
                    half4 y;
#pragma unroll
                    for (int i = 0; i < 4; ++i) y[i] = (_Float16)(o[j][4 * q + i] + bb[i]);
                    *(half4*)(yr + f0) = y;
                }
        }
    }
}

}  // namespace lm

#ifndef LM_HOST_EMULATION
extern "C" int lm_linear_h384_f16(const void* d_x, const void* d_wp, const float* d_bias, int32_t n_out, const void* d_residual,
                                  const void* d_gamma, const void* d_beta, float eps, void* d_out, int64_t tokens, void* stream) {
    using namespace lm;
    if (tokens == 0) return LM_OK;
    if (!d_x || !d_wp || !d_bias || !d_out || tokens < 0 || tokens > 0x7fffffff) LM_FAIL(LM_EINVAL, "bad linear arguments");
    if (n_out <= 0 || n_out % ML_H) LM_FAIL(LM_EINVAL, "n_out must be a positive multiple of 384");
    const bool ln = d_residual != nullptr;
    if (ln && (n_out != ML_H || !d_gamma || !d_beta)) LM_FAIL(LM_EINVAL, "residual + LayerNorm mode needs n_out == 384, gamma and beta");
    const char* st_env = getenv("LEANN_MI355X_LINEAR_STORE");  // "direct": 8-byte scattered stores straight from the accumulators (A/B)
    const bool lds_store = !ln && !(st_env && !strcmp(st_env, "direct"));
    const size_t shmem = (size_t)2 * LN_BUF + (lds_store ? LN_TILE_BYTES : 0);  // 61440 (+ 100352 = 161792 <= 160 KiB)
    dim3 grid((unsigned)((tokens + 127) / 128)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const __half *x = (const __half*)d_x, *w = (const __half*)d_wp, *r = (const __half*)d_residual;
    const __half *gm = (const __half*)d_gamma, *bt = (const __half*)d_beta;
    if (ln) {
        hipLaunchKernelGGL((k_linear_h384<1, false>), grid, block, shmem, st, x, w, d_bias, r, gm, bt, (__half*)d_out, (int)tokens, 1, eps);
    } else if (lds_store) {
        LM_HIP(hipFuncSetAttribute((const void*)k_linear_h384<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL((k_linear_h384<0, true>), grid, block, shmem, st, x, w, d_bias, r, gm, bt, (__half*)d_out, (int)tokens,
                           n_out / ML_H, eps);
    } else {
        hipLaunchKernelGGL((k_linear_h384<0, false>), grid, block, shmem, st, x, w, d_bias, r, gm, bt, (__half*)d_out, (int)tokens,
                           n_out / ML_H, eps);
    }
    LM_HIP(hipGetLastError());
    return LM_OK;
}
#endif  // LM_HOST_EMULATION
