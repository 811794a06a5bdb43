// lm_gemm_ws_h384.hip -- WEIGHT-STATIONARY linear layer for 384 input features:   out[T][N] = x W^T + b,   N = 192 nblk.
//
// Why this form (hardware numbers, profiles/r2_kbench_encoder_kernels_262k_tokens.txt + the ablation runs of the streaming kernel that
// preceded it, one wave per SIMD, 64 tokens x 192 features per wave): with ONE wave per SIMD nothing covers a wave's own load, barrier, DMA-issue and store phases --
// QKV at 262k tokens takes 406 us, of which the stores alone are 160 us, the weight DMA + per-slab waits 120 us; the bare
// MFMA / LDS loop is 198 us.  Streaming the weights through LDS slab by slab is what forces the per-slab barriers, and
// holding 64 tokens x 192 features of accumulators per wave is what forces one wave per SIMD.  K = 384 is small enough to
// turn the loop nest inside out:
//   * a workgroup keeps its 192 x 384 weight block (147 KB) RESIDENT in LDS for its whole life (loaded once by DMA, XOR
//     swizzled) and streams token tiles past it: no weight traffic, no barrier, no DMA in the main loop;
//   * 8 waves per workgroup = 2 per SIMD, each with its own 32-token tile: x^T fragments in registers (96), 6 accumulator
//     tiles (96): ~230 registers.  The waves run independently, so one wave's x loads / epilogue stores overlap the MFMAs
//     of the other wave on its SIMD;
//   * the nblk workgroups that need the same token tiles run on the same XCD (block b -> XCD b % 8 is how the hardware
//     dispatches today; a different mapping costs speed only): x is fetched from HBM once and re-read from that XCD's L2.
// W is packed on the host as [nblk][192][384] fp16 = the nn.Linear weight itself ([N][384] row major): no repacking.
// Role in the reference: the attention projections inside compute_embeddings' BERT forward (leann/embedding_compute.py:229-239).
#include <cstdlib>
#include <cstring>

#include "lm_h384_common.h"

namespace lm {

constexpr int WS_ROWS = 192;                       // output features per workgroup
constexpr int WS_W_BYTES = WS_ROWS * ML_H * 2;      // 147456
constexpr int WS_LDS_TOTAL = WS_W_BYTES + WS_ROWS * 4;  // + the bias slice as floats

#define ws_dma16 lm_dma16  // lm_h384_common.h (inline assembly form: see there why)
#ifdef LM_EMULATED_DEVICE
#define WS_WAIT_VM0() ((void)0)
#else
#define WS_WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif

// grid: 256 workgroups (one per CU).  xcd = b % 8, slot = b / 8 (0..31); groups per XCD gpx = 32 / nblk; slots >= gpx * nblk idle.
// feature block fb = slot % nblk, token group tg = xcd * gpx + slot / nblk of ntg = 8 gpx; the workgroup takes the 256-token
// tiles tg, tg + ntg, ...; wave w of it the 32 tokens [32 w, 32 w + 32) of each.
// STAMP (LEANN_MI355X_ABLATE=64 / 65, diagnosis only): s_memtime around the three phases of every tile -- row loads (waited for
// in full, which the product kernel does not do), the 144 MFMAs, the stores (65: drained with vmcnt(0)) -- summed per wave and
// written over the first bytes of `out` when the wave is done: {loads, mfma, stores, tiles} as four 64-bit words per wave
// (scripts/kbench.cpp "wsgemm"; profiles/r2_kbench_gemm_ws_phase_stamps.jsonl).
template <int STAMP>
__global__ __launch_bounds__(512) LM_TWO_WAVES_PER_SIMD void k_gemm_ws_h384(
    const __half* __restrict__ x, const __half* __restrict__ w, const float* __restrict__ bias, __half* __restrict__ out, int T, int nblk) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r31 = lane & 31, g = lane >> 5;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int gpx = 32 / nblk;
    if (slot >= gpx * nblk) return;
    const int fb = slot % nblk, tg = xcd * gpx + slot / nblk, ntg = 8 * gpx;
    const int N = WS_ROWS * nblk;

    // ---- weight block -> LDS, once: [192 rows][48 chunks of 16 B]; LDS chunk L = 512 i + tid = (row = L / 48, pos = L % 48)
    //      holds source chunk (pos & ~15) | ((pos ^ row) & 15) of that row (rows are 768 B = 3 x 256 B apart) ----
    {
        const unsigned char* wb = (const unsigned char*)w + (int64_t)fb * WS_W_BYTES;
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            const int L = 512 * i + tid, row = L / 48, pos = L - 48 * row;
            const int c = (pos & ~15) | ((pos ^ row) & 15);
            ws_dma16(wb + row * 768 + c * 16, smem + (512 * i + 64 * wv) * 16);
        }
    }
    float* bs = (float*)(smem + WS_W_BYTES);  // bias slice of this feature block
    if (tid < WS_ROWS) bs[tid] = bias[WS_ROWS * fb + tid];
    WS_WAIT_VM0();
    __syncthreads();  // the only barrier of the kernel: the weight block is complete

    // A fragment of (tile j, k-step ks): row 32 j + r31, chunk c = 2 ks + g at position (c & ~15) | ((c ^ r31) & 15)
    // two address sets (tiles 0..2 / 3..5) so that every read is base register + 16-bit immediate
    int a1[2][8];
#pragma unroll
    for (int k7 = 0; k7 < 8; ++k7) {
        a1[0][k7] = r31 * 768 + ((((2 * k7 + g) ^ r31) & 15) << 4);
        a1[1][k7] = a1[0][k7] + 3 * 24576;
    }
    const float* bl = bs + 4 * g;  // tile j, register 4q + i <-> feature 32 j + 8 q + 4 g + i of the block

    const int ntile = (T + 255) / 256;
    [[maybe_unused]] unsigned long long tsum[4] = {0, 0, 0, 0}, t0 = 0, t1 = 0, t2 = 0;
    for (int tile = tg; tile < ntile; tile += ntg) {
#ifndef LM_EMULATED_DEVICE
        if constexpr (STAMP != 0) {
            __builtin_amdgcn_sched_barrier(0);
            t0 = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
        const int token = tile * 256 + wv * 32 + r31;
        const bool valid = token < T;
        half8 xf[ML_KS];
        {
            const _Float16* xr = (const _Float16*)x + (int64_t)(valid ? token : 0) * ML_H + 8 * g;
#pragma unroll
            for (int ks = 0; ks < ML_KS; ++ks) xf[ks] = *(const half8*)(xr + 16 * ks);
        }
#ifndef LM_EMULATED_DEVICE
        if constexpr (STAMP != 0) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0);
            t1 = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
        float16v o[6];
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4v bb = *(const float4v*)(bl + 32 * j + 8 * q);
#pragma unroll
                for (int i = 0; i < 4; ++i) o[j][4 * q + i] = bb[i];
            }
        // 144 products: step m = 6 ks + j; A fragments are read 4 steps ahead
        half8 ring[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) ring[i] = *(const half8*)(smem + a1[i / 3][0] + 24576 * (i % 3));  // ks = 0, tile j = i
#pragma unroll
        for (int m = 0; m < 144; ++m) {
            const int ks = m / 6, j = m % 6;
            o[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[m & 3], xf[ks], o[j], 0, 0, 0);
            if (m + 4 < 144) {
                const int ks2 = (m + 4) / 6, j2 = (m + 4) % 6;
                ring[m & 3] = *(const half8*)(smem + a1[j2 / 3][ks2 & 7] + 256 * (ks2 >> 3) + 24576 * (j2 % 3));
            }
            __builtin_amdgcn_sched_barrier(0);  // source order = issue order: the register budget (256 per wave) has no room for hoisted reads
        }
#ifndef LM_EMULATED_DEVICE
        if constexpr (STAMP != 0) {
            __builtin_amdgcn_sched_barrier(0);
            t2 = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
        // ---- store: lanes l / l ^ 32 trade halves (v_permlane32_swap) so that every lane owns 8 consecutive features ----
        _Float16* yr = (_Float16*)out + (int64_t)token * N + WS_ROWS * fb + 8 * g;
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int qp = 0; qp < 2; ++qp) {
                half4 h0, h1;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    h0[i] = (_Float16)o[j][8 * qp + i];
                    h1[i] = (_Float16)o[j][8 * qp + 4 + i];
                }
                uint2 a = __builtin_bit_cast(uint2, h0), b = __builtin_bit_cast(uint2, h1);
                lane32_swap(a.x, b.x);
                lane32_swap(a.y, b.y);
                const uint4 y = {a.x, a.y, b.x, b.y};
                if (valid) *(uint4*)(yr + 32 * j + 16 * qp) = y;
                __builtin_amdgcn_sched_barrier(0);
            }
#ifndef LM_EMULATED_DEVICE
        if constexpr (STAMP != 0) {
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (STAMP == 2) __builtin_amdgcn_s_waitcnt(0);
            const unsigned long long t3 = __builtin_amdgcn_s_memtime();
            tsum[0] += t1 - t0;
            tsum[1] += t2 - t1;
            tsum[2] += t3 - t2;
            tsum[3] += 1;
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
    }
#ifndef LM_EMULATED_DEVICE
    if constexpr (STAMP != 0) {
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (lane == 0) {
            unsigned long long* dst = (unsigned long long*)out + (blockIdx.x * 8 + wv) * 4;
            for (int i = 0; i < 4; ++i) dst[i] = tsum[i];
        }
    }
#endif
}

}  // namespace lm

#ifndef LM_HOST_EMULATION
extern "C" int lm_gemm_ws_h384_f16(const void* d_x, const void* d_w, const float* d_bias, int32_t n_out, void* d_out, int64_t tokens,
                                   void* stream) {
    using namespace lm;
    if (tokens == 0) return LM_OK;
    if (!d_x || !d_w || !d_bias || !d_out || tokens < 0 || tokens > 0x7fffffff) LM_FAIL(LM_EINVAL, "bad linear arguments");
    if (n_out <= 0 || n_out % WS_ROWS || n_out / WS_ROWS > 32) LM_FAIL(LM_EINVAL, "n_out must be a multiple of 192, at most 6144");
    KtScope kt(LM_KT_GEMM_WS, stream, 2.0 * (double)tokens * n_out * ML_H);
#define WS_GO(S)                                                                                                                          \
    do {                                                                                                                                  \
        static DynLdsAttr attr; /* once per process and device, not per launch */                                                        \
        LM_HIP(ensure_dyn_lds(attr, (const void*)k_gemm_ws_h384<S>, WS_LDS_TOTAL));                                                       \
        hipLaunchKernelGGL(k_gemm_ws_h384<S>, dim3(256), dim3(512), WS_LDS_TOTAL, (hipStream_t)stream, (const __half*)d_x, (const __half*)d_w, \
                           d_bias, (__half*)d_out, (int)tokens, n_out / WS_ROWS);                                                          \
    } while (0)
#ifdef LM_DIAG  // stamped builds only in the diagnosis library (scripts/build_kbench.sh)
    static const int abl = [] { const char* ab = getenv("LEANN_MI355X_ABLATE"); return ab ? atoi(ab) : 0; }();
    if (abl == 64) WS_GO(1);
    else if (abl == 65) WS_GO(2);
    else
#endif
        WS_GO(0);
#undef WS_GO
    LM_HIP(hipGetLastError());
    return LM_OK;
}
#endif  // LM_HOST_EMULATION
