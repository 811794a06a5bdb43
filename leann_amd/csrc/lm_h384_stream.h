// lm_h384_stream.h -- what the weight-streaming kernels of the hidden-384 forward share (lm_layer_tail_h384.hip, lm_qkv_h384.hip):
// LDS-DMA of ready-made weight images in groups of 1 KB pieces (one M0 write per group: the instruction offset applies to the global
// and to the LDS address alike), a wave's 32 token rows as a swizzled 24 KB row tile, counted waits.
#pragma once
#include "lm_h384_common.h"

namespace lm {

constexpr int T4_SLAB = 24576;  // bytes of a weight slab image: 32 rows x 384 k (W1, W_qkv) or 384 rows x 32 k (W2, W_o)

#ifdef LM_EMULATED_DEVICE
#define T4_WAIT_VM(n) ((void)0)
#define T4_WAIT_LGKM0() ((void)0)
#define T4_BARRIER() __syncthreads()
#else
#define T4_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define T4_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define T4_BARRIER() __builtin_amdgcn_s_barrier()
#endif

// ---- LDS-DMA -------------------------------------------------------------------------------------------------------------------
// NP consecutive 1 KB pieces: LDS [dst + 1024 p + 16 lane, +16) <- global [sbase + voff + 1024 p, +16), p = 0 .. NP-1.  dst and sbase
// are wave uniform.  ONE M0 write serves the group: the instruction offset is added to the global AND to the LDS address.
// Inline assembly for the reason given in lm_h384_common.h (lm_dma16); M0 is written in the statement that uses it.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"  // the "m0" clobbers (lm_h384_common.h says why they are there)
template <int NP>
__device__ __forceinline__ void t4_dma_group(const void* sbase, unsigned voff, unsigned char* dst) {
    static_assert(NP >= 1 && NP <= 4, "instruction offsets reach 3072");
#ifdef LM_EMULATED_DEVICE
    for (int p = 0; p < NP; ++p) std::memcpy(dst + 1024 * p + 16 * (threadIdx.x & 63), (const unsigned char*)sbase + voff + 1024 * p, 16);
#else
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(size_t)dst);  // low half of the flat address = LDS offset
    if constexpr (NP == 4)
        asm volatile("s_mov_b32 m0, %2\n\t" LM_DMA_NOPS "global_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:2048\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072" ::"v"(voff), "s"(sbase), "s"(m0v) : "memory", "m0");
    else if constexpr (NP == 2)
        asm volatile("s_mov_b32 m0, %2\n\t" LM_DMA_NOPS "global_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024" ::"v"(voff), "s"(sbase), "s"(m0v) : "memory", "m0");
    else if constexpr (NP == 1)
        asm volatile("s_mov_b32 m0, %2\n\t" LM_DMA_NOPS "global_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(m0v) : "memory", "m0");
    else
        asm volatile("s_mov_b32 m0, %2\n\t" LM_DMA_NOPS "global_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:2048" ::"v"(voff), "s"(sbase), "s"(m0v) : "memory", "m0");
#endif
}
// piece Q (0 .. 3) of a group ALONE: the variants that spread the pieces over the MFMA gaps issue Q = 0 with the M0 write and
// Q = 1 .. 3 on the M0 it left behind (nothing else in this kernel writes M0: scripts/isa_report.sh checks the disassembly)
template <int Q>
__device__ __forceinline__ void t4_dma_piece(const void* sbase, unsigned voff, unsigned char* dst) {
#ifdef LM_EMULATED_DEVICE
    std::memcpy(dst + 1024 * Q + 16 * (threadIdx.x & 63), (const unsigned char*)sbase + voff + 1024 * Q, 16);
#else
    if constexpr (Q == 0) {
        const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(size_t)dst);
        asm volatile("s_mov_b32 m0, %2\n\t" LM_DMA_NOPS "global_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(m0v) : "memory", "m0");
    } else if constexpr (Q == 1) asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" ::"v"(voff), "s"(sbase) : "memory");
    else if constexpr (Q == 2) asm volatile("global_load_lds_dwordx4 %0, %1 offset:2048" ::"v"(voff), "s"(sbase) : "memory");
    else asm volatile("global_load_lds_dwordx4 %0, %1 offset:3072" ::"v"(voff), "s"(sbase) : "memory");
#endif
}
#pragma clang diagnostic pop
// a wave's quarter (6 KB: pieces 6 wv .. 6 wv + 5) of a 24 KB image -> the same place of a stage: prologue fills and the W_o ring
__device__ __forceinline__ void t4_copy_quarter(const unsigned char* img, unsigned char* stage, int wv, unsigned voff) {
    t4_dma_group<4>(img + 6144 * wv, voff, stage + 6144 * wv);
    t4_dma_group<2>(img + 6144 * wv + 4096, voff, stage + 6144 * wv + 4096);
}

// 32 token rows -- one contiguous 24 KB block of a [T][384] fp16 matrix -- into a 24 KB stage as the image of a W1 slab (row r,
// 16-byte chunk c at position (c & ~15) | ((c ^ r) & 15)): 24 fully coalesced 1 KB pieces, of which this wave issues pieces
// [P0, P0 + NP).  Rows >= rows_valid (past the end of the matrix) repeat the last valid row.  (Activations cannot be pre-swizzled: the
// permutation is in the source offsets.)
template <int P0, int NP>
__device__ __forceinline__ void t4_issue_rows(const unsigned char* rows, int rows_valid, unsigned char* stage, int lane) {
    LM_KEEP_LOCAL(lane);  // the source offsets are a few VALU operations each: recomputed per call, not kept alive between calls
#pragma unroll
    for (int p = P0; p < P0 + NP; ++p) {
        const int L = 64 * p + lane, row = L / 48, pos = L - 48 * row;
        const int rc = row < rows_valid ? row : rows_valid - 1;
#if defined(LM_T4_NT) && LM_T4_NT
        lm_dma16_sv_nt(rows, (unsigned)(rc * 768 + (((pos & ~15) | ((pos ^ row) & 15)) << 4)), stage + 1024 * p);
#else
        lm_dma16_sv(rows, (unsigned)(rc * 768 + (((pos & ~15) | ((pos ^ row) & 15)) << 4)), stage + 1024 * p);
#endif
    }
}

template <int N>
__device__ __forceinline__ void t4_wait_vm() {
#ifndef LM_EMULATED_DEVICE
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
// s_waitcnt lgkmcnt(N) alone (vmcnt 63, expcnt 7): a BUILTIN, so the compiler's own wait insertion sees it and drops the per-MFMA
// waits it covers
template <int N>
__device__ __forceinline__ void t4_wait_lgkm() {
#ifndef LM_EMULATED_DEVICE
    __builtin_amdgcn_s_waitcnt(0xC07F | (N << 8));
#endif
}


}  // namespace lm
