// vopbench.cpp -- issue cost of the VALU / MFMA / LDS instructions the attention kernel's softmax is made of, measured on the MI355X itself
// (s_memtime around an unrolled stream of 16 independent chains), at 1 / 2 / 4 waves per SIMD, plus the lane mapping of
// ds_read_b64_tr_b16 (dumped, so that the host emulation of tests/hip_emul can restate it from a measurement, not from memory).
// Round 5: the SQ counters of k_attn_varlen_hd32_v2 (profiles/r5_pmc_sq_attention_*.json) say 4.75 cycles per VALU instruction
// on average; this program says which instructions cost what.   Build: hipcc --offload-arch=gfx950 -O2 scripts/vopbench.cpp -o vopbench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

// one kernel per instruction: 16 independent registers, `iters` x 16 instructions between two s_memtime reads
#define DEF_KERNEL32(NAME, ASM)                                                                                   \
    __global__ void NAME(uint64_t* out, int iters, float seed) {                                                  \
        float r[16];                                                                                              \
        for (int i = 0; i < 16; ++i) r[i] = seed + 0.001f * (float)(threadIdx.x + i);                             \
        float c1 = 0.999f, c2 = 0.0001f;                                                                          \
        const uint64_t t0 = __builtin_amdgcn_s_memtime();                                                         \
        for (int it = 0; it < iters; ++it) {                                                                      \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(r[i]) : "v"(c1), "v"(c2));     \
        }                                                                                                         \
        asm volatile("s_nop 0" ::: "memory");                                                                     \
        const uint64_t t1 = __builtin_amdgcn_s_memtime();                                                         \
        float s = 0;                                                                                              \
        for (int i = 0; i < 16; ++i) s += r[i];                                                                   \
        if (s == 12345.678f) out[1 << 20] = 1;                                                                    \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;         \
    }
#define DEF_KERNEL64(NAME, ASM)                                                                                   \
    __global__ void NAME(uint64_t* out, int iters, float seed) {                                                  \
        f2 r[16];                                                                                                 \
        for (int i = 0; i < 16; ++i) r[i] = f2{seed + 0.001f * (float)(threadIdx.x + i), seed};                   \
        f2 c1 = {0.999f, 0.999f}, c2 = {0.0001f, 0.0001f};                                                        \
        const uint64_t t0 = __builtin_amdgcn_s_memtime();                                                         \
        for (int it = 0; it < iters; ++it) {                                                                      \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(r[i]) : "v"(c1), "v"(c2));     \
        }                                                                                                         \
        asm volatile("s_nop 0" ::: "memory");                                                                     \
        const uint64_t t1 = __builtin_amdgcn_s_memtime();                                                         \
        float s = 0;                                                                                              \
        for (int i = 0; i < 16; ++i) s += r[i][0] + r[i][1];                                                      \
        if (s == 12345.678f) out[1 << 20] = 1;                                                                    \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;         \
    }

DEF_KERNEL32(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
DEF_KERNEL32(k_mul_f32, "v_mul_f32 %0, %0, %1")
DEF_KERNEL32(k_max3_f32, "v_max3_f32 %0, %0, %1, %2")
DEF_KERNEL32(k_exp_f32, "v_exp_f32 %0, %0")
DEF_KERNEL32(k_exp_f16, "v_exp_f16 %0, %0")
DEF_KERNEL32(k_exp_f16_sdwa_hi, "v_exp_f16_sdwa %0, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1")
DEF_KERNEL32(k_rcp_f32, "v_rcp_f32 %0, %0")
DEF_KERNEL32(k_cvt_pk_f16_f32, "v_cvt_pk_f16_f32 %0, %0, %1")
DEF_KERNEL32(k_cvt_pkrtz_f16_f32, "v_cvt_pkrtz_f16_f32 %0, %0, %1")
DEF_KERNEL32(k_pk_add_f16, "v_pk_add_f16 %0, %0, %1")
DEF_KERNEL32(k_pk_fma_f16, "v_pk_fma_f16 %0, %0, %1, %2")
DEF_KERNEL32(k_pk_max_f16, "v_pk_max_f16 %0, %0, %1")
DEF_KERNEL32(k_pk_mul_f16, "v_pk_mul_f16 %0, %0, %1")
DEF_KERNEL32(k_pk_lshl_b16, "v_pk_lshlrev_b16 %0, 10, %0")
DEF_KERNEL32(k_dot2_f32_f16, "v_dot2_f32_f16 %0, %1, %2, %0")
DEF_KERNEL32(k_dot2c_f32_f16, "v_dot2c_f32_f16 %0, %1, %2")
DEF_KERNEL32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEF_KERNEL32(k_cmp_cndmask, "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
DEF_KERNEL32(k_mov_dpp_row_shr, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
DEF_KERNEL32(k_add_f32, "v_add_f32 %0, %0, %1")
DEF_KERNEL32(k_fma_mix_lo, "v_fma_mixlo_f16 %0, %0, %1, %2")
DEF_KERNEL64(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %2")
DEF_KERNEL64(k_pk_add_f32, "v_pk_add_f32 %0, %0, %1")
DEF_KERNEL64(k_pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")

// permlane32_swap: two registers per instruction
__global__ void k_permlane32_swap(uint64_t* out, int iters, float seed) {
    float r[16];
    for (int i = 0; i < 16; ++i) r[i] = seed + (float)(threadIdx.x + i);
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r[i]), "+v"(r[i + 1]));
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r[i + 1]), "+v"(r[i]));
        }
    }
    asm volatile("s_nop 0" ::: "memory");
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += r[i];
    if (s == 12345.678f) out[1 << 20] = 1;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

// MFMA 32x32x16 f16: 4 independent accumulators (back-to-back issue) or, with DEP, one dependent chain
template <bool DEP>
__global__ void k_mfma32(uint64_t* out, int iters, float seed) {
    f16v acc[4];
    for (int a = 0; a < 4; ++a)
        for (int i = 0; i < 16; ++i) acc[a][i] = seed;
    h8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(seed + i); y[i] = (_Float16)(0.001f * threadIdx.x); }
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int a = DEP ? 0 : (i & 3);
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[a], 0, 0, 0);
        }
    }
    asm volatile("s_nop 0" ::: "memory");
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int a = 0; a < 4; ++a)
        for (int i = 0; i < 16; ++i) s += acc[a][i];
    if (s == 12345.678f) out[1 << 20] = 1;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

// MFMA + exp interleaved: does a v_exp_f32 stream hide under MFMAs of the same wave?  per iteration 4 MFMAs + NEXP exps
template <int NEXP>
__global__ void k_mfma_exp(uint64_t* out, int iters, float seed) {
    f16v acc[4];
    for (int a = 0; a < 4; ++a)
        for (int i = 0; i < 16; ++i) acc[a][i] = seed;
    float r[16];
    for (int i = 0; i < 16; ++i) r[i] = seed + 0.001f * (float)(threadIdx.x + i);
    h8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(seed + i); y[i] = (_Float16)(0.001f * threadIdx.x); }
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[a], 0, 0, 0);
#pragma unroll
            for (int e = 0; e < NEXP / 4; ++e) asm volatile("v_exp_f32 %0, %0" : "+v"(r[(a * (NEXP / 4) + e) & 15]));
        }
    }
    asm volatile("s_nop 0" ::: "memory");
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int a = 0; a < 4; ++a)
        for (int i = 0; i < 16; ++i) s += acc[a][i];
    for (int i = 0; i < 16; ++i) s += r[i];
    if (s == 12345.678f) out[1 << 20] = 1;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

// ds_read_b64_tr_b16 lane mapping: LDS holds halfword i at byte 2 i (value = i); lane l reads at byte address 8 l (its "row" = halfwords
// 4 l .. 4 l + 3); the four halfwords each lane receives are written out: value v came from lane v / 4, element v % 4.
__global__ void k_tr_probe(uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    const uint32_t addr = (uint32_t)(uintptr_t)lds + 8 * l;
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    u2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[4 * l + 0] = (uint16_t)(v[0] & 0xffff);
    out[4 * l + 1] = (uint16_t)(v[0] >> 16);
    out[4 * l + 2] = (uint16_t)(v[1] & 0xffff);
    out[4 * l + 3] = (uint16_t)(v[1] >> 16);
}

template <typename K>
static void run(const char* name, K kern, int per_instr_regs = 1) {
    uint64_t* d;
    hipMalloc(&d, ((1 << 20) + 16) * 8);
    const int iters = 256;
    for (int wps : {1, 2, 4}) {
        const int threads = 256 * wps, blocks = 256;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f);
        hipDeviceSynchronize();
        std::vector<uint64_t> h((size_t)blocks * threads / 64);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        double sum = 0, mn = 1e30, mx = 0;
        for (auto v : h) { sum += (double)v; mn = v < mn ? (double)v : mn; mx = v > mx ? (double)v : mx; }
        const double per = sum / h.size() / (iters * 16.0);
        // s_memtime ticks at 100 MHz on some parts: report raw ticks per instruction and ticks x waves-per-SIMD (= SIMD time per instruction)
        printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"ticks_per_instr_per_wave\": %.3f, \"min\": %.3f, \"max\": %.3f, \"simd_ticks_per_instr\": %.3f}\n", name, wps, per,
               mn / (iters * 16.0), mx / (iters * 16.0), per / wps);
        fflush(stdout);
    }
    hipFree(d);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("{\"device\": \"%s\", \"clock_khz\": %d, \"wall_clock_khz\": %d}\n", p.gcnArchName, p.clockRate, 0);
#define RUN(k) run(#k, k)
    RUN(k_fma_f32); RUN(k_mul_f32); RUN(k_add_f32); RUN(k_max3_f32); RUN(k_exp_f32); RUN(k_exp_f16); RUN(k_exp_f16_sdwa_hi); RUN(k_rcp_f32);
    RUN(k_cvt_pk_f16_f32); RUN(k_cvt_pkrtz_f16_f32); RUN(k_fma_mix_lo); RUN(k_pk_add_f16); RUN(k_pk_fma_f16); RUN(k_pk_max_f16); RUN(k_pk_mul_f16); RUN(k_pk_lshl_b16);
    RUN(k_dot2_f32_f16); RUN(k_dot2c_f32_f16); RUN(k_cndmask); RUN(k_cmp_cndmask); RUN(k_mov_dpp_row_shr); RUN(k_pk_fma_f32); RUN(k_pk_add_f32); RUN(k_pk_mul_f32);
    RUN(k_permlane32_swap);
    run("mfma_32x32x16_f16 independent x4", k_mfma32<false>);
    run("mfma_32x32x16_f16 dependent chain", k_mfma32<true>);
    run("4 mfma + 0 exp per iteration-quarter (x4)", k_mfma_exp<0>);
    run("4 mfma + 8 v_exp_f32", k_mfma_exp<8>);
    run("4 mfma + 16 v_exp_f32", k_mfma_exp<16>);
    run("4 mfma + 32 v_exp_f32", k_mfma_exp<32>);
    {
        uint16_t* d;
        hipMalloc(&d, 64 * 4 * 2);
        hipLaunchKernelGGL(k_tr_probe, dim3(1), dim3(64), 0, 0, d);
        hipDeviceSynchronize();
        uint16_t h[256];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("{\"probe\": \"ds_read_b64_tr_b16, lane l reads byte address 8 l of halfwords 0..255; entry = [source lane, source element] per received halfword\", \"lanes\": [");
        for (int l = 0; l < 64; ++l) {
            printf("%s[", l ? ", " : "");
            for (int k = 0; k < 4; ++k) printf("%s[%d, %d]", k ? ", " : "", h[4 * l + k] / 4, h[4 * l + k] % 4);
            printf("]");
        }
        printf("]}\n");
    }
    return 0;
}
