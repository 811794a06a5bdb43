// mfma_sustained.cpp -- what does this MI355X sustain on v_mfma_f32_32x32x16_f16 with NOTHING else in the way?  Every roofline row of the
// encoder kernels is quoted against the 2.5 PFLOP/s dense-fp16 figure (256 CUs x 4 SIMDs x 1024 flop per cycle x 2.4 GHz); the benchmark's
// timed steps run at the socket's 1400 W cap at 1.6-1.9 GHz.  This program measures the ceiling the cap leaves: a register-only
// MFMA loop on the whole chip (operands and accumulators in registers, no LDS, no memory), for seconds, launch after launch, with
//   * operands of random fp16 bits (|x| < 2: realistic toggling of the multiplier arrays) or all zeros (the least a matrix pipe can draw),
//   * 1, 2 or 4 waves per SIMD (the layer tail runs 1, the QKV kernel 2),
//   * the matrix pipe fully busy (back-to-back independent MFMAs) or busy a fraction of the time (s_nop padding between MFMAs: duty 0.34, 0.44).
// Prints one JSON line per setting: TFLOP/s (event-timed, median launch of the last half), and the per-launch spread.  scripts/mfma_sustained.py wraps
// it with the clock / power sampler of bench.py.
//     hipcc --offload-arch=gfx950 -O2 -std=c++17 scripts/mfma_sustained.cpp -o leann_amd/lib/bin/mfma_sustained;   mfma_sustained [seconds [setting index]]
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Winline-asm"

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));   \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

// PAD = s_nop groups between MFMAs: 0 = back to back (four independent accumulators: the pipe never waits); an 8-pass MFMA occupies the pipe 32 cycles,
// PAD = 1 / 2: an s_nop 15 / s_nop 10 behind every MFMA (duty 0.34 / 0.44 at 2.4 GHz)
template <int PAD>
__global__ __launch_bounds__(256, 4) void k_mfma(float* out, int iters, unsigned seed, int zeros) {
    f16v acc[4];
    for (int a = 0; a < 4; ++a)
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.0f;
    h8 x[2], y[2];
    unsigned s = seed ^ (blockIdx.x * 2654435761u) ^ (threadIdx.x * 40503u);
    for (int j = 0; j < 2; ++j)
        for (int i = 0; i < 8; ++i) {
            s = s * 1664525u + 1013904223u;
            const unsigned short bx = zeros ? 0 : (unsigned short)(((s >> 16) & 0x83ffu) | 0x3800u | ((s >> 3) & 0x0400u));  // sign, random mantissa, exponent 14 or 15: 0.5 <= |x| < 2
            s = s * 1664525u + 1013904223u;
            const unsigned short by = zeros ? 0 : (unsigned short)(((s >> 16) & 0x83ffu) | 0x3000u | ((s >> 3) & 0x0400u));  // 0.125 <= |y| < 0.5
            x[j][i] = __builtin_bit_cast(_Float16, bx);
            y[j][i] = __builtin_bit_cast(_Float16, by);
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[i & 1], y[(i >> 1) & 1], acc[i & 3], 0, 0, 0);
            // (s_nop N holds the wave's issue for N + 1 quad-cycles and issues while the MFMA in front of it is still in the pipe.  Measured at 2.4 GHz: s_nop 10 behind every MFMA
            // = 0.44 of the pipe's rate, s_nop 15 = 0.34; three s_nop 15 = 0.14.  The wrapper reports the duty from the sampled clock.)
            if constexpr (PAD == 1) asm volatile("s_nop 15" ::: "memory");
            if constexpr (PAD == 2) asm volatile("s_nop 10" ::: "memory");
        }
        if ((it & 63) == 63)  // keep the sums finite over millions of iterations: halve them now and then (VALU work of 1 / 1024 of the MFMAs)
            for (int a = 0; a < 4; ++a) acc[a] *= 0.0009765625f;
    }
    float r = 0;
    for (int a = 0; a < 4; ++a)
        for (int i = 0; i < 16; ++i) r += acc[a][i];
    if (r == 12345.678f) out[0] = r;
}

// ---- what the rest of a real kernel costs, as matrix-pipe rate at the power cap: the same MFMA stream with the layer tail's side traffic added piece by piece ----
//   LDSREAD: the A operand of every MFMA comes from LDS (one ds_read_b128 per MFMA and wave, requested four MFMAs ahead): k_layer_tail_h384's weight fragments
//   DMA:     one global_load_lds_dwordx4 per four MFMAs and wave (256 B per MFMA: the tail's 48 KB of W1 / W2 slabs per 192 MFMAs of a workgroup) from a 2.6 MB buffer
//            every CU walks in the same order (L2 hits, as the weight stream)
//   VALU:    three v_fma_f32 per MFMA (the GELU micro-operations between the tail's MFMAs)
//   REUSE:   2 = every LDS fragment feeds TWO MFMAs and the DMA stream runs at half the rate: what a kernel with two token blocks per wave would see (it does not fit
//            the register file with fp32 accumulators at hidden 384: DESIGN 8) -- the worth of halving the operand bytes per flop
template <int LDSREAD, int DMA, int VALU, int REUSE = 1>
__global__ __launch_bounds__(256, 1) void k_mix(float* out, const unsigned char* __restrict__ wbuf, unsigned wbytes, int iters, unsigned seed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // [0, 64 K): fragments; [64 K, 80 K): DMA landing area
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned s = seed ^ (blockIdx.x * 2654435761u) ^ (threadIdx.x * 40503u);
    auto rnd16 = [&](unsigned e) {
        s = s * 1664525u + 1013904223u;
        return (unsigned short)(((s >> 16) & 0x83ffu) | e | ((s >> 3) & 0x0400u));
    };
    for (int i = threadIdx.x; i < 32768; i += 256) ((unsigned short*)lds)[i] = rnd16(0x3800u);
    __syncthreads();
    f16v acc[4];
    for (int a = 0; a < 4; ++a)
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.0f;
    h8 x[2], y[2], frag[4];
    for (int j = 0; j < 2; ++j)
        for (int i = 0; i < 8; ++i) {
            x[j][i] = __builtin_bit_cast(_Float16, rnd16(0x3800u));
            y[j][i] = __builtin_bit_cast(_Float16, rnd16(0x3000u));
        }
    for (int j = 0; j < 4; ++j) frag[j] = *(const h8*)(lds + 1024 * j + 16 * lane);
    float v0 = 1.0f + lane, v1 = 0.5f, v2 = 0.25f;
    unsigned roff = 4096, goff = (unsigned)(wave * 1024 + lane * 16);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(LDSREAD ? frag[i & 3] : x[i & 1], y[(i >> 1) & 1], acc[i & 3], 0, 0, 0);
            if constexpr (LDSREAD) {
                if (REUSE == 1 || (i & 1)) frag[i & 3] = *(const h8*)(lds + ((roff + 1024 * i) & 0xffffu) + 16 * lane);
                else frag[i & 3] = frag[(i + 3) & 3];  // (REUSE 2: the fragment read one slot earlier serves a second MFMA: a register move the compiler folds into the operand choice)
            }
            if constexpr (DMA) {
                if ((i & (4 * REUSE - 1)) == 0) {
                    const unsigned m0v = __builtin_amdgcn_readfirstlane(65536u + (unsigned)wave * 4096u + (unsigned)(i >> 2) * 1024u);
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(goff), "s"(wbuf), "s"(m0v) : "memory", "m0");
                    goff += 4096;
                    if (goff >= wbytes) goff -= wbytes;
                }
            }
            if constexpr (VALU) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(v1), "v"(v2));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v1) : "v"(v2), "v"(v0));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v2) : "v"(v0), "v"(v1));
            }
        }
        roff += 16384;
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        if ((it & 63) == 63)
            for (int a = 0; a < 4; ++a) acc[a] *= 0.0009765625f;
    }
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float r = v0 + v1 + v2;
    for (int a = 0; a < 4; ++a)
        for (int i = 0; i < 16; ++i) r += acc[a][i];
    if (r == 12345.678f) out[0] = r;
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
    CK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float* d_out;
    CK(hipMalloc(&d_out, 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    struct Setting { const char* name; int pad, zeros, waves_per_simd, mix; };
    const Setting settings[] = {{"random operands, back to back", 0, 0, 1}, {"random operands, back to back", 0, 0, 2}, {"random operands, back to back", 0, 0, 4},
                                {"zero operands, back to back", 0, 1, 1},   {"random operands, s_nop 10 behind every MFMA", 2, 0, 1},    {"random operands, s_nop 15 behind every MFMA", 1, 0, 1},
                                {"mix: MFMA only (the mixing kernel's own baseline)", 0, 0, 1, 1}, {"mix: + A operand from LDS (ds_read_b128 per MFMA)", 0, 0, 1, 2},
                                {"mix: + A from LDS + LDS-DMA stream from L2 (256 B per MFMA and wave)", 0, 0, 1, 3}, {"mix: + three v_fma_f32 per MFMA", 0, 0, 1, 4},
                                {"mix: + A from LDS + LDS-DMA stream + three v_fma_f32 per MFMA", 0, 0, 1, 5},
                                {"mix: all three, every LDS fragment and DMA byte serving TWO MFMAs", 0, 0, 1, 6}};
    const unsigned wbytes = 2654208;  // 648 x 4096 B: the layer tail's W_o + W1 + W2 images
    unsigned char* d_w;
    CK(hipMalloc(&d_w, wbytes + 8192));
    CK(hipMemset(d_w, 0x3c, wbytes + 8192));  // fp16 0x3c3c = 1.06: the stream's bytes are never multiplied, only moved
    const size_t mix_lds = 65536 + 16384;
    CK(hipFuncSetAttribute((const void*)k_mix<0, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mix_lds));
    CK(hipFuncSetAttribute((const void*)k_mix<1, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mix_lds));
    CK(hipFuncSetAttribute((const void*)k_mix<1, 1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mix_lds));
    CK(hipFuncSetAttribute((const void*)k_mix<0, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mix_lds));
    CK(hipFuncSetAttribute((const void*)k_mix<1, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mix_lds));
    CK(hipFuncSetAttribute((const void*)k_mix<1, 1, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mix_lds));
    const int only = argc > 2 ? atoi(argv[2]) : -1;  // one setting per process (scripts/mfma_sustained.py samples clocks / power per setting)
    int index = -1;
    for (const Setting& st : settings) {
        if (++index != only && only >= 0) continue;
        const int blocks = cus * st.waves_per_simd;  // 256 threads = 4 waves = one per SIMD; waves_per_simd blocks per CU
        const int iters = st.pad == 1 ? 20000 : st.pad == 2 ? 30000 : 48000 / st.waves_per_simd;  // ~10-25 ms per launch
        auto launch = [&]() {
            if (st.mix == 1) hipLaunchKernelGGL((k_mix<0, 0, 0>), dim3(blocks), dim3(256), mix_lds, 0, d_out, d_w, wbytes, iters, 12345u);
            else if (st.mix == 2) hipLaunchKernelGGL((k_mix<1, 0, 0>), dim3(blocks), dim3(256), mix_lds, 0, d_out, d_w, wbytes, iters, 12345u);
            else if (st.mix == 3) hipLaunchKernelGGL((k_mix<1, 1, 0>), dim3(blocks), dim3(256), mix_lds, 0, d_out, d_w, wbytes, iters, 12345u);
            else if (st.mix == 4) hipLaunchKernelGGL((k_mix<0, 0, 1>), dim3(blocks), dim3(256), mix_lds, 0, d_out, d_w, wbytes, iters, 12345u);
            else if (st.mix == 5) hipLaunchKernelGGL((k_mix<1, 1, 1>), dim3(blocks), dim3(256), mix_lds, 0, d_out, d_w, wbytes, iters, 12345u);
            else if (st.mix == 6) hipLaunchKernelGGL((k_mix<1, 1, 1, 2>), dim3(blocks), dim3(256), mix_lds, 0, d_out, d_w, wbytes, iters, 12345u);
            else if (st.pad == 0) hipLaunchKernelGGL(k_mfma<0>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 12345u, st.zeros);
            else if (st.pad == 1) hipLaunchKernelGGL(k_mfma<1>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 12345u, st.zeros);
            else hipLaunchKernelGGL(k_mfma<2>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 12345u, st.zeros);
        };
        launch();
        CK(hipDeviceSynchronize());
        std::vector<float> ms;
        double total = 0;
        while (total < seconds * 1e3) {
            CK(hipEventRecord(e0, 0));
            launch();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float t;
            CK(hipEventElapsedTime(&t, e0, e1));
            ms.push_back(t);
            total += t;
        }
        const double flop = (double)blocks * 4 * iters * 16 * 32768.0;  // per launch: waves x iterations x 16 MFMAs x 2 x 32 x 32 x 16
        std::vector<float> tail(ms.begin() + ms.size() / 2, ms.end());  // the second half: the clock has settled under the load
        std::sort(tail.begin(), tail.end());
        const double med = tail[tail.size() / 2];
        printf("{\"setting\": \"%s\", \"waves_per_simd\": %d, \"zero_operands\": %d, \"launches\": %zu, \"ms_per_launch_median_second_half\": %.3f, \"TFLOPs\": %.1f, "
               "\"TFLOPs_first_launch\": %.1f, \"of_2500\": %.3f, \"matrix_pipe_cycles_per_launch_and_simd\": %.0f, \"implied_clock_MHz_if_the_pipe_never_idles\": %.0f}\n",
               st.name, st.waves_per_simd, st.zeros, ms.size(), med, flop / (med * 1e-3) / 1e12, flop / (ms[0] * 1e-3) / 1e12, flop / (med * 1e-3) / 2.5e15,
               (double)st.waves_per_simd * iters * 16 * 32.0, st.pad == 0 ? (double)st.waves_per_simd * iters * 16 * 32.0 / (med * 1e-3) / 1e6 : 0.0);
        fflush(stdout);
    }
    return 0;
}
