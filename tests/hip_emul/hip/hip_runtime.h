// Stub of <hip/hip_runtime.h> for tests/hip_emul: lets the MFMA kernels of leann_amd/csrc compile as HOST code
// (x86, clang++) so that their index algebra can be executed lane by lane on the CPU.  Test infrastructure only.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __launch_bounds__(...)
#define __shared__
#define __align__(n) __attribute__((aligned(n)))

struct dim3 {
    unsigned x = 1, y = 1, z = 1;
};
using hipStream_t = void*;
enum hipError_t { hipSuccess = 0 };
inline const char* hipGetErrorString(hipError_t) { return "emulation"; }
inline float rsqrtf(float x) { return 1.0f / std::sqrt(x); }
inline int min(int a, int b) { return a < b ? a : b; }

#include "../emul.h"

#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) emul::mfma_32x32x16((a), (b), (c))
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_fmed3f(a, b, c) emul::med3((a), (b), (c))
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)
#define __shfl_xor(v, m) emul::shfl_xor((v), (m))
#define __syncthreads() emul::syncthreads()
#define threadIdx emul::tls.thread
#define blockIdx emul::tls.block
