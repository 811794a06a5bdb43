// Stub of <hip/hip_runtime.h> for the FULL host build of libleann_mi355x (tests/hip_emul/build_emul_lib.py):
// device memory = host memory, a launch runs its blocks one after another with one OS thread per lane, wave
// collectives / __syncthreads() are barriers, atomics are host atomics.  Test infrastructure only -- the product
// library is the gfx950 build and has no CPU path.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __align__(n) __attribute__((aligned(n)))
#define __shared__  // raw-source harness (run_kernels.cpp); the full-library build substitutes it textually

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 {
    float x, y;
};
struct float4 {
    float x, y, z, w;
};
struct uint2 {
    uint32_t x, y;
};
struct uint4 {
    uint32_t x, y, z, w;
};
inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
inline uint4 make_uint4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return uint4{a, b, c, d}; }
struct __half {
    _Float16 v;
};
struct __half2 {
    _Float16 x, y;
};
inline float2 __half22float2(__half2 h) { return float2{(float)h.x, (float)h.y}; }
inline __half2 __floats2half2_rn(float a, float b) { return __half2{(_Float16)a, (_Float16)b}; }
inline float __half2float(__half h) { return (float)h.v; }
inline __half __float2half(float f) { return __half{(_Float16)f}; }
inline __half __float2half_rn(float f) { return __half{(_Float16)f}; }

// ---------------------------------------------------------------- runtime API (host memory, synchronous)
using hipStream_t = void*;
using hipEvent_t = void*;
enum hipError_t { hipSuccess = 0, hipErrorInvalidValue = 1 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize };
enum hipDeviceAttribute_t { hipDeviceAttributeWallClockRate, hipDeviceAttributeMultiprocessorCount };
inline const char* hipGetErrorString(hipError_t) { return "emulation"; }
inline hipError_t hipMalloc(void** p, size_t n) {
    *p = nullptr;  // exact size (AddressSanitizer then sees every byte past the end), 256-byte aligned like hipMalloc
    return posix_memalign(p, 256, n ? n : 1) == 0 ? hipSuccess : hipErrorInvalidValue;
}
template <class T>
inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n) { return hipMalloc(p, n); }
inline hipError_t hipHostFree(void* p) { return hipFree(p); }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t) { return hipMemcpy(d, s, n, k); }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { if (n) std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { if (n) std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 100000; return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { static char token; *e = &token; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }

#include "../../emul_full.h"

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    emul::launch(dim3(grid), dim3(block), (size_t)(shmem), [&] { (kernel)(__VA_ARGS__); })
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) emul::mfma_32x32x16((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x8f16(a, b, c, x, y, z) emul::mfma_32x32x8((a), (b), (c))
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_fmed3f(a, b, c) emul::med3((a), (b), (c))
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)
#define LM_EMULATED_DEVICE 1  // kernels with DMA / counted waits select their host form on this
#define LM_KEEP_LOCAL(v) ((void)0)
#define LM_ONE_WAVE_PER_SIMD
#define LM_TWO_WAVES_PER_SIMD
#define LM_WAVE_SYNC() emul::wave_sync()
inline float rsqrtf(float x) { return 1.0f / std::sqrt(x); }
#define threadIdx emul::tls.thread
#define blockIdx emul::tls.block
#define blockDim emul::tls.bdim
#define gridDim emul::tls.gdim
#define __syncthreads() emul::syncthreads()
#define __ballot(p) emul::ballot(p)
#define __shfl_up(v, d) emul::shfl_up((v), (d))
#define __shfl(v, s) emul::shfl_idx((v), (s))
#define __shfl_xor(...) emul::shfl_xor(__VA_ARGS__)
#define __popc(x) __builtin_popcount(x)
#define __popcll(x) __builtin_popcountll(x)
#define __ffs(x) __builtin_ffs(x)
#define __ffsll(x) __builtin_ffsll(x)
#define atomicOr(p, v) emul::atomic_or((p), (v))
#define atomicMin(p, v) emul::atomic_min((p), (v))
#define atomicAdd(p, v) emul::atomic_add((p), (v))
#define wall_clock64() emul::wall_clock()
using std::max;
using std::min;
