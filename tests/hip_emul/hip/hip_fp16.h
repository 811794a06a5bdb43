// Stub of <hip/hip_fp16.h> for tests/hip_emul: just enough of the half API for the kernels that are emulated.
#pragma once
#include <cstdint>
struct __half {
    _Float16 v;
};
struct __half2 {
    _Float16 x, y;
};
struct float2 {
    float x, y;
};
struct float4 {
    float x, y, z, w;
};
struct uint4 {
    uint32_t x, y, z, w;
};
inline uint4 make_uint4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return uint4{a, b, c, d}; }
inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
inline float2 __half22float2(__half2 h) { return float2{(float)h.x, (float)h.y}; }
inline __half2 __floats2half2_rn(float a, float b) { return __half2{(_Float16)a, (_Float16)b}; }
inline float __half2float(__half h) { return (float)h.v; }
inline __half __float2half_rn(float f) { return __half{(_Float16)f}; }
