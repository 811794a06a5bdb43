// Stub of <hip/hip_fp16.h> for tests/hip_emul (the emulated kernels only use __half as an opaque 2-byte type).
#pragma once
struct __half {
    _Float16 v;
};
