"""Raw device-pointer plumbing between the C ABI and PyTorch-ROCm (no compute here)."""
from __future__ import annotations

import ctypes as C

_hip = None


def _hiprt():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        _hip.hipMemcpyAsync.restype = C.c_int
    return _hip


def copy_d2d(dst_ptr: int, src_ptr: int, nbytes: int, stream_ptr: int = 0) -> None:
    """hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToDevice, stream)."""
    if nbytes == 0:
        return
    rc = _hiprt().hipMemcpyAsync(C.c_void_p(dst_ptr), C.c_void_p(src_ptr), nbytes, 3, C.c_void_p(stream_ptr))
    if rc != 0:
        raise RuntimeError(f"hipMemcpyAsync failed with hipError {rc}")


class _DevArray:
    def __init__(self, ptr: int, shape, typestr: str):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False), "version": 2}


_TYPESTR = {"int32": "<i4", "float32": "<f4", "int64": "<i8", "float16": "<f2"}


def as_tensor(ptr: int, shape, dtype: str = "int32", device="cuda"):
    """Zero-copy torch view of library-owned device memory (valid until the library reuses it)."""
    import torch

    return torch.as_tensor(_DevArray(ptr, shape, _TYPESTR[dtype]), device=device)
