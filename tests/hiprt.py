"""Minimal ctypes access to the HIP runtime for tests that need raw device buffers (no torch:
importing torch after liborbhip.so would bring a second copy of the HIP runtime into the process)."""
import ctypes as C

import numpy as np

_rt = None


def rt():
    global _rt
    if _rt is None:
        from orbhip import capi
        capi.load()                       # liborbhip.so pulls in libamdhip64.so.7
        L = C.CDLL("libamdhip64.so.7")
        L.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        L.hipFree.argtypes = [C.c_void_p]
        L.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        L.hipDeviceSynchronize.argtypes = []
        _rt = L
    return _rt


class DevBuf:
    def __init__(self, nbytes):
        self.ptr = C.c_void_p()
        self.nbytes = nbytes
        assert rt().hipMalloc(C.byref(self.ptr), max(nbytes, 16)) == 0
        assert rt().hipMemset(self.ptr, 0, max(nbytes, 16)) == 0

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        assert rt().hipMemcpy(b.ptr, a.ctypes.data, a.nbytes, 1) == 0
        return b

    def to_numpy(self, dtype, shape):
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        assert rt().hipDeviceSynchronize() == 0
        assert rt().hipMemcpy(out.ctypes.data, self.ptr, out.nbytes, 2) == 0
        return out

    def free(self):
        if self.ptr:
            rt().hipFree(self.ptr)
            self.ptr = C.c_void_p()
