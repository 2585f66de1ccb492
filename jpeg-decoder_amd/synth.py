"""Deterministic synthetic baseline JPEGs (synth/zj_synth.c): the benchmark's and the tests' inputs."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "synth", "libzjsynth.so")
_lib = None

NATURAL, SMOOTH, NOISE = 0, 1, 2


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build()")
        L = C.CDLL(LIB_PATH)
        L.zj_synth_encode.restype = C.POINTER(C.c_uint8)
        L.zj_synth_encode.argtypes = [C.c_int] * 6 + [C.c_uint64, C.c_int, C.POINTER(C.c_size_t)]
        L.zj_synth_free.argtypes = [C.POINTER(C.c_uint8)]
        L.zj_synth_free.restype = None
        _lib = L
    return _lib


def encode(w, h, hs=2, vs=2, quality=90, content=NATURAL, seed=0, restart_interval=0) -> bytes:
    """One synthetic image as JPEG file bytes.  (hs, vs) = luma sampling: (2,2) is 4:2:0, (1,1) is 4:4:4,
    (0,0) is a one-component (grayscale) file that codes the luma plane of the same content."""
    n = C.c_size_t()
    p = lib().zj_synth_encode(w, h, hs, vs, quality, content, seed, restart_interval, C.byref(n))
    if not p:
        raise ValueError("bad synth parameters")
    try:
        return C.string_at(p, n.value)
    finally:
        lib().zj_synth_free(p)
