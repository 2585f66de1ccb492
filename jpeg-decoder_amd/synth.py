"""Deterministic synthetic baseline JPEGs (synth/zj_synth.c): the benchmark's and the tests' inputs."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "synth", "libzjsynth.so")
_lib = None

NATURAL, SMOOTH, NOISE = 0, 1, 2
# added to the content class: which table ids the luma component uses (the chroma components get the other id)
LUMA_DC_ID1, LUMA_AC_ID1, LUMA_QT_ID1 = 0x100, 0x200, 0x400


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


def encode_tiled(w, strip_h, reps, hs=2, vs=2, quality=90, content=NATURAL, seed=0, restart_interval=16) -> bytes:
    """A tall restart-marked image built from `reps` vertical repetitions of one encoded strip of
    `strip_h` rows: restart intervals are independent (byte aligned, predictors reset), so the strip's
    entropy-coded data can be repeated with the RSTn marker that belongs between two repetitions and the
    frame height patched.  Encoding a 16384x16384 image pixel by pixel takes a minute; this takes seconds and
    decodes to the strip's pixels repeated.  Requires whole restart intervals per strip, a number of
    intervals per strip that is a multiple of 8 (so that the RSTn numbering continues), and strip_h a
    multiple of the MCU height."""
    mh, mw = 8 * max(vs, 1), 8 * max(hs, 1)
    if restart_interval <= 0 or strip_h % mh or reps < 1 or strip_h * reps > 65535:
        raise ValueError("bad tiling parameters")
    mcus = ((w + mw - 1) // mw) * (strip_h // mh)
    if mcus % restart_interval or (mcus // restart_interval) % 8:
        raise ValueError("strip must hold a multiple of 8 whole restart intervals")
    strip = encode(w, strip_h, hs, vs, quality, content, seed, restart_interval)
    sof = strip.index(b"\xff\xc0")
    sos = strip.index(b"\xff\xda", sof)
    scan0 = sos + 2 + ((strip[sos + 2] << 8) | strip[sos + 3])
    if strip[-2:] != b"\xff\xd9":
        raise ValueError("unexpected end of the strip file")
    scan = strip[scan0:-2]
    head = bytearray(strip[:scan0])
    hh = strip_h * reps
    head[sof + 5], head[sof + 6] = hh >> 8, hh & 255
    return bytes(head) + b"\xff\xd7".join([scan] * reps) + b"\xff\xd9"
