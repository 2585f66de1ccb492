"""ctypes front end of the CPU checker (oracle/zj_oracle.c).

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg of
bench.py -- never by the shipped package.  Parity is unpinned by the reference itself (it has no
tests or golden vectors and cannot be built here); see the header of zj_oracle.c.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libzjoracle.so")

OK, E_FORMAT, E_UNSUPPORTED, E_HUFF, E_COEF, E_TRUNC = 0, -1, -2, -3, -4, -5


class Info(C.Structure):
    _fields_ = [
        ("status", C.c_int32),
        ("w", C.c_int32), ("h", C.c_int32), ("ncomp", C.c_int32),
        ("hs", C.c_int32 * 3), ("vs", C.c_int32 * 3),
        ("mcux", C.c_int32), ("mcuy", C.c_int32), ("blocks_per_mcu", C.c_int32),
        ("restart_interval", C.c_int32),
        ("nblocks", C.c_int64),
        ("scan_off", C.c_int64), ("scan_len", C.c_int64), ("scan_consumed", C.c_int64),
        ("nsymbols", C.c_uint64), ("nshort", C.c_uint64), ("nff00", C.c_uint64),
        ("err_block", C.c_int64),
    ]


def build(force=False):
    """Compile the checker with gcc (seconds).  Building the checker is not using it."""
    src = os.path.join(_HERE, "zj_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        u8p, i16p, f32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int16), C.POINTER(C.c_float)
        L.zjo_probe.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Info)]
        L.zjo_probe.restype = C.c_int
        L.zjo_decode.argtypes = [C.c_char_p, C.c_size_t, i16p, i16p, u8p, C.POINTER(Info)]
        L.zjo_decode.restype = C.c_int
        L.zjo_blocks_to_rgb.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, i16p, u8p]
        L.zjo_blocks_to_rgb.restype = C.c_int
        L.zjo_idct.argtypes = [f32p, f32p]
        L.zjo_idct.restype = None
        L.zjo_sign_code.argtypes = [C.c_uint32, C.c_uint16]
        L.zjo_sign_code.restype = C.c_int16
        L.zjo_ycbcr_to_rgb.argtypes = [C.c_float, C.c_float, C.c_float, u8p]
        L.zjo_ycbcr_to_rgb.restype = None
        L.zjo_tables.argtypes = [u8p, f32p]
        L.zjo_tables.restype = None
        L.zjo_write_ppm.argtypes = [C.c_char_p, C.c_int, C.c_int, u8p]
        L.zjo_write_ppm.restype = C.c_int
        _lib = L
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def probe(data: bytes) -> Info:
    info = Info()
    lib().zjo_probe(data, len(data), C.byref(info))
    return info


class Decoded:
    """Result of a full oracle decode."""

    def __init__(self, info, qcoef, dqcoef, rgb):
        self.info, self.qcoef, self.dqcoef, self.rgb = info, qcoef, dqcoef, rgb
        self.status = info.status


def decode(data: bytes, want_rgb=True, want_coef=True) -> Decoded:
    """Decode a JPEG file image held in memory.

    qcoef:  int16 [nblocks, 64], zig-zag order, before dequantisation (DC after prediction)
    dqcoef: int16 [nblocks, 64], natural order, dequantised (the reference's `blocks`)
    rgb:    uint8 [h, w, 3]
    """
    info = probe(data)
    if info.status != OK:
        return Decoded(info, None, None, None)
    nb = info.nblocks
    q = np.empty((nb, 64), np.int16) if want_coef else None
    d = np.empty((nb, 64), np.int16) if want_coef else None
    rgb = np.zeros((info.h, info.w, 3), np.uint8) if want_rgb else None
    info2 = Info()
    lib().zjo_decode(data, len(data),
                     _p(q, C.c_int16) if want_coef else None,
                     _p(d, C.c_int16) if want_coef else None,
                     _p(rgb, C.c_uint8) if want_rgb else None, C.byref(info2))
    if info2.status != OK:
        # an entropy error (HUFF / COEF / TRUNC, where the reference panics): the blocks before info.err_block are complete
        keep = want_coef and info2.status in (E_HUFF, E_COEF, E_TRUNC) and info2.err_block >= 0
        return Decoded(info2, q[: info2.err_block] if keep else None, d[: info2.err_block] if keep else None, None)
    return Decoded(info2, q, d, rgb)


def blocks_to_rgb(w, h, hs0, vs0, dqcoef: np.ndarray) -> np.ndarray:
    """Stage 2+3 only: dequantised natural-order blocks (scan order) -> RGB."""
    dq = np.ascontiguousarray(dqcoef, np.int16)
    mcux = (w + 8 * hs0 - 1) // (8 * hs0)
    mcuy = (h + 8 * vs0 - 1) // (8 * vs0)
    assert dq.size == mcux * mcuy * (hs0 * vs0 + 2) * 64
    rgb = np.zeros((h, w, 3), np.uint8)
    lib().zjo_blocks_to_rgb(w, h, hs0, vs0, _p(dq, C.c_int16), _p(rgb, C.c_uint8))
    return rgb


def idct(block: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(block, np.float32).reshape(64)
    out = np.empty(64, np.float32)
    lib().zjo_idct(_p(a, C.c_float), _p(out, C.c_float))
    return out


def sign_code(n, code):
    return int(lib().zjo_sign_code(n, code))


def ycbcr_to_rgb(y, cb, cr):
    px = (C.c_uint8 * 3)()
    lib().zjo_ycbcr_to_rgb(y, cb, cr, px)
    return tuple(px)


def tables():
    nat2zz = np.empty(64, np.uint8)
    cos = np.empty(64, np.float32)
    lib().zjo_tables(_p(nat2zz, C.c_uint8), _p(cos, C.c_float))
    return nat2zz, cos


def write_ppm(path, rgb: np.ndarray):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w, _ = rgb.shape
    return lib().zjo_write_ppm(path.encode(), w, h, _p(rgb, C.c_uint8))
