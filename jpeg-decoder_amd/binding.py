"""ctypes binding of the C ABI declared in include/zj.h (libzj.so, built in-tree by build.py).

The library is the product; this module only marshals arguments.  Loading fails loudly when the
shared object is missing -- there is no CPU fallback for the decode path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libzj.so")

ZJ_OK, ZJ_E_INVALID, ZJ_E_FORMAT, ZJ_E_UNSUPPORTED, ZJ_E_DEVICE, ZJ_E_NOMEM, ZJ_E_IO, ZJ_E_CHAIN = 0, -1, -2, -3, -4, -5, -6, -7
ZJ_ST_BAD_CODE, ZJ_ST_COEF_OVERFLOW, ZJ_ST_TRUNCATED, ZJ_ST_UNSUPPORTED = 1, 2, 4, 8
ZJ_MEM_HOST, ZJ_MEM_DEVICE = 0, 1
ZJ_FLAG_NO_SECOND_LEVEL = 0x1000000
ZJ_FLAG_HOST_MARKER_SCAN = 0x2000000
ZJ_NUM_KERNELS = 9
ZJ_NOALLOC_MAX_BYTES = 64 << 20


class zj_huff(C.Structure):
    _fields_ = [("counts", C.c_uint8 * 16), ("symbols", C.c_uint8 * 256)]


class zj_image(C.Structure):
    _fields_ = [
        ("scan", C.c_void_p), ("scan_len", C.c_size_t),
        ("width", C.c_uint16), ("height", C.c_uint16),
        ("ncomp", C.c_uint8),
        ("hsamp", C.c_uint8 * 3), ("vsamp", C.c_uint8 * 3),
        ("tq", C.c_uint8 * 3), ("td", C.c_uint8 * 3), ("ta", C.c_uint8 * 3),
        ("restart_interval", C.c_uint16),
        ("qt", C.c_void_p * 4),
        ("ht", (C.c_void_p * 2) * 2),
    ]


class zj_parsed(C.Structure):
    _fields_ = [
        ("image", zj_image),
        ("qt_store", (C.c_uint8 * 64) * 4),
        ("ht_store", (zj_huff * 2) * 2),
        ("n_restart_markers", C.c_uint32),
    ]


class zj_range_report(C.Structure):
    _fields_ = [("start_delta", C.c_int32), ("start_k", C.c_uint32), ("end_delta", C.c_int32), ("end_k", C.c_uint32),
                ("mcus", C.c_uint32), ("dc_sum", C.c_int16 * 3), ("reserved", C.c_uint16), ("reserved2", C.c_uint32)]


class zj_output(C.Structure):
    _fields_ = [("rgb", C.c_void_p), ("rgb_stride", C.c_size_t), ("qcoef", C.c_void_p), ("dqcoef", C.c_void_p),
                ("mem", C.c_int)]


_lib = None

_EXPORTS = [
    "zj_parse", "zj_image_blocks", "zj_image_mcus", "zj_restart_offsets", "zj_image_band", "zj_write_ppm",
    "zj_device_count", "zj_create", "zj_destroy", "zj_last_error",
    "zj_batch_create", "zj_batch_destroy", "zj_batch_decode", "zj_batch_finish", "zj_batch_qcoef",
    "zj_batch_export_coef", "zj_batch_set_profiling", "zj_batch_kernel_times", "zj_kernel_name", "zj_batch_bytes",
    "zj_batch_segments",
    "zj_decode_batch",
    "zj_workspace_estimate", "zj_workspace_create", "zj_workspace_destroy", "zj_batch_create_in", "zj_batch_memory",
    "zj_batch_counters", "zj_batch_create_rows",
    "zj_decode_batch_noalloc",
    "zj_batch_decode_entropy", "zj_batch_decode_pixels",
    "zj_batch_create_range", "zj_batch_range_front", "zj_batch_range_resolve", "zj_batch_range_decode",
]


def lib():
    """Load libzj.so (once) and declare the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # torch ships its own HIP runtime; it has to be in the process before libzj.so is mapped so that
        # both use one runtime (two runtimes in one process cannot both open the GPU)
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the decode path has no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp, sz, u32p = C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)
    L.zj_parse.argtypes = [vp, sz, C.POINTER(zj_parsed)]; L.zj_parse.restype = C.c_int
    L.zj_image_blocks.argtypes = [C.POINTER(zj_image)]; L.zj_image_blocks.restype = C.c_uint32
    L.zj_image_mcus.argtypes = [C.POINTER(zj_image)]; L.zj_image_mcus.restype = C.c_uint32
    L.zj_restart_offsets.argtypes = [C.POINTER(zj_image), C.POINTER(sz), sz]; L.zj_restart_offsets.restype = sz
    L.zj_image_band.argtypes = [C.POINTER(zj_image), C.c_uint32, C.c_uint32, C.POINTER(zj_image)]; L.zj_image_band.restype = C.c_int
    L.zj_write_ppm.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, vp, sz]; L.zj_write_ppm.restype = C.c_int
    L.zj_device_count.argtypes = []; L.zj_device_count.restype = C.c_int
    L.zj_create.argtypes = [C.c_int, vp]; L.zj_create.restype = vp
    L.zj_destroy.argtypes = [vp]; L.zj_destroy.restype = None
    L.zj_last_error.argtypes = [vp]; L.zj_last_error.restype = C.c_char_p
    L.zj_batch_create.argtypes = [vp, C.POINTER(zj_image), sz, C.c_uint32]; L.zj_batch_create.restype = vp
    L.zj_batch_destroy.argtypes = [vp]; L.zj_batch_destroy.restype = None
    L.zj_batch_decode.argtypes = [vp, C.POINTER(vp), C.POINTER(sz)]; L.zj_batch_decode.restype = C.c_int
    L.zj_batch_finish.argtypes = [vp, u32p]; L.zj_batch_finish.restype = C.c_int
    L.zj_batch_qcoef.argtypes = [vp, sz]; L.zj_batch_qcoef.restype = vp
    L.zj_batch_export_coef.argtypes = [vp, sz, vp, vp, C.c_int]; L.zj_batch_export_coef.restype = C.c_int
    L.zj_batch_set_profiling.argtypes = [vp, C.c_int]; L.zj_batch_set_profiling.restype = C.c_int
    L.zj_batch_kernel_times.argtypes = [vp, C.POINTER(C.c_float)]; L.zj_batch_kernel_times.restype = C.c_int
    L.zj_kernel_name.argtypes = [C.c_int]; L.zj_kernel_name.restype = C.c_char_p
    L.zj_batch_bytes.argtypes = [vp] + [C.POINTER(C.c_uint64)] * 4; L.zj_batch_bytes.restype = C.c_int
    L.zj_batch_segments.argtypes = [vp, sz, C.POINTER(C.c_uint64), sz]; L.zj_batch_segments.restype = sz
    L.zj_decode_batch.argtypes = [vp, C.POINTER(zj_image), sz, C.POINTER(zj_output), u32p]; L.zj_decode_batch.restype = C.c_int
    L.zj_workspace_estimate.argtypes = [C.POINTER(zj_image), sz, C.POINTER(sz), C.POINTER(sz)]; L.zj_workspace_estimate.restype = C.c_int
    L.zj_workspace_create.argtypes = [vp, sz, sz, C.c_uint32]; L.zj_workspace_create.restype = vp
    L.zj_workspace_destroy.argtypes = [vp]; L.zj_workspace_destroy.restype = None
    L.zj_batch_create_in.argtypes = [vp, C.c_uint32, C.POINTER(zj_image), sz, C.c_uint32]; L.zj_batch_create_in.restype = vp
    L.zj_batch_memory.argtypes = [vp, C.POINTER(sz), C.POINTER(sz)]; L.zj_batch_memory.restype = C.c_int
    L.zj_batch_create_rows.argtypes = [vp, C.POINTER(zj_image), sz, C.c_uint32, u32p, u32p]; L.zj_batch_create_rows.restype = vp
    L.zj_batch_counters.argtypes = [vp, u32p]; L.zj_batch_counters.restype = C.c_int
    L.zj_decode_batch_noalloc.argtypes = [vp, C.POINTER(zj_image), sz, C.POINTER(vp), C.POINTER(sz), u32p]; L.zj_decode_batch_noalloc.restype = C.c_int
    L.zj_batch_decode_entropy.argtypes = [vp]; L.zj_batch_decode_entropy.restype = C.c_int
    L.zj_batch_decode_pixels.argtypes = [vp, C.POINTER(vp), C.POINTER(sz), vp]; L.zj_batch_decode_pixels.restype = C.c_int
    L.zj_batch_create_range.argtypes = [vp, C.POINTER(zj_image), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]; L.zj_batch_create_range.restype = vp
    L.zj_batch_range_front.argtypes = [vp, C.POINTER(zj_range_report)]; L.zj_batch_range_front.restype = C.c_int
    L.zj_batch_range_resolve.argtypes = [vp, C.POINTER(zj_range_report), C.c_uint32, u32p, u32p, u32p, u32p]; L.zj_batch_range_resolve.restype = C.c_int
    L.zj_batch_range_decode.argtypes = [vp, vp, sz]; L.zj_batch_range_decode.restype = C.c_int
    _lib = L
    return L


def exported_symbols():
    """Names declared in include/zj.h that the shared object must export."""
    return list(_EXPORTS)
