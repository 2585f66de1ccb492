"""The entry point that allocates nothing (zj_decode_batch_noalloc), called through the C ABI and checked
against zj_decode_batch with device outputs: two calls in a row on the same GPU, the second with a smaller
input, so that whatever the first call left in the fixed scratch array lies under and behind the second
call's buffers.  Pixels must be identical byte for byte (both run the same kernels; the bar is the one of
the other parity tests), status words equal, and bytes outside the image rows untouched."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle

import jpeg_decoder_amd as J
from jpeg_decoder_amd import binding as B, synth

pytestmark = pytest.mark.gpu


def _existing_entry_point(ctx, imgs, shapes):
    """zj_decode_batch with ZJ_MEM_DEVICE outputs."""
    import torch
    n = len(imgs)
    outs = [torch.full(s, 0x5A, dtype=torch.uint8, device="cuda:0") for s in shapes]
    arr = J.decoder.image_array(imgs)
    zo = (B.zj_output * n)()
    for i, t in enumerate(outs):
        zo[i] = B.zj_output(t.data_ptr(), t.stride(0), None, None, B.ZJ_MEM_DEVICE)
    st = (C.c_uint32 * n)()
    rc = B.lib().zj_decode_batch(ctx.h, arr, n, zo, st)
    assert rc == 0, ctx.last_error()
    torch.cuda.synchronize()
    return outs, np.array(st[:], np.uint32)


def _noalloc_entry_point(ctx, imgs, shapes):
    import torch
    n = len(imgs)
    outs = [torch.full(s, 0x5A, dtype=torch.uint8, device="cuda:0") for s in shapes]
    arr = J.decoder.image_array(imgs)
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in outs])
    strides = (C.c_size_t * n)(*[t.stride(0) for t in outs])
    st = (C.c_uint32 * n)()
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info(0)[0]
    rc = B.lib().zj_decode_batch_noalloc(ctx.h, arr, n, ptrs, strides, st)
    assert rc == 0, ctx.last_error()
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info(0)[0] == free_before  # the call took no device memory (and gave none back)
    return outs, np.array(st[:], np.uint32)


def test_no_alloc_entry_point_equals_decode_batch_twice_second_smaller():
    import torch
    ctx = J.Context(0)
    try:
        # first call: three images of three layouts, one with restart markers, one damaged (status bits must agree too)
        good = synth.encode(1280, 720, 2, 2, 90, 0, 101, 0)
        info = oracle.probe(good)
        cut = good[: info.scan_off + info.scan_len // 2] + b"\xff\xd9"
        first = [good, synth.encode(640, 480, 1, 1, 90, 2, 102, 7), synth.encode(333, 211, 2, 1, 85, 0, 103, 0), cut]
        # second call: smaller input (fewer, smaller images, shorter streams)
        second = [synth.encode(320, 240, 2, 2, 90, 0, 104, 3), synth.encode(97, 61, 1, 1, 90, 0, 105, 0)]
        for files in (first, second):
            imgs = [J.ParsedImage(f) for f in files]
            shapes = [(im.height + 2, (im.width * 3 + 15) // 16 * 16 + 16) for im in imgs]  # two spare rows, spare columns
            want, st_want = _existing_entry_point(ctx, imgs, shapes)
            got, st_got = _noalloc_entry_point(ctx, imgs, shapes)
            assert np.array_equal(st_got, st_want), (st_got, st_want)
            for i, (g, w) in enumerate(zip(got, want)):
                assert torch.equal(g, w), f"image {i} differs from zj_decode_batch"
            for f, im, g, s in zip(files, imgs, got, st_got):
                if s == 0:  # and against the oracle, like every other parity test
                    ref = oracle.decode(f).rgb
                    assert np.array_equal(g[: im.height, : im.width * 3].reshape(im.height, im.width, 3).cpu().numpy(), ref)
    finally:
        ctx.close()


def test_no_alloc_entry_point_rejects_what_does_not_fit():
    import torch
    ctx = J.Context(0)
    try:
        big = J.ParsedImage(synth.encode(4096, 4096, 1, 1, 90, 0, 106, 0))  # 4:4:4, 100 MB of coefficients
        out = torch.empty((4096, 4096 * 3), dtype=torch.uint8, device="cuda:0")
        arr = J.decoder.image_array([big])
        ptrs = (C.c_void_p * 1)(out.data_ptr())
        strides = (C.c_size_t * 1)(out.stride(0))
        st = (C.c_uint32 * 1)()
        rc = B.lib().zj_decode_batch_noalloc(ctx.h, arr, 1, ptrs, strides, st)
        assert rc == B.ZJ_E_NOMEM
        assert str(B.ZJ_NOALLOC_MAX_BYTES) in ctx.last_error()
        # and the entry point still works afterwards
        small = J.ParsedImage(synth.encode(64, 64, 2, 2, 90, 0, 107, 0))
        o2 = torch.empty((64, 64 * 3), dtype=torch.uint8, device="cuda:0")
        st2 = J.decode_noalloc(ctx, [small], [o2])
        assert st2[0] == 0
        assert np.array_equal(o2.reshape(64, 64, 3).cpu().numpy(), oracle.decode(synth.encode(64, 64, 2, 2, 90, 0, 107, 0)).rgb)
    finally:
        ctx.close()
