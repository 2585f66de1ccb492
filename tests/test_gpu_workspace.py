"""Batches that share device memory (zj_workspace_*, zj_batch_create_in): parity with the oracle when the
scratch range still holds what a larger batch left there, the estimate as an upper bound, clean refusal when
a batch does not fit, and the argument checks of the decode call."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle

import jpeg_decoder_amd as J
from jpeg_decoder_amd import binding as B, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = J.Context(0)
    yield c
    c.close()


def _check(files, imgs, outs, st, batch=None):
    for i, (f, im) in enumerate(zip(files, imgs)):
        ref = oracle.decode(f)
        assert st[i] == 0
        rgb = outs[i][:, : im.width * 3].reshape(im.height, im.width, 3).cpu().numpy()
        assert np.array_equal(rgb, ref.rgb), i
        if batch is not None:
            assert np.array_equal(batch.coefficients(i), ref.qcoef), i


def test_batches_share_scratch_and_alternate(ctx):
    """Two batches alive in two plan slots decode in turn, several times, on one scratch range: the small
    batch always finds the large one's leftovers under its buffers (restart segments, a flat image whose
    stream barely fills a window, white noise), and the other way round."""
    big = [synth.encode(1920, 1080, 2, 2, 90, 0, 201, 0), synth.encode(1280, 720, 1, 1, 92, 2, 202, 11),
           synth.encode(1000, 600, 2, 1, 90, 0, 203, 0), synth.encode(800, 608, 2, 2, 90, 1, 204, 25)]
    small = [synth.encode(640, 360, 2, 2, 90, 1, 205, 0), synth.encode(200, 120, 1, 1, 90, 0, 206, 2),
             synth.encode(77, 29, 0, 0, 90, 0, 207, 0), synth.encode(512, 512, 1, 2, 75, 2, 208, 9)]
    ib, is_ = [J.ParsedImage(f) for f in big], [J.ParsedImage(f) for f in small]
    pb, sb = J.Workspace.estimate(ib)
    ps, ss = J.Workspace.estimate(is_)
    ws = J.Workspace(ctx, max(sb, ss), max(pb, ps), 2)
    b0 = J.Batch(ctx, ib, workspace=ws, slot=0)
    b1 = J.Batch(ctx, is_, workspace=ws, slot=1)
    for b, est in ((b0, (pb, sb)), (b1, (ps, ss))):
        used = b.memory()
        assert 0 < used[0] <= est[0] and 0 < used[1] <= est[1], (used, est)
    o0, o1 = b0.alloc_rgb(), b1.alloc_rgb()
    for rep in range(3):
        for o in o0 + o1:
            o.fill_(0x33)
        b0.decode(o0)
        b1.decode(o1)
        st1 = b1.finish()
        _check(small, is_, o1, st1, b1)  # b1 decoded last: its coefficients are the ones in the arena
        st0 = b0.finish()
        _check(big, ib, o0, st0)
        # asking the batch that did not decode last for its coefficients is against the rules (they are gone) and must at
        # least leave alone what is in the shared scratch now: the other batch's
        b0.coefficients(0)
        assert np.array_equal(b1.coefficients(1), oracle.decode(small[1]).qcoef)
        b0.decode(o0)
        st0 = b0.finish()
        _check(big, ib, o0, st0, b0)
    assert b0.counters() == (0, 0, 0) and b1.counters() == (0, 0, 0)
    b0.close()
    # a slot is reused for a third batch while the other batch stays alive
    third = [synth.encode(333, 211, 2, 2, 90, 0, 209, 4), synth.encode(1920, 1080, 2, 2, 90, 0, 210, 0)]
    i3 = [J.ParsedImage(f) for f in third]
    b2 = J.Batch(ctx, i3, workspace=ws, slot=0)
    o2 = b2.alloc_rgb()
    b2.decode(o2)
    _check(third, i3, o2, b2.finish(), b2)
    b1.decode(o1)
    _check(small, is_, o1, b1.finish(), b1)
    b1.close(); b2.close(); ws.close()


def test_estimate_is_an_upper_bound(ctx):
    """The host-side estimate covers what the planner really takes, for streams from nearly empty to white
    noise, many short restart segments, every layout, forced subsequence sizes."""
    rng = np.random.default_rng(5)
    cases = []
    for k in range(14):
        w, h = int(rng.integers(1, 900)), int(rng.integers(1, 700))
        hs, vs = [(2, 2), (1, 1), (2, 1), (1, 2), (0, 0)][k % 5]
        ri = int(rng.choice([0, 0, 1, 2, 7, 40]))
        cases.append(synth.encode(w, h, hs, vs, int(rng.choice([50, 90, 100])), int(rng.integers(0, 3)), 300 + k, ri))
    imgs = [J.ParsedImage(f) for f in cases]
    for group in (imgs, imgs[:1], imgs[3:9], []):
        plan, scratch = J.Workspace.estimate(group)
        ws = J.Workspace(ctx, scratch, plan, 1)
        for sb in (0, 512, 4096):
            b = J.Batch(ctx, group, sb, workspace=ws, slot=0)
            used = b.memory()
            assert used[0] <= plan and used[1] <= scratch
            outs = b.alloc_rgb()
            b.decode(outs)
            assert not b.finish().any()
            b.close()
        ws.close()
    _check(cases[:4], imgs[:4], *_decode_in_fresh_workspace(ctx, imgs[:4]))


def _decode_in_fresh_workspace(ctx, imgs):
    plan, scratch = J.Workspace.estimate(imgs)
    ws = J.Workspace(ctx, scratch, plan, 1)
    b = J.Batch(ctx, imgs, workspace=ws, slot=0)
    outs = b.alloc_rgb()
    b.decode(outs)
    st = b.finish()
    b.close(); ws.close()
    return outs, st


def test_workspace_too_small_is_refused(ctx):
    imgs = [J.ParsedImage(synth.encode(1920, 1080, 2, 2, 90, 0, 220, 0))]
    plan, scratch = J.Workspace.estimate(imgs)
    ws = J.Workspace(ctx, scratch // 4, plan, 1)
    with pytest.raises(J.DecodeError, match="scratch"):
        J.Batch(ctx, imgs, workspace=ws, slot=0)
    ws.close()
    ws = J.Workspace(ctx, scratch, plan // 8, 1)
    with pytest.raises(J.DecodeError, match="plan"):
        J.Batch(ctx, imgs, workspace=ws, slot=0)
    with pytest.raises(J.DecodeError):
        J.Batch(ctx, imgs, workspace=ws, slot=1)  # no such slot
    ws.close()


def test_two_decodes_in_flight_with_different_outputs(ctx):
    """zj_batch_decode is asynchronous: a second call with other output buffers, issued before the first
    has run, must not redirect the first call's pixels."""
    import torch
    files = [synth.encode(1920, 1080, 2, 2, 90, 0, 230 + k, 0) for k in range(6)]
    imgs = [J.ParsedImage(f) for f in files]
    b = J.Batch(ctx, imgs)
    a, c = b.alloc_rgb(), b.alloc_rgb()
    for o in a + c:
        o.fill_(0x77)
    for _ in range(4):  # keeps the stream busy so that the later calls are queued behind running kernels
        b.decode(a)
    b.decode(c)
    b.decode(a)
    st = b.finish()
    _check(files, imgs, a, st)
    _check(files, imgs, c, st)
    b.close()


def test_two_decodes_in_flight_on_a_workspace_batch(ctx):
    """The same for a batch that lives in a workspace: its pointer tables are staged in pinned memory of the plan slot and
    rewritten only behind the event of the copies that read them."""
    files = [synth.encode(1920, 1080, 2, 2, 90, 0, 250 + k, 0) for k in range(4)]
    imgs = [J.ParsedImage(f) for f in files]
    plan, scratch = J.Workspace.estimate(imgs)
    ws = J.Workspace(ctx, scratch, plan, 1)
    b = J.Batch(ctx, imgs, workspace=ws, slot=0)
    a, c = b.alloc_rgb(), b.alloc_rgb()
    for o in a + c:
        o.fill_(0x77)
    for _ in range(4):
        b.decode(a)
    b.decode(c)
    b.decode(a)
    st = b.finish()
    _check(files, imgs, a, st)
    _check(files, imgs, c, st)
    b.close()
    ws.close()


def test_plan_slots_and_workspace_lifetime(ctx):
    """A plan slot holds one live batch: a second zj_batch_create_in on it is refused until the first is destroyed (it
    would overwrite scans and tables a queued decode still reads).  A workspace destroyed while batches are alive is
    freed by the last of them, so closing in the wrong order is safe."""
    files = [synth.encode(320, 240, 2, 2, 90, 0, 260 + k, 0) for k in range(2)]
    imgs = [J.ParsedImage(f) for f in files]
    plan, scratch = J.Workspace.estimate(imgs)
    ws = J.Workspace(ctx, scratch, plan, 2)
    b0 = J.Batch(ctx, imgs, workspace=ws, slot=0)
    with pytest.raises(J.DecodeError, match="still held"):
        J.Batch(ctx, imgs, workspace=ws, slot=0)
    b1 = J.Batch(ctx, imgs[:1], workspace=ws, slot=1)
    b0.close()
    b0 = J.Batch(ctx, imgs, workspace=ws, slot=0)  # the slot is free again
    outs = b0.alloc_rgb()
    b0.decode(outs)
    ws.close()  # before its batches: deferred
    with pytest.raises(J.DecodeError):
        J.Batch(ctx, imgs, workspace=ws, slot=0)  # a closed workspace takes no new batches
    st = b0.finish()
    _check(files, imgs, outs, st)
    b1.close()
    b0.close()  # the last batch frees the workspace


def test_decode_rejects_bad_output_geometry(ctx):
    import torch
    im = J.ParsedImage(synth.encode(100, 40, 2, 2, 90, 0, 240, 0))
    b = J.Batch(ctx, [im])
    buf = torch.zeros(100 * 40 * 3, dtype=torch.uint8, device="cuda:0")
    ptrs = (C.c_void_p * 1)(buf.data_ptr())
    strides = (C.c_size_t * 1)(299)  # one byte short of 3 * width
    assert B.lib().zj_batch_decode(b.h, ptrs, strides) == B.ZJ_E_INVALID
    for bad in (torch.zeros((40, 299), dtype=torch.uint8, device="cuda:0"), torch.zeros((39, 304), dtype=torch.uint8, device="cuda:0"),
                torch.zeros((40, 304), dtype=torch.int8, device="cuda:0"), torch.zeros((40, 304), dtype=torch.uint8),
                torch.zeros((40, 608), dtype=torch.uint8, device="cuda:0")[:, ::2]):
        with pytest.raises(J.DecodeError):
            b.decode([bad])
    ok = torch.zeros((40, 304), dtype=torch.uint8, device="cuda:0")
    b.decode([ok])
    assert b.finish()[0] == 0
    b.close()


def test_pixel_pass_on_a_stream_of_its_own(ctx):
    """zj_batch_decode_entropy + zj_batch_decode_pixels on a second stream give what zj_batch_decode gives, for
    two batches whose halves interleave (entropy 0, pixels 0, entropy 1, pixels 1), repeated so that an entropy
    pass follows the same batch's pixel pass on the other stream; also with the scratch in a workspace, and
    with output tables prepared once."""
    import torch
    fa = [synth.encode(1920, 1080, 2, 2, 90, 0, 301, 0), synth.encode(640, 480, 1, 1, 90, 2, 302, 7)]
    fb = [synth.encode(1280, 720, 2, 1, 85, 0, 303, 0), synth.encode(333, 211, 2, 2, 90, 1, 304, 3), synth.encode(64, 64, 0, 0, 90, 0, 305, 0)]
    ia, ib = [J.ParsedImage(f) for f in fa], [J.ParsedImage(f) for f in fb]
    pix = torch.cuda.Stream()
    ests = [J.Workspace.estimate(ia), J.Workspace.estimate(ib)]
    ws = J.Workspace(ctx, max(e[1] for e in ests), max(e[0] for e in ests), 2)
    for shared in (False, True):
        ba = J.Batch(ctx, ia, workspace=ws, slot=0) if shared else J.Batch(ctx, ia)
        bb = J.Batch(ctx, ib, workspace=ws, slot=1) if shared else J.Batch(ctx, ib)
        oa, ob = ba.alloc_rgb(), bb.alloc_rgb()
        ta, tb = ba.output_tables(oa), bb.output_tables(ob)
        with pytest.raises(J.DecodeError):
            ba.decode_pixels(ta, pix.cuda_stream)  # nothing to convert yet
        for rep in range(3):
            for o in oa + ob:
                o.fill_(0x5A)
            torch.cuda.synchronize()
            ba.decode_entropy()
            ba.decode_pixels(ta, pix.cuda_stream)
            bb.decode_entropy()
            bb.decode_pixels(tb if rep else ob, pix.cuda_stream)
            _check(fb, ib, ob, bb.finish())  # finish waits for the pixel stream as well
            _check(fa, ia, oa, ba.finish())
        # and the two halves on the context's own stream
        for o in oa:
            o.fill_(0)
        ba.decode_entropy()
        ba.decode_pixels(ta)
        _check(fa, ia, oa, ba.finish(), None if shared else ba)
        ba.close(); bb.close()
    with pytest.raises(ValueError):
        J.Batch(ctx, ia).decode(tb)  # tables of another batch
