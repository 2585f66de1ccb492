"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same
inputs.  Bars: quantised coefficients bit-exact; dequantised natural-order blocks bit-exact; pixels
bit-exact (the required bar is <= 1 LSB; the kernels keep the reference's f32 operation order, so
equality is asserted and the maximum difference is reported on failure)."""
import ctypes as C
import json
import os
import zlib

import numpy as np
import pytest

from conftest import FIXTURES, GOLDEN, fixture_bytes
from oracle import oracle

import jpeg_decoder_amd as J
from jpeg_decoder_amd import binding as B, synth

pytestmark = pytest.mark.gpu
G = json.load(open(os.path.join(GOLDEN, "golden.json")))


@pytest.fixture(scope="module")
def ctx():
    c = J.Context(0)
    yield c
    c.close()


def assert_same(data, rgb, status, qcoef, dq=None):
    ref = oracle.decode(data)
    assert status == 0, f"status {status}"
    assert np.array_equal(qcoef, ref.qcoef), "quantised coefficients differ"
    if dq is not None:
        assert np.array_equal(dq, ref.dqcoef), "dequantised blocks differ"
    if not np.array_equal(rgb, ref.rgb):
        d = np.abs(rgb.astype(int) - ref.rgb.astype(int))
        raise AssertionError(f"pixels differ: max {d.max()} LSB, {np.count_nonzero(d)} samples")


@pytest.mark.parametrize("name", [f for f in FIXTURES if f != "large.jpg"])
@pytest.mark.parametrize("subbits", [0, 512, 1024, 8192, 65536])
def test_fixture_parity(ctx, name, subbits):
    data = fixture_bytes(name)
    img = J.ParsedImage(data)
    b = J.Batch(ctx, [img], subbits)
    outs = b.alloc_rgb()
    b.decode(outs)
    st = b.finish()
    rgb = outs[0][:, : img.width * 3].reshape(img.height, img.width, 3).cpu().numpy()
    assert_same(data, rgb, st[0], b.coefficients(0), b.coefficients(0, dequantised=True))
    b.close()


def test_one_component_images(ctx):
    """Grayscale files (one block per MCU, R = G = B): alone at several sizes, with restart markers, and in one
    batch with colour images."""
    files = [synth.encode(w, h, 0, 0, q, cls, seed, ri) for (w, h, q, cls, seed, ri) in
             [(100, 60, 90, 0, 5, 0), (1, 1, 90, 0, 6, 0), (769, 9, 75, 2, 7, 0), (640, 480, 90, 0, 8, 5), (1920, 1080, 90, 0, 9, 0)]]
    for data in files:
        for sb in (0, 2048):
            rgb, st, coef = J.decode_images(ctx, [J.ParsedImage(data)], subseq_bits=sb, want_coef=True)
            assert_same(data, rgb[0], st[0], coef[0])
    mixed = [files[0], synth.encode(200, 120, 2, 2, seed=2), files[3], synth.encode(90, 150, 1, 1, seed=4), files[2]]
    rgb, st, coef = J.decode_images(ctx, [J.ParsedImage(f) for f in mixed], want_coef=True)
    for f, r, s_, c in zip(mixed, rgb, st, coef):
        assert_same(f, r, s_, c)


def test_canonical_search_path(ctx):
    """Tables built without their second level: every code longer than nine bits is found by the canonical
    search inside the kernels (the path that only unusual tables reach otherwise)."""
    for name in ("porsche.jpg", "out.jpg"):
        data = fixture_bytes(name)
        rgb, st, coef = J.decode_images(ctx, [J.ParsedImage(data)], subseq_bits=B.ZJ_FLAG_NO_SECOND_LEVEL | 4096, want_coef=True)
        assert_same(data, rgb[0], st[0], coef[0])


def test_large_fixture_parity_and_golden(ctx):
    """large.jpg (7680x4320, no restart markers): bit-exact vs the oracle and vs the committed hashes."""
    data = fixture_bytes("large.jpg")
    img = J.ParsedImage(data)
    rgb, st, coef = J.decode_images(ctx, [img], want_coef=True)
    e = G["fixtures"]["large.jpg"]
    assert st[0] == 0
    assert zlib.crc32(coef[0].tobytes()) == e["qcoef_crc"]
    assert zlib.crc32(np.ascontiguousarray(rgb[0]).tobytes()) == e["rgb_crc"]


@pytest.mark.parametrize("name", sorted(G["synthetic"]))
def test_synthetic_golden(ctx, name):
    """Committed golden hashes (sampling 4:4:4 / 4:2:2 / 4:4:0 / 4:2:0, restart intervals, odd sizes)."""
    e = G["synthetic"][name]
    data = synth.encode(*e["params"])
    rgb, st, coef = J.decode_images(ctx, [J.ParsedImage(data)], want_coef=True)
    assert st[0] == 0
    assert zlib.crc32(coef[0].tobytes()) == e["qcoef_crc"]
    assert zlib.crc32(np.ascontiguousarray(rgb[0]).tobytes()) == e["rgb_crc"]


def test_mixed_batch(ctx):
    """One launch over images of different size, sampling, tables and restart structure."""
    files = [fixture_bytes("porsche.jpg"), synth.encode(333, 211, 2, 2, seed=5, restart_interval=4),
             fixture_bytes("test-baseline.jpg"), synth.encode(200, 120, 1, 1, seed=2), fixture_bytes("out.jpg"),
             synth.encode(90, 150, 1, 2, 75, seed=4), synth.encode(150, 90, 2, 1, 75, seed=3),
             synth.encode(8, 8, 1, 1, seed=9), synth.encode(17, 9, 2, 2, seed=10, restart_interval=1),
             fixture_bytes("profile.jpg")]
    imgs = [J.ParsedImage(f) for f in files]
    b = J.Batch(ctx, imgs)
    outs = b.alloc_rgb()
    b.decode(outs)
    st = b.finish()
    for i, (f, im) in enumerate(zip(files, imgs)):
        rgb = outs[i][:, : im.width * 3].reshape(im.height, im.width, 3).cpu().numpy()
        assert_same(f, rgb, st[i], b.coefficients(i), b.coefficients(i, dequantised=True))
    b.close()


def test_edge_sizes(ctx):
    """Empty batch, 1x1 image, widths/heights that are not multiples of the MCU (crop), unaligned rows."""
    b = J.Batch(ctx, [])
    b.decode(None)
    assert len(b.finish()) == 0
    b.close()
    for (w, h, hs, vs) in [(1, 1, 1, 1), (1, 1, 2, 2), (15, 17, 2, 2), (33, 7, 1, 1), (129, 65, 2, 1), (255, 31, 1, 2), (1000, 3, 2, 2)]:
        data = synth.encode(w, h, hs, vs, seed=w * 7 + h)
        rgb, st, coef = J.decode_images(ctx, [J.ParsedImage(data)], want_coef=True)
        assert_same(data, rgb[0], st[0], coef[0])


def test_output_pitch_and_alignment(ctx):
    """Caller-chosen output buffers: a tight pitch of 3 * width bytes that is no multiple of 16, a buffer that
    starts at an odd address, and a wide pitch -- every sampling layout and one-component frames.  Bytes
    outside the image's rows must stay untouched."""
    import torch
    cases = [(101, 37, 2, 2), (67, 50, 1, 1), (130, 23, 2, 1), (45, 70, 1, 2), (77, 29, 0, 0), (1921, 40, 2, 2)]
    files = [synth.encode(w, h, hs, vs, 90, 0, 60 + i, 0) for i, (w, h, hs, vs) in enumerate(cases)]
    imgs = [J.ParsedImage(f) for f in files]
    refs = [oracle.decode(f).rgb for f in files]
    for mode in ("tight", "odd", "wide"):
        b = J.Batch(ctx, imgs)
        bufs, outs = [], []
        for im in imgs:
            row = im.width * 3
            pitch = row if mode == "tight" else (row + 5 if mode == "odd" else (row + 63) // 16 * 16 + 16)
            off = 3 if mode == "odd" else 0
            buf = torch.full((off + im.height * pitch + 64,), 0xA5, dtype=torch.uint8, device="cuda:0")
            bufs.append(buf)
            outs.append(torch.as_strided(buf, (im.height, pitch), (pitch, 1), off))
        b.decode(outs)
        st = b.finish()
        assert not st.any()
        for i, im in enumerate(imgs):
            row = im.width * 3
            host = bufs[i].cpu().numpy()
            pitch = outs[i].stride(0)
            off = outs[i].storage_offset()
            got = np.stack([host[off + y * pitch: off + y * pitch + row] for y in range(im.height)]).reshape(im.height, im.width, 3)
            assert np.array_equal(got, refs[i]), (mode, i)
            mask = np.ones(host.shape, bool)
            for y in range(im.height):
                mask[off + y * pitch: off + y * pitch + row] = False
            assert (host[mask] == 0xA5).all(), (mode, i)
        b.close()


def test_content_classes_1080p(ctx):
    """Full-size 1080p 4:2:0 and 4:4:4 images of all three content classes (smooth streams have many
    blocks per subsequence, noise streams few)."""
    for (hs, vs, cls, seed) in [(2, 2, 0, 1), (2, 2, 1, 2), (2, 2, 2, 3), (1, 1, 0, 4)]:
        data = synth.encode(1920, 1080, hs, vs, 90, cls, seed)
        for sb in (2048, 65536):
            rgb, st, coef = J.decode_images(ctx, [J.ParsedImage(data)], subseq_bits=sb, want_coef=True)
            assert_same(data, rgb[0], st[0], coef[0])


def test_long_blocks_and_many_segments(ctx):
    """Blocks of many hundred bits (white noise at quality 100: the write pass has to restage a block's
    bits), and an image cut into more than a thousand restart segments, each decoded at the default
    subsequence size and with long subsequences."""
    cases = [synth.encode(512, 512, 2, 2, 100, 2, 31), synth.encode(256, 256, 1, 1, 100, 2, 32),
             synth.encode(1024, 1024, 2, 2, 90, 0, 33, 4), synth.encode(640, 480, 2, 1, 95, 2, 34, 3)]
    for data in cases:
        for sb in (0, 32768):
            rgb, st, coef = J.decode_images(ctx, [J.ParsedImage(data)], subseq_bits=sb, want_coef=True)
            assert_same(data, rgb[0], st[0], coef[0])


def test_short_segments_share_wavefronts(ctx):
    """Restart intervals far shorter than a wavefront's worth of subsequences are packed several to a
    wavefront, in lane groups of 1 to 32; forced subsequence sizes make the group size change from segment to
    segment inside one image, and a batch mixes such images with ordinary ones."""
    files = [synth.encode(640, 480, 2, 2, 90, 0, 51, 5), synth.encode(320, 240, 1, 1, 95, 2, 52, 3), synth.encode(512, 512, 2, 1, 75, 1, 53, 16),
             synth.encode(256, 256, 2, 2, 90, 0, 54, 0), synth.encode(1024, 64, 0, 0, 90, 2, 55, 7), synth.encode(640, 480, 2, 2, 98, 2, 56, 40)]
    imgs = [J.ParsedImage(f) for f in files]
    refs = [oracle.decode(f) for f in files]
    for sb in (0, 512, 1024, 2048, 4096, 16384):
        rgb, st, coef = J.decode_images(ctx, imgs, subseq_bits=sb, want_coef=True)
        assert not st.any(), (sb, st)
        for i, r in enumerate(refs):
            assert np.array_equal(coef[i], r.qcoef), (sb, i)
            assert np.array_equal(rgb[i], r.rgb), (sb, i)


def test_batch_of_1080p_is_deterministic_and_uniform(ctx):
    """Size-independent property at benchmark scale: 64 copies of 4 distinct 1080p images decode to
    identical coefficients and pixels per source image, run after run (no order-dependent results)."""
    import torch
    files = [synth.encode(1920, 1080, 2, 2, seed=s) for s in range(4)]
    refs = [oracle.decode(f) for f in files]
    imgs = [J.ParsedImage(files[i % 4]) for i in range(64)]
    b = J.Batch(ctx, imgs, 2048)
    outs = b.alloc_rgb()
    for rep in range(2):
        for o in outs:
            o.zero_()
        b.decode(outs)
        st = b.finish()
        assert not st.any()
        for i in range(64):
            r = refs[i % 4]
            rgb = outs[i][:, : 1920 * 3].reshape(1080, 1920, 3)
            assert torch.equal(rgb.cpu(), torch.from_numpy(r.rgb)), f"image {i} rep {rep}"
        for i in (0, 5, 63):
            assert np.array_equal(b.coefficients(i), refs[i % 4].qcoef)
    b.close()


def test_large_batches_use_long_write_runs(ctx):
    """From a few hundred thousand MCUs on, the planner gives every wavefront of the write pass longer runs of
    MCUs (up to eight times the base).  800 images of three layouts, every output compared."""
    import torch
    files = [synth.encode(1920, 1080, 2, 2, seed=71), synth.encode(1920, 1080, 1, 1, seed=72), synth.encode(1000, 600, 2, 1, seed=73),
             synth.encode(1920, 1080, 2, 2, seed=74, restart_interval=120)]
    refs = [oracle.decode(f) for f in files]
    pick = [0, 0, 0, 1, 0, 2, 0, 3]
    imgs = [J.ParsedImage(files[pick[i % 8]]) for i in range(800)]
    b = J.Batch(ctx, imgs)
    outs = b.alloc_rgb()
    b.decode(outs)
    st = b.finish()
    assert not st.any()
    want = [torch.from_numpy(r.rgb).to("cuda:0") for r in refs]
    for i, im in enumerate(imgs):
        k = pick[i % 8]
        assert torch.equal(outs[i][:, : im.width * 3].reshape(im.height, im.width, 3), want[k]), i
    for i in (0, 3, 5, 7, 799):
        assert np.array_equal(b.coefficients(i), refs[pick[i % 8]].qcoef)
    b.close()


def test_many_long_segments_scan_their_dc_values_by_segment(ctx):
    """A batch of hundreds of images without restart markers takes the DC scan that gives a whole segment to one
    workgroup: every layout (6, 3, 4 and 1 blocks per MCU), segments of one step (up to 4096 MCUs) and of several."""
    import torch
    files = [synth.encode(1920, 1080, 2, 2, seed=81), synth.encode(640, 480, 1, 1, seed=82), synth.encode(1024, 512, 2, 1, seed=83),
             synth.encode(512, 1024, 1, 2, seed=84), synth.encode(520, 520, 0, 0, seed=85), synth.encode(1280, 720, 1, 1, 80, seed=86)]
    refs = [oracle.decode(f) for f in files]
    imgs = [J.ParsedImage(files[i % 6]) for i in range(300)]
    b = J.Batch(ctx, imgs)
    outs = b.alloc_rgb()
    b.decode(outs)
    st = b.finish()
    assert not st.any()
    want = [torch.from_numpy(r.rgb).to("cuda:0") for r in refs]
    for i, im in enumerate(imgs):
        assert torch.equal(outs[i][:, : im.width * 3].reshape(im.height, im.width, 3), want[i % 6]), i
    for i in (0, 1, 2, 3, 4, 5, 299):
        assert np.array_equal(b.coefficients(i), refs[i % 6].qcoef)
    b.close()


def test_images_of_many_chunks_in_a_batch_of_many_segments(ctx):
    """The window scan's form for batches of 256 segments and more, on segments of ten chunks of 64 long subsequences
    (a 4K 4:4:4 image at 64 Kbit per subsequence) beside segments of one chunk and segments that share a wavefront."""
    import torch
    files = [synth.encode(3840, 2160, 1, 1, 90, 0, 61, 0), synth.encode(1920, 1080, 2, 2, 90, 0, 62, 0), synth.encode(64, 48, 2, 2, 90, 0, 63, 0)]
    refs = [oracle.decode(f) for f in files]
    pick = [0, 2, 2, 1] + [2] * 28
    imgs = [J.ParsedImage(files[pick[i % 32]]) for i in range(288)]
    b = J.Batch(ctx, imgs, 65536)
    outs = b.alloc_rgb()
    b.decode(outs)
    st = b.finish()
    assert not st.any()
    assert b.counters() == (0, 0, 0)
    want = [torch.from_numpy(r.rgb).to("cuda:0") for r in refs]
    for i, im in enumerate(imgs):
        assert torch.equal(outs[i][:, : im.width * 3].reshape(im.height, im.width, 3), want[pick[i % 32]]), i
    for i in (0, 1, 3, 256, 287):
        assert np.array_equal(b.coefficients(i), refs[pick[i % 32]].qcoef)
    b.close()


def test_restart_band_equals_rows_of_full_decode(ctx):
    """The multi-GPU split of one restart-marked image: every band decoded on its own equals the same
    rows of the whole image."""
    data = synth.encode(512, 256, 2, 2, seed=21, restart_interval=16)  # 32 MCUs per row
    p = J.ParsedImage(data)
    full = oracle.decode(data).rgb
    bands = [p.band(0, 5), p.band(5, 3), p.band(8, 8)]
    rgb, st = J.decode_images(ctx, bands)
    assert not st.any()
    got = np.concatenate(rgb, axis=0)
    assert np.array_equal(got, full)


def test_device_marker_search_equals_host_loop(ctx):
    """The planner finds the restart markers on the device; the segments it cuts must be those of the host
    loop (zj_restart_offsets) -- on clean files and on scans with markers planted at tile and thread
    boundaries of the search, behind fill bytes (FF FF Dn), at the very end, too many and too few."""
    rng = np.random.default_rng(7)
    base = synth.encode(768, 512, 2, 2, 92, 2, 41, 4)
    info = oracle.probe(base)
    assert info.scan_len > 5 * 16384
    files = [base, synth.encode(200, 120, 1, 1, 90, 0, 42, 1), synth.encode(320, 240, 2, 1, 90, 0, 43, 7), synth.encode(64, 64, 2, 2, 90, 0, 44, 0)]

    def plant(d, at, fill=False):
        at += info.scan_off
        if fill:
            d[at - 1] = 0xFF
        d[at] = 0xFF
        d[at + 1] = 0xD0 + int(rng.integers(0, 8))

    d = bytearray(base)
    for at in (16383, 16384, 16382, 2 * 16384 - 1, 63, 64, 127, 4095, 4096, info.scan_len - 2, info.scan_len - 3):
        plant(d, at)
    plant(d, 3 * 16384 + 1, fill=True)
    plant(d, 70000, fill=True)
    files.append(bytes(d))
    d = bytearray(base)  # far more markers than intervals
    for at in rng.integers(2, info.scan_len - 4, 3000):
        plant(d, int(at))
    files.append(bytes(d))
    d = bytearray(base)  # markers removed: fewer than intervals
    for k in range(info.scan_off, info.scan_off + info.scan_len - 1):
        if d[k] == 0xFF and 0xD0 <= d[k + 1] <= 0xD7 and rng.integers(0, 3):
            d[k + 1] = 0x00
    files.append(bytes(d))
    files.append(base[: info.scan_off + 40000] + b"\xff")  # ends in a lone FF
    imgs = [J.ParsedImage(f) for f in files]
    seg = {}
    for flag in (0, B.ZJ_FLAG_HOST_MARKER_SCAN):
        b = J.Batch(ctx, imgs, flag)
        outs = b.alloc_rgb()
        b.decode(outs)
        st = b.finish()
        seg[flag] = ([b.segments(i) for i in range(len(imgs))], st.copy(), [o[:, : im.width * 3].cpu().numpy() for o, im in zip(outs, imgs)])
        b.close()
    dev, host = seg[0], seg[B.ZJ_FLAG_HOST_MARKER_SCAN]
    for i in range(len(imgs)):
        assert dev[0][i] == host[0][i], f"segments of image {i}"
        if dev[1][i] == 0:  # blocks that a damaged stream does not reach are not written
            assert np.array_equal(dev[2][i], host[2][i])
    assert np.array_equal(dev[1], host[1])
    assert not dev[1][:4].any()
    p0 = imgs[0]
    offs = p0.restart_offsets()
    assert [s[0] for s in dev[0][0]] == offs[: len(dev[0][0])]


ORACLE_CLASS = {oracle.E_HUFF: B.ZJ_ST_BAD_CODE, oracle.E_COEF: B.ZJ_ST_COEF_OVERFLOW, oracle.E_TRUNC: B.ZJ_ST_TRUNCATED}


def marker_pairs(file):
    """Offsets (inside the scan) of every FF that is followed by something other than 00."""
    i = oracle.probe(file)
    scan = np.frombuffer(file, np.uint8)[i.scan_off: i.scan_off + i.scan_len]
    ff = np.nonzero((scan[:-1] == 0xFF) & (scan[1:] != 0x00))[0]
    return set(int(k) for k in ff), int(i.scan_len)


def damage_touched_a_marker(damaged, base):
    """Did the damage plant a marker inside the entropy-coded data, or destroy one of the base file's restart markers?
    The reference's reader drops the two bytes of any marker it meets and lets the look-ahead in progress see zero bits
    instead (src/bitstream.rs:41-47, :79-84), so what it decodes from there on depends on where its refill loop happened
    to stand; the GPU path (and T.81) know markers in the entropy-coded data only as the RSTn of a restart interval and
    treat anything else as data bytes.  That deviation is deliberate; such streams are decoded here (no fault, no hang,
    the same status at every subsequence size) but not compared with the oracle."""
    got, n = marker_pairs(damaged)
    want, _ = marker_pairs(base)
    return got != set(k for k in want if k + 1 < n)


def assert_status_is_the_oracles(files, imgs, st, coef, rgb=None):
    """Where the reference panics inside a block (src/decoder.rs:194-228, src/ec.rs:27,47) the oracle stops with an
    error class and the index of that block.  The GPU decodes every block in parallel, so its status word holds the
    bits of all bad blocks: the oracle's class must be among them, every block before the oracle's first bad one must
    equal the oracle's, and a stream the oracle decodes cleanly must come out clean and identical."""
    classes = {}
    for k, (f, im) in enumerate(zip(files, imgs)):
        o = oracle.decode(f)
        classes[o.status] = classes.get(o.status, 0) + 1
        if o.status == oracle.OK:
            assert st[k] == 0, (k, st[k])
            assert np.array_equal(coef[k], o.qcoef), k
            if rgb is not None:
                assert np.array_equal(rgb[k], o.rgb), k
            continue
        assert o.status in ORACLE_CLASS, (k, o.status)
        assert st[k] & ORACLE_CLASS[o.status], (k, o.status, st[k])
        n = int(o.info.err_block)
        assert 0 <= n < im.nblocks
        assert np.array_equal(coef[k][:n], o.qcoef), (k, n)
    return classes


def test_corrupt_streams_set_status_not_faults(ctx):
    """Malformed entropy data: the reference panics (src/decoder.rs:194-228, src/ec.rs:27,47); the GPU
    path must return status bits and never fault, and the bits are the oracle's error class."""
    good = synth.encode(128, 128, 2, 2, seed=3)
    i = oracle.probe(good)
    cut = good[: i.scan_off + i.scan_len // 3] + b"\xff\xd9"
    rng = np.random.default_rng(0)
    noisy = bytearray(good)
    for k in rng.integers(i.scan_off + 16, i.scan_off + i.scan_len - 16, 40):
        if noisy[k] != 0xFF and noisy[k - 1] != 0xFF:
            noisy[k] = (noisy[k] ^ 0x5A) & 0xFE
    zeros = good[: i.scan_off] + bytes(i.scan_len) + b"\xff\xd9"
    files = [cut, bytes(noisy), zeros, good]
    imgs = [J.ParsedImage(f) for f in files]
    rgb, st, coef = J.decode_images(ctx, imgs, want_coef=True)
    assert st[0] & B.ZJ_ST_TRUNCATED
    assert st[3] == 0
    assert_status_is_the_oracles(files, imgs, st, coef, rgb)
    # a scan of zero bytes is a run of two-bit codes: every block ends at once and the stream runs out long after
    # the last MCU, so both decoders call it clean; noise cannot be clean
    assert oracle.decode(bytes(noisy)).status != oracle.OK and st[1] != 0


def test_fuzzed_streams_give_the_same_status_at_every_subsequence_size(ctx):
    """Random damage to the entropy data (bit flips, random bytes, stray FF bytes, truncation) of images with
    and without restart markers: no fault, no hang, no device error, the status words do not depend on how
    the stream was cut into subsequences, and they carry the oracle's error class."""
    rng = np.random.default_rng(123)
    files = []
    for b in (synth.encode(256, 256, 2, 2, 90, 0, 1, 0), synth.encode(200, 120, 1, 1, 90, 2, 2, 8)):
        info = oracle.probe(b)
        for _ in range(30):
            d = bytearray(b)
            for k in rng.integers(info.scan_off, info.scan_off + info.scan_len - 2, int(rng.integers(1, 30))):
                mode = rng.integers(0, 3)
                d[k] = (d[k] ^ (1 << int(rng.integers(0, 8)))) if mode == 0 else (int(rng.integers(0, 256)) if mode == 1 else 0xFF)
            if rng.integers(0, 4) == 0:
                d = d[: int(rng.integers(info.scan_off + 4, info.scan_off + info.scan_len))] + b"\xff\xd9"
            files.append((bytes(d), b))
    imgs, kept, plain = [], [], []
    for f, b in files:
        try:
            imgs.append(J.ParsedImage(f))
            kept.append(f)
            plain.append(not damage_touched_a_marker(f, b))
        except J.DecodeError:
            pass
    assert len(imgs) > 40 and sum(plain) > 25
    res = [J.decode_images(ctx, imgs, subseq_bits=sb, want_coef=True) for sb in (0, 512, 16384)]
    sts = [r[1] for r in res]
    assert np.array_equal(sts[0], sts[1]) and np.array_equal(sts[0], sts[2])
    assert (sts[0] != 0).sum() > len(imgs) // 2
    sel = [k for k in range(len(imgs)) if plain[k]]
    for r in res:
        classes = assert_status_is_the_oracles([kept[k] for k in sel], [imgs[k] for k in sel], r[1][sel], [r[2][k] for k in sel])
    assert len([c for c in classes if c != oracle.OK]) >= 2  # the damage produced more than one class of error


def test_one_shot_entry_point_host_outputs(ctx):
    """zj_decode_batch with host output pointers (the call shape of the reference: file in, pixels out)."""
    data = fixture_bytes("porsche.jpg")
    p = J.ParsedImage(data)
    rgb = np.zeros((p.height, p.width, 3), np.uint8)
    q = np.zeros((p.nblocks, 64), np.int16)
    dq = np.zeros((p.nblocks, 64), np.int16)
    out = B.zj_output(rgb.ctypes.data, 0, q.ctypes.data, dq.ctypes.data, B.ZJ_MEM_HOST)
    st = C.c_uint32(99)
    img = (B.zj_image * 1)(p.image)
    rc = B.lib().zj_decode_batch(ctx.h, img, 1, C.byref(out), C.byref(st))
    assert rc == 0
    assert_same(data, rgb, st.value, q, dq)


def test_decoder_api_writes_ppm(tmp_path, ctx):
    """Decoder(file).decode() mirrors Decoder::new / decode (src/decoder.rs:323-330): out.ppm holds the
    header of :580 and the payload of :648."""
    path = os.path.join(GOLDEN, "profile.jpg")
    out = str(tmp_path / "out.ppm")
    with open(path, "rb") as f:
        J.Decoder(f).decode(out)
    blob = open(out, "rb").read()
    ref = oracle.decode(fixture_bytes("profile.jpg"))
    head = b"P6\n400 400\n255\n"
    assert blob.startswith(head) and blob[len(head):] == ref.rgb.tobytes()
