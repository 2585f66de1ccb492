"""GPU tests of the configurations the headline metric names beyond the 1080p batch, and of the pieces around
the library: table selectors other than the reference's hard-wired ones, the mixed 1080p / 4K job decoded in
sub-batches of one workspace, the 16384x16384 restart-marked image whole and in eight bands, and the
command-line driver."""
import os
import subprocess
import zlib

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, band_file, fixture_bytes
from oracle import oracle

import jpeg_decoder_amd as J
from jpeg_decoder_amd import shard, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = J.Context(0)
    yield c
    c.close()


def test_table_selectors_other_than_the_hard_wired_ones(ctx):
    """E4: luma tables stored under id 1 (DC, AC and quantisation tables in every combination), chroma under id 0,
    headers naming them accordingly; src/decoder.rs:171-173 hard-wires Y -> 0, C -> 1 and :368-370 skips the scan
    header.  One batch holds all combinations, so table sets with different numbering sit side by side."""
    files = []
    for (hs, vs, ri, w, h) in [(2, 2, 3, 333, 211), (1, 1, 0, 200, 120), (2, 1, 0, 150, 90), (0, 0, 2, 100, 60)]:
        files += [synth.encode(w, h, hs, vs, 90, opt << 8, 40 + opt, ri) for opt in range(8)]
    imgs = [J.ParsedImage(f) for f in files]
    assert any(im.image.td[0] == 1 and im.image.ta[0] == 0 for im in imgs) and any(im.image.tq[0] == 1 for im in imgs)
    for sb in (0, 1024):
        rgb, st, coef = J.decode_images(ctx, imgs, subseq_bits=sb, want_coef=True)
        for f, r, s_, c in zip(files, rgb, st, coef):
            ref = oracle.decode(f)
            assert s_ == 0 and np.array_equal(c, ref.qcoef) and np.array_equal(r, ref.rgb)


def test_mixed_1080p_and_4k_job_in_sub_batches(ctx):
    """Config 4 in small: 1920x1080 and 3840x2160 images, 4:4:4 and 4:2:0, class by index bits as in the
    benchmark, partitioned over two ranks by scan bytes; each rank's share decoded in sub-batches that share one
    workspace; every image compared with the oracle, pixels and coefficients."""
    import torch
    classes = [(1920, 1080, 2, 2), (3840, 2160, 2, 2), (1920, 1080, 1, 1), (3840, 2160, 1, 1)]
    files = {c: synth.encode(*classes[c], 90, 0, 500 + c, 0) for c in range(4)}
    refs = {c: oracle.decode(f) for c, f in files.items()}
    parsed = {c: J.ParsedImage(f) for c, f in files.items()}
    job = [i & 3 for i in range(16)]
    parts = shard.partition_batch([len(files[c]) for c in job], 2)
    loads = [sum(len(files[job[i]]) for i in p) for p in parts]
    assert sorted(parts[0] + parts[1]) == list(range(16)) and abs(loads[0] - loads[1]) <= max(len(f) for f in files.values())
    for mine in parts:
        groups = [[job[i] for i in mine[g: g + 3]] for g in range(0, len(mine), 3)]
        est = [J.Workspace.estimate([parsed[c] for c in g]) for g in groups]
        ws = J.Workspace(ctx, max(e[1] for e in est), max(e[0] for e in est), len(groups))
        batches = [J.Batch(ctx, [parsed[c] for c in g], workspace=ws, slot=k) for k, g in enumerate(groups)]
        outs = [b.alloc_rgb() for b in batches]
        for rep in range(2):
            for b, o in zip(batches, outs):
                b.decode(o)
        for b, o, g in zip(batches, outs, groups):
            assert not b.finish().any()
            for t, c in zip(o, g):
                w, h = classes[c][0], classes[c][1]
                assert torch.equal(t[:, : w * 3].reshape(h, w, 3).cpu(), torch.from_numpy(refs[c].rgb)), c
        # coefficients: the arena holds the last decoded sub-batch
        batches[0].decode(outs[0])
        assert not batches[0].finish().any()
        for i, c in enumerate(groups[0]):
            assert np.array_equal(batches[0].coefficients(i), refs[c].qcoef)
        assert all(b.counters() == (0, 0, 0) for b in batches)
        for b in batches:
            b.close()
        ws.close()


def test_16384_square_restart_image_whole_and_in_eight_bands(ctx):
    """Config 5 at full size: one 16384x16384 4:2:0 image with a restart interval of 16 MCUs (65 536 intervals),
    decoded whole and as the eight bands eight ranks would take.  The image is sixteen repetitions of one
    1024-row strip (restart intervals are independent), so the oracle decodes the strip once and every
    1024 rows of both decodes must equal it; the bands must also equal the whole decode's CRC."""
    import torch
    size, strip_h, reps = 16384, 1024, 16
    data = synth.encode_tiled(size, strip_h, reps, 2, 2, 90, 0, 77, 16)
    ref = torch.from_numpy(oracle.decode(synth.encode(size, strip_h, 2, 2, 90, 0, 77, 16)).rgb).to("cuda:0")
    p = J.ParsedImage(data)
    assert (p.width, p.height, p.image.restart_interval) == (size, size, 16)
    whole = torch.empty((size, size * 3), dtype=torch.uint8, device="cuda:0")
    b = J.Batch(ctx, [p])
    b.decode([whole])
    assert not b.finish().any()
    assert b.counters() == (0, 0, 0)
    b.close()
    for k in range(reps):
        assert torch.equal(whole[k * strip_h: (k + 1) * strip_h].reshape(strip_h, size, 3), ref), f"strip {k} of the whole decode"
    mcux, mcuy, _, mh = shard.mcu_geometry(size, size, 2, 2)
    bands = shard.band_rows(mcux, mcuy, 16, 8)
    assert bands == [(128 * r, 128) for r in range(8)]
    gathered = torch.empty_like(whole)
    for (r0, n) in bands:  # one rank after the other on this GPU, each decoding straight into its rows
        bb = J.Batch(ctx, [p.band(r0, n)])
        bb.decode([gathered[r0 * mh: (r0 + n) * mh]])
        assert not bb.finish().any()
        bb.close()
    assert torch.equal(gathered, whole)
    # one band as a standalone file through the oracle (2 048 rows, a few seconds of CPU)
    alone = oracle.decode(band_file(data, p.band(3 * 128, 128)))
    assert alone.status == 0
    assert zlib.crc32(alone.rgb.tobytes()) == zlib.crc32(whole[3 * 2048: 4 * 2048].cpu().numpy().tobytes())


def test_command_line_driver_writes_the_reference_output(tmp_path):
    """The zj_decode binary (C++ zj::Decoder over the C ABI; counterpart of examples/decode.rs:1-9, which decodes
    test-images/profile.jpg to ./out.ppm): out.ppm must be the header of src/decoder.rs:580 and the oracle's pixels."""
    exe = os.path.join(ROOT, "jpeg-decoder_amd", "zj_decode")
    assert os.access(exe, os.X_OK)
    r = subprocess.run([exe, os.path.join(GOLDEN, "profile.jpg")], cwd=tmp_path, capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stderr
    blob = open(tmp_path / "out.ppm", "rb").read()
    ref = oracle.decode(fixture_bytes("profile.jpg"))
    head = b"P6\n400 400\n255\n"
    assert blob.startswith(head) and blob[len(head):] == ref.rgb.tobytes()
    # -o, a 4:2:0 file with a width that is no multiple of 8, and a file that is not a JPEG
    r = subprocess.run([exe, os.path.join(GOLDEN, "out.jpg"), "-o", str(tmp_path / "o2.ppm")], capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stderr
    blob = open(tmp_path / "o2.ppm", "rb").read()
    ref = oracle.decode(fixture_bytes("out.jpg"))
    head = b"P6\n1015 1000\n255\n"
    assert blob.startswith(head) and blob[len(head):] == ref.rgb.tobytes()
    bad = tmp_path / "bad.jpg"
    bad.write_bytes(b"not a jpeg")
    r = subprocess.run([exe, str(bad)], cwd=tmp_path, capture_output=True, text=True, timeout=100)
    assert r.returncode != 0 and "decode failed" in r.stderr


def test_rows_of_images_without_restart_markers(ctx):
    """An image without restart markers spread over several ranks: every rank repeats the entropy front over the
    whole stream and produces the coefficients and pixels of its band of MCU rows only (zj_batch_create_rows).  The
    bands of 1 to 5 ranks must tile the oracle's image, the coefficients inside a band must be the oracle's, and rows
    outside the image are refused.  large.jpg is the reference's restart-free 7680x4320 file; the others cover 4:2:0
    with a height that is no multiple of the MCU height, one-component frames, and a mixed batch with windows on
    some images only."""
    import torch
    cases = [("large.jpg", fixture_bytes("large.jpg")), ("out.jpg", fixture_bytes("out.jpg")),
             ("gray", synth.encode(333, 211, 0, 0, 90, 0, 81, 0)), ("422", synth.encode(640, 200, 2, 1, 90, 2, 82, 0))]
    for name, data in cases:
        p = J.ParsedImage(data)
        ref = oracle.decode(data)
        hs, vs = (p.image.hsamp[0], p.image.vsamp[0])
        mcux, mcuy, _, mh = shard.mcu_geometry(p.width, p.height, hs, vs)
        bpm = 1 if p.image.ncomp == 1 else hs * vs + 2
        for world in ((1, 3) if name == "large.jpg" else (1, 2, 5)):
            per = (mcuy + world - 1) // world
            bands = [(r0, min(per, mcuy - r0)) for r0 in range(0, mcuy, per)]
            got = []
            for (r0, n) in bands:
                b = J.Batch(ctx, [p], rows=[(r0, n)])
                outs = b.alloc_rgb()
                assert outs[0].shape[0] == min(n * mh, p.height - r0 * mh)
                b.decode(outs)
                assert not b.finish().any()
                got.append(outs[0][:, : p.width * 3].reshape(-1, p.width, 3).cpu().numpy())
                q = b.coefficients(0)
                lo, hi = r0 * mcux * bpm, (r0 + n) * mcux * bpm
                assert np.array_equal(q[lo:hi], ref.qcoef[lo:hi]), (name, world, r0)
                assert not q[:lo].any() and not q[hi:].any()  # outside the window nothing was produced: exported as zeros
                dq = b.coefficients(0, dequantised=True)
                assert np.array_equal(dq[lo:hi], ref.dqcoef[lo:hi]) and not dq[:lo].any() and not dq[hi:].any()
                b.close()
            assert np.array_equal(np.concatenate(got, 0), ref.rgb), (name, world)
        with pytest.raises(J.DecodeError):
            J.Batch(ctx, [p], rows=[(mcuy, 1)])
        with pytest.raises(J.DecodeError):
            J.Batch(ctx, [p], rows=[(1, mcuy)])
    # a batch where only some images have a window
    files = [c[1] for c in cases[1:]] + [synth.encode(200, 120, 2, 2, 90, 0, 83, 4)]
    imgs = [J.ParsedImage(f) for f in files]
    rows = [(3, 10), None, (0, 0), (2, 3)]
    b = J.Batch(ctx, imgs, rows=rows)
    outs = b.alloc_rgb()
    b.decode(outs)
    assert not b.finish().any()
    for f, im, o, r, i in zip(files, imgs, outs, rows, range(4)):
        ref = oracle.decode(f).rgb
        mh = 8 * im.image.vsamp[0]
        y0 = r[0] * mh if r and r[1] else 0
        assert np.array_equal(o[:, : im.width * 3].reshape(-1, im.width, 3).cpu().numpy(), ref[y0: y0 + b.window_height(i)]), i
    b.close()


def decode_split(ctx, data, nparts, lead_bytes, subseq_bits=0):
    """One image without restart markers decoded as `nparts` parts of its stream, the way `nparts` ranks would: every
    part's front on its own, the reports exchanged (here: a list), every part resolved and decoded, the pixels of the
    MCUs each part owns assembled.  Returns (rgb HxWx3, coefficients [blocks, 64], parts' windows, byte ranges)."""
    import torch
    p = J.ParsedImage(data)
    hs, vs = p.image.hsamp[0], p.image.vsamp[0]
    mcux, mcuy, mw, mh = shard.mcu_geometry(p.width, p.height, hs, vs)
    parts = [J.RangeBatch(ctx, p, r, nparts, lead_bytes, subseq_bits) for r in range(nparts)]
    try:
        reports = [q.front() for q in parts]
        wins = [q.resolve(reports) for q in parts]
        assert [w[:2] for w in wins] == shard.split_windows(reports, mcux * mcuy)
        pitch = (p.width * 3 + 15) // 16 * 16
        full = torch.zeros((p.height, pitch), dtype=torch.uint8, device="cuda:0")
        coef = np.zeros((p.nblocks, 64), np.int16)
        bpm = p.nblocks // (mcux * mcuy)
        for q, (m0, n, row0, nrows) in zip(parts, wins):
            win = torch.empty((max(q.window_height(), 1), pitch), dtype=torch.uint8, device="cuda:0")
            q.decode(win)
            assert not q.finish().any()
            pieces = shard.mcu_range_pieces(m0, n, mcux, mw, mh, p.width, p.height)
            shard.unpack_pieces(full, shard.pack_pieces(win, row0 * mh, pieces), pieces)
            coef[m0 * bpm: (m0 + n) * bpm] = q.coefficients()
        rgb = full[:, : p.width * 3].reshape(p.height, p.width, 3).cpu().numpy()
        return rgb, coef, wins, [q.segments() for q in parts]
    finally:
        for q in parts:
            q.close()


def test_stream_without_restart_markers_split_into_parts(ctx):
    """The serial chain of src/decoder.rs:376-389 cut into parts by bytes: every part unstuffs and walks only its own
    bytes plus a lead-in, starts somewhere inside an MCU at an arbitrary bit with unknown predictors, and the
    exchange of one small report per part gives it its first MCU and its predictors.  The assembled result must be
    the oracle's, pixels and coefficients, for 4:4:4 and subsampled layouts, with parts that own whole rows, pieces of
    rows, or nothing at all."""
    cases = [(fixture_bytes("large.jpg"), (2, 3, 8), 16384), (fixture_bytes("porsche.jpg"), (2, 5), 4096),
             (synth.encode(1000, 600, 2, 2, 90, 0, 91, 0), (2, 4, 7), 8192), (synth.encode(640, 480, 2, 1, 85, 2, 92, 0), (3,), 8192),
             (synth.encode(333, 211, 0, 0, 90, 0, 93, 0), (2,), 2048),
             (synth.encode(64, 48, 2, 2, 90, 0, 94, 0), (1, 8), 4096), (synth.encode(8, 8, 1, 1, 90, 0, 95, 0), (3,), 64)]
    for data, counts, lead in cases:
        ref = oracle.decode(data)
        scan_len = oracle.probe(data).scan_len
        for nparts in counts:
            rgb, coef, wins, ranges = decode_split(ctx, data, nparts, lead)
            assert np.array_equal(coef, ref.qcoef), (len(data), nparts)
            assert np.array_equal(rgb, ref.rgb), (len(data), nparts)
            # every part reads its share of the stream plus the lead-in and a short tail, not the whole stream
            for r, (lo, hi) in enumerate(ranges):
                assert hi - lo <= scan_len // nparts + lead + 4096 + 128, (r, lo, hi)
            assert sum(w[1] for w in wins) == ref.info.mcux * ref.info.mcuy


def test_split_parts_that_do_not_join_up_are_refused(ctx):
    """A lead-in far shorter than the distance this stream needs to synchronise (a flat image: one short code per
    block, any bit looks like a block start): the chain of boundaries does not validate, every part learns so from
    the same reports, and the caller falls back to row windows -- nothing is decoded from a wrong start."""
    data = fixture_bytes("test-baseline.jpg")
    p = J.ParsedImage(data)
    parts = [J.RangeBatch(ctx, p, r, 4, 64) for r in range(4)]
    try:
        reports = [q.front() for q in parts]
        refused = 0
        for q in parts:
            with pytest.raises(J.ChainError):
                q.resolve(reports)
            refused += 1
        assert refused == 4
    except J.ChainError:
        pass  # a part may already find no block behind its boundary: the same refusal, one call earlier
    finally:
        for q in parts:
            q.close()
    # with a lead-in that covers the distance the same stream splits fine
    ref = oracle.decode(data)
    rgb, coef, _, _ = decode_split(ctx, data, 4, 65536)
    assert np.array_equal(coef, ref.qcoef) and np.array_equal(rgb, ref.rgb)


def test_random_images_match_the_oracle(ctx):
    """Three batches of the randomised sweep (tests/sweep_parity.py): 64 random valid images each -- sizes, sampling
    layouts, qualities, content classes, restart intervals, table ids -- decoded with two subsequence sizes."""
    import sweep_parity
    total, bad, _ = sweep_parity.run(3, 11, ctx=ctx, verbose=False)
    assert total == 3 * 64 * 2 and bad == 0
