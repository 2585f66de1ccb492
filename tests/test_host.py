"""CPU tests of the host side: container parse, geometry, restart bookkeeping, table construction and
the lane-by-lane walk of the stage-1 algorithm against the oracle; the C ABI library loads and
exports everything include/zj.h declares."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import band_file, FIXTURES, ROOT, fixture_bytes
from oracle import oracle

import jpeg_decoder_amd as J
from jpeg_decoder_amd import binding as B, synth


class SimStats(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in
                ["rounds_max", "rounds_sum", "nchunks", "fix_steps_max", "fix_steps_sum", "need_more", "verify_fail"]]


def sim():
    S = C.CDLL(os.path.join(ROOT, "tests", "sim", "libzjsim.so"))
    S.zjsim_decode.argtypes = [C.POINTER(B.zj_image), C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(SimStats)]
    return S


def sim_decode(data, subbits):
    p = J.ParsedImage(data)
    q = np.zeros((p.nblocks, 64), np.int16)
    st, ss = C.c_uint32(), SimStats()
    rc = sim().zjsim_decode(C.byref(p.parsed.image), subbits, q.ctypes.data_as(C.c_void_p), C.byref(st), C.byref(ss))
    return rc, st.value, q, ss


def test_library_exports_header_symbols():
    L = B.lib()
    header = open(os.path.join(ROOT, "include", "zj.h")).read()
    declared = set(re.findall(r"\b(zj_[a-z0-9_]+)\s*\(", header))
    declared -= {"zj_huff", "zj_image", "zj_parsed", "zj_output", "zj_ctx", "zj_batch"}
    assert declared, "no declarations found"
    assert declared == set(B.exported_symbols())
    for s in declared:
        assert hasattr(L, s), s


@pytest.mark.parametrize("name", FIXTURES)
def test_parse_matches_oracle_probe(name):
    data = fixture_bytes(name)
    p = J.ParsedImage(data)
    i = oracle.probe(data)
    assert (p.width, p.height, p.nblocks) == (i.w, i.h, i.nblocks)
    assert p.image.scan_len == i.scan_len
    assert (p.image.hsamp[0], p.image.vsamp[0]) == (i.hs[0], i.vs[0])
    assert [p.image.td[c] for c in range(3)] == [0, 1, 1] and [p.image.ta[c] for c in range(3)] == [0, 1, 1]


def test_parse_rejects_garbage_and_unsupported():
    with pytest.raises(J.DecodeError) as e:
        J.ParsedImage(b"not a jpeg at all")
    assert e.value.kind == "format"
    data = bytearray(fixture_bytes("porsche.jpg"))
    data[data.index(b"\xff\xc0") + 1] = 0xC2  # progressive SOF
    with pytest.raises(J.DecodeError) as e:
        J.ParsedImage(bytes(data))
    assert e.value.kind == "unsupported"
    with pytest.raises(J.DecodeError):
        J.ParsedImage(fixture_bytes("porsche.jpg")[:300])


def test_restart_offsets_and_bands():
    data = synth.encode(256, 128, 2, 2, seed=5, restart_interval=8)  # 16 MCUs per row -> 2 intervals per row
    p = J.ParsedImage(data)
    assert p.image.restart_interval == 8 and p.parsed.n_restart_markers == 16 * 8 // 8 - 1
    offs = (C.c_size_t * 64)()
    n = B.lib().zj_restart_offsets(C.byref(p.parsed.image), offs, 64)
    assert n == 16 and offs[0] == 0
    scan = C.string_at(p.image.scan, p.image.scan_len)
    for k in range(1, n):
        assert scan[offs[k] - 2] == 0xFF and 0xD0 <= scan[offs[k] - 1] <= 0xD7
    full = oracle.decode(data)
    # a band is a valid standalone image: a file built around it decodes to the same rows of the whole image
    band = p.band(2, 3)
    assert (band.width, band.height) == (256, 48)
    alone = oracle.decode(band_file(data, band))
    assert alone.status == 0 and np.array_equal(alone.rgb, full.rgb[32:80])
    assert np.array_equal(alone.qcoef, full.qcoef[2 * 16 * 6: 5 * 16 * 6])
    with pytest.raises(J.DecodeError):
        J.ParsedImage(synth.encode(256, 128, 2, 2, seed=5, restart_interval=5)).band(1, 2)


SIM_CASES = [(f, sb) for f in ["porsche.jpg", "profile.jpg", "test-baseline.jpg", "out.jpg"] for sb in (512, 1024, 4096, 65536)]


@pytest.mark.parametrize("name,subbits", SIM_CASES)
def test_stage1_walk_fixtures(name, subbits):
    """Self-synchronising decode, stitched and scanned as the kernels do it, reproduces the oracle's
    quantised coefficients bit for bit."""
    data = fixture_bytes(name)
    rc, st, q, ss = sim_decode(data, subbits)
    assert rc == 0 and st == 0 and ss.verify_fail == 0
    assert np.array_equal(q, oracle.decode(data, want_rgb=False).qcoef)


@pytest.mark.parametrize("params", [
    (320, 240, 2, 2, 90, 0, 1, 0), (200, 120, 1, 1, 90, 0, 2, 0), (150, 90, 2, 1, 75, 0, 3, 0),
    (90, 150, 1, 2, 75, 0, 4, 0), (333, 211, 2, 2, 90, 0, 5, 4), (256, 256, 2, 2, 95, 2, 6, 0),
    (512, 384, 2, 2, 50, 1, 7, 0), (64, 64, 1, 1, 90, 0, 8, 1), (8, 8, 1, 1, 90, 0, 9, 0), (17, 9, 2, 2, 90, 0, 10, 1),
    (300, 200, 0, 0, 90, 0, 11, 0), (65, 33, 0, 0, 75, 2, 12, 2),
])
def test_stage1_walk_synthetic(params):
    data = synth.encode(*params)
    want = oracle.decode(data, want_rgb=False).qcoef
    for sb in (512, 2048, 20000):
        rc, st, q, ss = sim_decode(data, sb)
        assert rc == 0 and st == 0 and ss.verify_fail == 0
        assert np.array_equal(q, want)


def test_segment_layout_properties():
    """Every segment is covered by its subsequences; long segments fill whole 64-lane chunks evenly, short ones
    get a power-of-two group of lanes that holds all their subsequences."""
    S = sim()
    S.zjsim_seg_layout.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32] + [C.POINTER(C.c_uint32)] * 3
    S.zjsim_seg_layout.restype = None
    rng = np.random.default_rng(3)
    sb, ns, ln = C.c_uint32(), C.c_uint32(), C.c_uint32()
    cases = [(0, 1024, 1024), (1, 512, 512), (511, 512, 512), (512, 512, 512), (32 * 1024 - 1, 1024, 16384), (32 * 1024, 1024, 16384)]
    for _ in range(3000):
        target = int(rng.choice([512, 1024, 2048, 4096, 65536, 20000]))
        cases.append((int(rng.integers(0, 1 << int(rng.integers(4, 31)))), target, max(target, int(rng.choice([512, 4096, 16384])))))
    for bits, target, small in cases:
        S.zjsim_seg_layout(bits, target, small, C.byref(sb), C.byref(ns), C.byref(ln))
        assert sb.value % 512 == 0 and sb.value >= 512
        assert ns.value * sb.value >= bits and (ns.value == 0 or (ns.value - 1) * sb.value < bits)
        if bits >= 32 * max(target, 512):
            assert ln.value == 64
            chunks = (ns.value + 63) // 64
            assert chunks * 64 - ns.value < 64 and sb.value < 2 * max(target, 512) + 512
        else:
            assert ln.value in (1, 2, 4, 8, 16, 32) and ns.value <= ln.value


def test_stage1_walk_reports_corruption():
    data = bytearray(synth.encode(96, 96, 2, 2, seed=3))
    i = oracle.probe(bytes(data))
    cut = bytes(data[: i.scan_off + i.scan_len // 3]) + b"\xff\xd9"
    rc, st, _, _ = sim_decode(cut, 1024)
    assert rc == 0 and (st & B.ZJ_ST_TRUNCATED)


def test_cli_and_cpp_decoder_exist():
    assert os.access(os.path.join(ROOT, "jpeg-decoder_amd", "zj_decode"), os.X_OK)


def test_no_gpu_is_a_loud_error():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert B.lib().zj_device_count() == 0
    assert not B.lib().zj_create(0, None)
    with pytest.raises(J.DecodeError) as e:
        J.Context(0)
    assert e.value.kind == "device"


def test_null_handles_are_errors_not_crashes():
    """Nothing unwinds or faults across the C ABI: calls on missing handles return an error code, a null
    pointer or zero (the reference panics instead, src/decoder.rs:358 and the unwraps around it)."""
    L = B.lib()
    st = (C.c_uint32 * 1)()
    ms = (C.c_float * B.ZJ_NUM_KERNELS)()
    v = [C.c_uint64() for _ in range(4)]
    assert not L.zj_batch_create(None, None, 0, 0)
    assert L.zj_batch_decode(None, None, None) == B.ZJ_E_INVALID
    assert L.zj_batch_decode_entropy(None) == B.ZJ_E_INVALID
    assert L.zj_batch_decode_pixels(None, None, None, None) == B.ZJ_E_INVALID
    assert L.zj_batch_finish(None, st) == B.ZJ_E_INVALID
    assert not L.zj_batch_qcoef(None, 0)
    assert L.zj_batch_export_coef(None, 0, None, None, B.ZJ_MEM_HOST) == B.ZJ_E_INVALID
    assert L.zj_batch_segments(None, 0, None, 0) == 0
    assert L.zj_batch_set_profiling(None, 1) == B.ZJ_E_INVALID
    assert L.zj_batch_kernel_times(None, ms) == B.ZJ_E_INVALID
    assert L.zj_batch_bytes(None, *[C.byref(x) for x in v]) == B.ZJ_E_INVALID
    assert L.zj_decode_batch(None, None, 0, None, st) == B.ZJ_E_INVALID
    assert L.zj_last_error(None) == b"no context"
    L.zj_batch_destroy(None)
    L.zj_destroy(None)
    assert L.zj_kernel_name(-1) == b"" and L.zj_kernel_name(B.ZJ_NUM_KERNELS) == b""
    assert L.zj_write_ppm(None, 1, 1, None, 0) == B.ZJ_E_INVALID


def test_ac_value_fast_form_equals_extend():
    """The compare-free EXTEND used by the block writer equals the reference form (src/ec.rs sign_code) for
    every AC size 0..15: all value patterns of the size, with random bits behind them."""
    S = sim()
    S.zjsim_value_ac.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    rng = np.random.default_rng(7)
    f, r = C.c_int32(), C.c_int32()
    for nbits in range(0, 16):
        vals = range(1 << nbits) if nbits <= 10 else rng.integers(0, 1 << nbits, 2048)
        for v in vals:
            tail = int(rng.integers(0, 1 << (32 - nbits))) if nbits < 32 else 0
            x = ((int(v) << (32 - nbits)) | tail) & 0xFFFFFFFF if nbits else tail
            assert S.zjsim_value_ac(x, nbits, C.byref(f), C.byref(r)), (nbits, v, f.value, r.value)
        # edge patterns
        for x in (0, 0xFFFFFFFF, 0x80000000, 0x7FFFFFFF):
            assert S.zjsim_value_ac(x, nbits, C.byref(f), C.byref(r)), (nbits, x)


def test_lookup_tables_equal_canonical_search():
    """Two-level tables (and their walk-entry form) against the canonical search of src/ec.rs:30-47 on random
    bit windows: the reference's own tables, random valid prefix codes, and tables built without a second
    level, where every long code takes the canonical fallback."""
    S = sim()
    S.zjsim_lookup_check.argtypes = [C.POINTER(B.zj_huff), C.c_uint32, C.c_uint32, C.c_uint64]
    tables = []
    for name in ("porsche.jpg", "profile.jpg", "test-baseline.jpg"):
        p = J.ParsedImage(fixture_bytes(name))
        for i in range(2):
            for k in range(2):
                if p.image.ht[i][k]:
                    h = B.zj_huff()  # a copy: the parsed image owns the original
                    C.memmove(C.byref(h), p.image.ht[i][k], C.sizeof(h))
                    tables.append(h)
    rng = np.random.default_rng(11)
    for _ in range(40):
        # random complete-or-incomplete prefix code: spend a Kraft budget of 2^16 over lengths 1..16
        counts, budget, nsym = [0] * 16, 1 << 16, 0
        for l in range(1, 17):
            unit = 1 << (16 - l)
            room = min(budget // unit, 256 - nsym)
            want = int(rng.integers(0, 3 + l)) if l < 14 else int(rng.integers(0, 120))
            c = min(room, want)
            counts[l - 1] = c
            budget -= c * unit
            nsym += c
        if nsym == 0:
            continue
        h = B.zj_huff()
        for i in range(16):
            h.counts[i] = counts[i]
        syms = rng.permutation(256)[:nsym]
        for i, v in enumerate(syms):
            h.symbols[i] = int(v)
        tables.append(h)
    assert len(tables) > 20
    for h in tables:
        for cap in (256, 64, 0):
            assert S.zjsim_lookup_check(C.byref(h), cap, 20000, 5) == 0


def test_pair_entries_walk_like_single_symbols():
    """The sync pass looks up to two symbols at a time (pair entries).  On random bit streams, from random states,
    window by window, that walk must stop in the same states and mark the same block starts as the
    symbol-by-symbol walk: the fixtures' tables (test-baseline.jpg has single-code tables), the synthetic
    generator's tables, random prefix codes, and every table without its second level (canonical search)."""
    S = sim()
    S.zjsim_pair_walk_check.argtypes = [C.POINTER(B.zj_huff), C.POINTER(B.zj_huff), C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint32)]
    S.zjsim_pair_walk_check.restype = C.c_int
    pairs = []
    files = [fixture_bytes(n) for n in ("porsche.jpg", "profile.jpg", "test-baseline.jpg", "out.jpg")] + [synth.encode(64, 64, 2, 2, seed=1)]
    for data in files:
        p = J.ParsedImage(data)
        for tid in range(2):
            if not (p.image.ht[tid][0] and p.image.ht[tid][1]):
                continue
            ac, dc = B.zj_huff(), B.zj_huff()
            C.memmove(C.byref(ac), p.image.ht[tid][0], C.sizeof(ac))
            C.memmove(C.byref(dc), p.image.ht[tid][1], C.sizeof(dc))
            for cap in (256, 0):
                n = C.c_uint32()
                assert S.zjsim_pair_walk_check(C.byref(dc), C.byref(ac), cap, 300, 7 + tid, C.byref(n)) == 0
                pairs.append(n.value)
    assert max(pairs) > 300  # the tables of real files do hold pairs: most 10-bit windows begin with two short AC codes
    rng = np.random.default_rng(12)
    for _ in range(25):
        hs = []
        for _t in range(2):
            counts, budget, nsym = [0] * 16, 1 << 16, 0
            for l in range(1, 17):
                unit = 1 << (16 - l)
                c = min(budget // unit, 256 - nsym, int(rng.integers(0, 3 + l)) if l < 14 else int(rng.integers(0, 120)))
                counts[l - 1] = c
                budget -= c * unit
                nsym += c
            h = B.zj_huff()
            for i in range(16):
                h.counts[i] = counts[i]
            for i, v in enumerate(rng.permutation(256)[:nsym]):
                h.symbols[i] = int(v)
            hs.append((h, nsym))
        if min(n for _, n in hs) == 0:
            continue
        for cap in (256, 0):
            assert S.zjsim_pair_walk_check(C.byref(hs[0][0]), C.byref(hs[1][0]), cap, 100, 3, None) == 0
