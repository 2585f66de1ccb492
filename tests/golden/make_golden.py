"""Regenerates tests/golden/golden.json from the CPU oracle.

The reference ships no tests or expected outputs and cannot be built in this image (Rust), so these
vectors pin the ORACLE (oracle/zj_oracle.c), not the reference: they freeze its output on the
reference's five sample images (copied here as data) and on seeded synthetic images, so that any
later change of oracle or compiler shows up.  What ties the oracle to the reference are the
structural invariants stored next to the hashes (block, symbol and short-path counts, bytes
consumed, FF00 pairs), which were measured independently on the reference's algorithm.
Run:  python tests/golden/make_golden.py
"""
import json
import os
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import jpeg_decoder_amd  # noqa: E402,F401
from jpeg_decoder_amd import synth  # noqa: E402
from oracle import oracle  # noqa: E402

SYNTH_CASES = [
    # name, w, h, hs, vs, quality, content, seed, restart_interval
    ("s420_320x240", 320, 240, 2, 2, 90, 0, 1, 0),
    ("s444_200x120", 200, 120, 1, 1, 90, 0, 2, 0),
    ("s422_150x90", 150, 90, 2, 1, 75, 0, 3, 0),
    ("s440_90x150", 90, 150, 1, 2, 75, 0, 4, 0),
    ("s420_rst4_333x211", 333, 211, 2, 2, 90, 0, 5, 4),
    ("s420_noise_256x256", 256, 256, 2, 2, 95, 2, 6, 0),
    ("s420_smooth_512x384", 512, 384, 2, 2, 50, 1, 7, 0),
    ("s444_rst1_64x64", 64, 64, 1, 1, 90, 0, 8, 1),
    ("gray_300x200", 300, 200, 0, 0, 90, 0, 11, 0),        # hs = vs = 0: one component
    ("gray_rst2_65x33", 65, 33, 0, 0, 75, 2, 12, 2),
]


def entry(data):
    d = oracle.decode(data)
    i = d.info
    return {
        "bytes": len(data), "file_crc": zlib.crc32(data),
        "w": i.w, "h": i.h, "hs": i.hs[0], "vs": i.vs[0], "nblocks": i.nblocks,
        "restart_interval": i.restart_interval, "scan_len": i.scan_len, "scan_consumed": i.scan_consumed,
        "nsymbols": i.nsymbols, "nshort": i.nshort, "nff00": i.nff00,
        "qcoef_crc": zlib.crc32(d.qcoef.tobytes()), "dqcoef_crc": zlib.crc32(d.dqcoef.tobytes()),
        "rgb_crc": zlib.crc32(d.rgb.tobytes()),
        "max_abs_dq": int(abs(d.dqcoef.astype(int)).max()),
        "rgb_first_row": d.rgb[0, :8].reshape(-1).tolist(),
    }


def main():
    out = {"fixtures": {}, "synthetic": {}}
    for name in ["porsche.jpg", "profile.jpg", "test-baseline.jpg", "out.jpg", "large.jpg"]:
        out["fixtures"][name] = entry(open(os.path.join(HERE, name), "rb").read())
    for (name, w, h, hs, vs, q, c, seed, ri) in SYNTH_CASES:
        data = synth.encode(w, h, hs, vs, q, c, seed, ri)
        e = entry(data)
        e["params"] = [w, h, hs, vs, q, c, seed, ri]
        out["synthetic"][name] = e
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote golden.json")


if __name__ == "__main__":
    main()
