"""Scratch: which blocks of the fuzzed images differ from the oracle before the oracle's first bad block."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import jpeg_decoder_amd as J
from jpeg_decoder_amd import synth
from oracle import oracle
from jpeg_decoder_amd import binding as B
if len(sys.argv) > 1: B.LIB_PATH = os.path.abspath(sys.argv[1])
ctx = J.Context(0)
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
        files.append(bytes(d))
imgs, kept = [], []
for f in files:
    try:
        imgs.append(J.ParsedImage(f)); kept.append(f)
    except J.DecodeError:
        pass
for sb in (0, 512, 16384):
    rgb, st, coef = J.decode_images(ctx, imgs, subseq_bits=sb, want_coef=True)
    for k, (f, im) in enumerate(zip(kept, imgs)):
        o = oracle.decode(f)
        if o.status == 0:
            continue
        n = int(o.info.err_block)
        bad = np.nonzero((coef[k][:n] != o.qcoef).any(axis=1))[0]
        if len(bad):
            b0 = bad[0]
            print(f"sb {sb} image {k}: oracle status {o.status} err_block {n} of {im.nblocks}, gpu status {st[k]}, {len(bad)} blocks differ, first {bad[:8]}, ri {im.image.restart_interval}")
            print("  gpu   ", coef[k][b0][:24].tolist())
            print("  oracle", o.qcoef[b0][:24].tolist())
            i = oracle.probe(f)
            print("  scan_len", i.scan_len, "file len", len(f))
