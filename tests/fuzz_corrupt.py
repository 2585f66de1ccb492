"""Robustness run on the GPU (by hand: python tests/fuzz_corrupt.py; not collected by pytest): corrupted entropy data must
produce status bits, never a fault or a hang."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import jpeg_decoder_amd as J
from jpeg_decoder_amd import synth
from oracle import oracle
ctx = J.Context(0)
rng = np.random.default_rng(123)
bases = [synth.encode(256, 256, 2, 2, 90, 0, 1, 0), synth.encode(200, 120, 1, 1, 90, 2, 2, 8), synth.encode(320, 240, 2, 1, 75, 1, 3, 0)]
total = 0
for rnd in range(6):
    files = []
    for b in bases:
        info = oracle.probe(b)
        for _ in range(40):
            d = bytearray(b)
            nflip = int(rng.integers(1, 30))
            for k in rng.integers(info.scan_off, info.scan_off + info.scan_len - 2, nflip):
                mode = rng.integers(0, 3)
                if mode == 0: d[k] ^= 1 << int(rng.integers(0, 8))
                elif mode == 1: d[k] = int(rng.integers(0, 256))
                else: d[k] = 0xFF
            if rng.integers(0, 4) == 0:
                cut = int(rng.integers(info.scan_off + 4, info.scan_off + info.scan_len))
                d = d[:cut] + b"\xff\xd9"
            files.append(bytes(d))
    imgs = []
    for f in files:
        try: imgs.append(J.ParsedImage(f))
        except J.DecodeError: pass
    for sb in (0, 512, 16384):
        rgb, st = J.decode_images(ctx, imgs, subseq_bits=sb)
        total += len(imgs)
        print(f"round {rnd} subbits {sb}: {len(imgs)} images, clean {int((st == 0).sum())}, status bits seen {sorted(set(int(x) for x in st))}", flush=True)
print("done", total)
