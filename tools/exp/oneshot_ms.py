"""Latency of the one-shot call (zj_decode_batch: plan, allocate, upload, decode, download, free) per image."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import jpeg_decoder_amd as J
ctx = J.Context(0)
for name in ("porsche.jpg", "profile.jpg", "large.jpg"):
    data = open(os.path.join(ROOT, "tests", "golden", name), "rb").read()
    p = J.ParsedImage(data)
    ts = []
    for _ in range(12):
        t0 = time.perf_counter()
        rgb, st = J.decode_images(ctx, [p])
        ts.append((time.perf_counter() - t0) * 1e3)
        assert st[0] == 0
    ts = sorted(ts[2:])
    print(f"{name}: {p.width}x{p.height} one-shot host-to-host median {ts[len(ts)//2]:.3f} ms, min {ts[0]:.3f} ms", flush=True)
