"""Batch planning time with the marker search on the device and on the host."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import jpeg_decoder_amd as J
from jpeg_decoder_amd import synth, binding as B
ctx = J.Context(0)
cases = {"256 x 1080p DRI 8": [synth.encode(1920, 1080, 2, 2, 90, 0, s, 8) for s in range(8)] * 32,
         "1 x 8192x8192 DRI 16": [synth.encode(8192, 8192, 2, 2, 90, 0, 1, 16)]}
for name, files in cases.items():
    imgs = [J.ParsedImage(f) for f in files]
    for flag, lab in ((0, "device"), (B.ZJ_FLAG_HOST_MARKER_SCAN, "host")):
        for rep in range(2):
            t0 = time.perf_counter(); b = J.Batch(ctx, imgs, flag); t1 = time.perf_counter(); b.close()
        print(f"{name}: {lab} marker search, zj_batch_create {1e3 * (t1 - t0):.1f} ms, scan bytes {sum(len(f) for f in files) / 1e6:.1f} MB", flush=True)
