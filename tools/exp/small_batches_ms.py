"""Scratch: device time per decode of small inputs (two fixtures, 1 / 4 / 16 synthetic 1080p images), at the planner's
own subsequence size and at forced ones."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, jpeg_decoder_amd as J
from jpeg_decoder_amd import synth
ctx = J.Context(0)
def fix(name): return J.ParsedImage(open(os.path.join(ROOT, "tests", "golden", name), "rb").read())
imgs1080 = [J.ParsedImage(synth.encode(1920, 1080, 2, 2, 90, 0, s, 0)) for s in range(16)]
cases = {"porsche": [fix("porsche.jpg")], "large": [fix("large.jpg")], "1x1080p": imgs1080[:1], "4x1080p": imgs1080[:4], "16x1080p": imgs1080}
for name, imgs in cases.items():
    row = []
    for sb in [0] + [int(x) for x in sys.argv[1:]]:
        b = J.Batch(ctx, imgs, sb); outs = b.alloc_rgb()
        for _ in range(3): b.decode(outs)
        assert not b.finish().any()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st = torch.cuda.current_stream()
        torch.cuda.synchronize()
        N = 20
        t0 = time.perf_counter()
        for _ in range(N): b.decode(outs)
        b.finish()
        dt = (time.perf_counter() - t0) / N * 1e3
        row.append((sb, round(dt, 4)))
        b.close()
    print(name, row, flush=True)
