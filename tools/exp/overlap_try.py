"""Does the IDCT/colour kernel of one sub-batch overlap usefully with the entropy passes of the next?
The benchmark batch in P sub-batches: entropy passes on the context's stream, pixel passes on a second one."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import jpeg_decoder_amd as J
from jpeg_decoder_amd import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
files = [synth.encode(1920, 1080, 2, 2, 90, 0, s, 0) for s in range(8)]
parsed = [J.ParsedImage(f) for f in files]
imgs = [parsed[i % 8] for i in range(N)]
torch.cuda.set_device(0)
ctx = J.Context(0)
pix = torch.cuda.Stream()

def timed(fn, reps=4):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) * 1e3)
    return " ".join("%.2f" % v for v in out)

whole = J.Batch(ctx, imgs)
outs = whole.alloc_rgb()
wt = whole.output_tables(outs)
print("one batch, one stream:", timed(lambda: whole.decode(wt)), flush=True)
assert not whole.finish().any()
del whole
for P in (2, 3, 4, 6, 8):
    per = N // P
    imgs_p = imgs[:per * P]
    subs = [J.Batch(ctx, imgs[i * per:(i + 1) * per]) for i in range(P)]
    o = [subs[i].output_tables(outs[i * per:(i + 1) * per]) for i in range(P)]
    def seq():
        for b, oo in zip(subs, o):
            b.decode(oo)
    def ovl():
        for b, oo in zip(subs, o):
            b.decode_entropy()
            b.decode_pixels(oo, pix.cuda_stream)
    print("P=%2d one stream: %s | pixel stream: %s" % (P, timed(seq), timed(ovl)), flush=True)
    for b in subs:
        assert not b.finish().any()
    del subs, b
