import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, jpeg_decoder_amd as J
from jpeg_decoder_amd import synth
ctx = J.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
p = J.ParsedImage(synth.encode(n, n, 2, 2, 90, 0, 77, 16))
b = J.Batch(ctx, [p]); outs = b.alloc_rgb()
b.decode(outs); b.finish(); b.set_profiling(True)
acc = {}
for _ in range(3):
    b.decode(outs)
    for k, v in b.kernel_times().items(): acc[k] = acc.get(k, 0) + v / 3
print(n, {k: round(v, 3) for k, v in acc.items()}, flush=True)
