"""Kernel times of one decode of the benchmark batch without looking at the status words (for timing
experiments that break the result on purpose)."""
import sys, os
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import jpeg_decoder_amd as J
from jpeg_decoder_amd import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
with ThreadPoolExecutor(16) as ex:
    files = list(ex.map(lambda s: synth.encode(1920, 1080, 2, 2, 90, 0, s, 0), range(16)))
parsed = [J.ParsedImage(f) for f in files]
ctx = J.Context(0)
b = J.Batch(ctx, [parsed[i % 16] for i in range(N)])
outs = b.alloc_rgb()
b.set_profiling(True)
acc = {}
for rep in range(4):
    b.decode(outs); torch.cuda.synchronize()
    if rep:
        for k, v in b.kernel_times().items(): acc[k] = acc.get(k, 0) + v / 3
print({k: round(v, 3) for k, v in acc.items()}, flush=True)
