import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import jpeg_decoder_amd as J
from jpeg_decoder_amd import synth
files = [synth.encode(4096, 2160, 2, 2, 90, 0, s, 0) for s in range(8)]
parsed = [J.ParsedImage(f) for f in files]
ctx = J.Context(0)
for n in (512, 1024, 4096):
    b = J.Batch(ctx, [parsed[i % 8] for i in range(n)])
    outs = b.alloc_rgb()
    b.decode(outs); torch.cuda.synchronize()
    t0 = time.perf_counter(); b.decode(outs); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    b.set_profiling(True)
    b.decode(outs); torch.cuda.synchronize()
    kt = b.kernel_times()
    print(n, "host call %.2f ms, until done %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3), {k: round(v, 2) for k, v in kt.items()}, flush=True)
    assert not b.finish().any()
    del b, outs
    torch.cuda.empty_cache()
