import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, jpeg_decoder_amd as J
ctx = J.Context(0)
for name in sys.argv[1:]:
    p = J.ParsedImage(open(os.path.join(ROOT, "tests", "golden", name), "rb").read())
    for sb in (0, 4096, 16384):
        b = J.Batch(ctx, [p], sb); outs = b.alloc_rgb()
        b.decode(outs); b.finish(); b.set_profiling(True)
        acc = {}
        for _ in range(3):
            b.decode(outs)
            for k, v in b.kernel_times().items(): acc[k] = acc.get(k, 0) + v / 3
        print(name, sb, {k: round(v, 3) for k, v in acc.items()}, flush=True)
        b.close()
