"""A/B timing of two builds of the library on one GPU box: kernel times of the benchmark batch, the two
builds alternating.  Usage: ab_kernel_ms.py <images> <a.so> <b.so>; each build runs in a child process."""
import sys, os, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    from concurrent.futures import ThreadPoolExecutor
    import torch
    import jpeg_decoder_amd as J
    from jpeg_decoder_amd import synth, binding as B
    B.LIB_PATH = os.path.abspath(sys.argv[3])  # this scratch script only: pick the build under test
    N = int(sys.argv[2])
    with ThreadPoolExecutor(16) as ex:
        files = list(ex.map(lambda s: synth.encode(1920, 1080, 2, 2, 90, 0, s, 0), range(16)))
    parsed = [J.ParsedImage(f) for f in files]
    ctx = J.Context(0)
    b = J.Batch(ctx, [parsed[i % 16] for i in range(N)])
    outs = b.alloc_rgb()
    b.set_profiling(True)
    acc = {}
    for rep in range(6):
        b.decode(outs); torch.cuda.synchronize()
        if rep:
            for k, v in b.kernel_times().items(): acc[k] = acc.get(k, 0) + v / 5
    print(json.dumps({k: round(v, 3) for k, v in acc.items()}))
else:
    n, a, bb = sys.argv[1], sys.argv[2], sys.argv[3]
    for rnd in range(2):
        for tag, lib in (("A", a), ("B", bb)):
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", n, lib], capture_output=True, text=True, timeout=300)
            print(tag, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
