"""Kernel times of the 16384x16384 restart-marked image (config 5, one GPU) for several builds of the library.
Usage: c5_ab.py <a.so> <b.so> ...; each build runs in a child process."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    import jpeg_decoder_amd as J
    from jpeg_decoder_amd import synth, binding as B
    B.LIB_PATH = os.path.abspath(sys.argv[2])  # this scratch script only: pick the build under test
    data = synth.encode_tiled(16384, 1024, 16, 2, 2, 90, 0, 77, 16)
    p = J.ParsedImage(data)
    ctx = J.Context(0)
    b = J.Batch(ctx, [p])
    outs = b.alloc_rgb()
    b.set_profiling(True)
    acc = {}
    for rep in range(6):
        b.decode(outs); torch.cuda.synchronize()
        if rep:
            for k, v in b.kernel_times().items(): acc[k] = acc.get(k, 0) + v / 5
    assert not b.finish().any()
    print(json.dumps({k: round(v, 3) for k, v in acc.items()}))
else:
    for lib in sys.argv[1:]:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib], capture_output=True, text=True, timeout=300)
        print(os.path.basename(lib), out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
