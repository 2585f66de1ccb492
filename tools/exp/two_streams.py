"""Experiment: does running two half batches on two streams beat one full batch on one stream?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import jpeg_decoder_amd as J
from jpeg_decoder_amd import synth
from concurrent.futures import ThreadPoolExecutor
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
with ThreadPoolExecutor(16) as ex:
    files = list(ex.map(lambda s: synth.encode(1920, 1080, 2, 2, 90, 0, s, 0), range(32)))
parsed = [J.ParsedImage(f) for f in files]
stride = (1920 * 3 + 15) // 16 * 16
def make(n, stream):
    ctx = J.Context(0, stream=stream.cuda_stream)
    b = J.Batch(ctx, [parsed[i % 32] for i in range(n)], 0)
    out = torch.empty((n, 1080, stride), dtype=torch.uint8, device="cuda:0")
    return ctx, b, [out[i] for i in range(n)]
s0, s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
full = make(N, s0)
ha, hb = make(N // 2, s1), make(N // 2, s2)
def run_full(k):
    for _ in range(k): full[1].decode(full[2])
    torch.cuda.synchronize()
def run_half(k, stagger):
    if stagger: ha[1].decode(ha[2])
    for _ in range(k):
        hb[1].decode(hb[2]); ha[1].decode(ha[2])
    if stagger: hb[1].decode(hb[2])
    torch.cuda.synchronize()
for name, fn in [("full", lambda: run_full(5)), ("2 halves", lambda: run_half(5, False)), ("2 halves staggered", lambda: run_half(5, True))]:
    fn(); t = time.perf_counter(); fn(); dt = time.perf_counter() - t
    steps = 5 if name != "2 halves staggered" else 5.5
    print(f"{name}: {dt / steps * 1e3:.2f} ms per {N} images", flush=True)
