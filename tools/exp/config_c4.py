"""One-off measurement of the mixed batch: 50 % 1080p / 50 % 4K, 50 % 4:4:4 / 50 % 4:2:0 (index bits pick the
class), a one-GPU share of 1024 images; one image per class is checked against the CPU oracle."""
import sys, os, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import jpeg_decoder_amd as J
from jpeg_decoder_amd import synth
from oracle import oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
SEEDS = 4
def make(k):
    cls, seed = k % 4, k // 4
    w, h = ((1920, 1080), (3840, 2160))[cls & 1]
    s = (1, 2)[(cls >> 1) & 1]
    return synth.encode(w, h, s, s, 90, 0, 100 + seed, 0)
with ThreadPoolExecutor(16) as ex:
    files = list(ex.map(make, range(4 * SEEDS)))
parsed = [J.ParsedImage(f) for f in files]
images = [parsed[(i % 4) + 4 * ((i // 4) % SEEDS)] for i in range(N)]
ctx = J.Context(0)
b = J.Batch(ctx, images)
outs = b.alloc_rgb()
b.decode(outs); st = b.finish(); assert not st.any()
for k in range(4):
    r = oracle.decode(files[k])
    got = outs[k][:, : images[k].width * 3].reshape(images[k].height, images[k].width, 3).cpu().numpy()
    assert np.array_equal(got, r.rgb), k
    assert np.array_equal(b.coefficients(k), r.qcoef), k
print("one image per class equals the oracle (pixels and coefficients)", flush=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
reps = 5
for _ in range(reps): b.decode(outs)
b.finish(); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
px = sum(im.width * im.height for im in images)
bc = b.byte_counts()
A = bc["scan"] + 2 * bc["coef"] + bc["rgb"]
print(f"mixed batch of {N}: {dt * 1e3:.2f} ms per decode, {px / dt / 1e9:.1f} GMP/s, algorithmic bytes {A / 1e9:.1f} GB -> {A / dt / 1e12:.2f} TB/s", flush=True)
b.set_profiling(True); b.decode(outs); b.finish()
print({k: round(v, 3) for k, v in b.kernel_times().items()}, flush=True)
