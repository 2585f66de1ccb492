"""Randomised parity sweep on the GPU (by hand: python tests/sweep_parity.py [batches] [seed]; a short run of it is
part of the GPU tests): batches of random valid images -- sizes from 1x1 to about 700x700, every sampling layout, one-component
frames, qualities 30..98, the three content classes, restart intervals from one MCU to none, swapped table ids --
decoded in one batch per round and compared with the oracle, pixels and coefficients, bit for bit."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import jpeg_decoder_amd as J  # noqa: E402
from jpeg_decoder_amd import synth  # noqa: E402
from oracle import oracle  # noqa: E402

LAYOUTS = [(0, 0), (1, 1), (2, 1), (1, 2), (2, 2)]
RI = [0, 0, 0, 1, 2, 3, 7, 16, 50, 500]


def run(nbatches, seed, ctx=None, verbose=True, scale=1, per_batch=64):
    """Returns (decodes checked, mismatches, batches that left the fast path)."""
    rng = np.random.default_rng(seed)
    ctx = ctx or J.Context(0)
    bad = total = slow = 0
    for rnd in range(nbatches):
        specs = []
        for _ in range(per_batch):
            big = rng.integers(0, 4) == 0
            w = int(rng.integers(1, (700 if big else 120) * scale))
            h = int(rng.integers(1, (700 if big else 120) * scale))
            hs, vs = LAYOUTS[int(rng.integers(0, 5))]
            content = int(rng.integers(0, 3)) + (int(rng.integers(0, 8)) << 8 if hs else 0)
            specs.append((w, h, hs, vs, int(rng.integers(30, 99)), content, int(rng.integers(0, 1 << 30)), RI[int(rng.integers(0, len(RI)))]))
        with ThreadPoolExecutor(16) as ex:
            files = list(ex.map(lambda s: synth.encode(*s), specs))
            refs = list(ex.map(oracle.decode, files))
        imgs = [J.ParsedImage(f) for f in files]
        for sb in (0, 1024):
            b = J.Batch(ctx, imgs, subseq_bits=sb)
            outs = b.alloc_rgb()
            b.decode(outs)
            st = b.finish()
            for i, (im, ref) in enumerate(zip(imgs, refs)):
                rgb = outs[i][:, : im.width * 3].reshape(im.height, im.width, 3).cpu().numpy()
                ok = st[i] == 0 and np.array_equal(rgb, ref.rgb) and np.array_equal(b.coefficients(i), ref.qcoef)
                total += 1
                if not ok:
                    bad += 1
                    print("MISMATCH", specs[i], "subseq_bits", sb, "status", st[i], flush=True)
            if any(b.counters()):
                slow += 1
                if verbose:
                    print("left the fast path (results still checked):", b.counters(), flush=True)
            b.close()
        if verbose:
            print(f"batch {rnd}: {total} decodes checked, {bad} mismatches", flush=True)
    return total, bad, slow


if __name__ == "__main__":
    # python tests/sweep_parity.py [batches] [seed] [size scale] [images per batch]
    total, bad, slow = run(int(sys.argv[1]) if len(sys.argv) > 1 else 10, int(sys.argv[2]) if len(sys.argv) > 2 else 1,
                           scale=int(sys.argv[3]) if len(sys.argv) > 3 else 1, per_batch=int(sys.argv[4]) if len(sys.argv) > 4 else 64)
    print("done", total, "mismatches", bad, "batches off the fast path", slow)
    sys.exit(1 if bad else 0)
