#!/usr/bin/env python3
"""Benchmark of the decode hot path: decoded megapixels/s on a batch of synthetic 1080p 4:2:0 q90
baseline JPEGs (4096 images per GPU by default), device-resident timing, one process per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one decode of the whole batch (all stage-1 kernels + the IDCT/colour kernel) with the
scans already resident in HBM and the pixels left in HBM.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--images", type=int, default=4096, help="images per GPU (weak scaling)")
    ap.add_argument("--distinct", type=int, default=64, help="distinct synthetic images (seeds), cycled")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--sampling", default="420", choices=["420", "444", "422", "440"])
    ap.add_argument("--quality", type=int, default=90)
    ap.add_argument("--content", type=int, default=0, help="0 natural, 1 smooth, 2 white noise")
    ap.add_argument("--subseq-bits", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --one-device: rehearsal of the multi-rank path on a box with a single GPU")
    ap.add_argument("--one-device", action="store_true", help="every rank uses GPU 0 (rehearsal only)")
    ap.add_argument("--cpu-sample", type=int, default=48, help="images decoded by the CPU baseline")
    ap.add_argument("--h2h-sample", type=int, default=0,
                    help="images of the optional host-to-host (PCIe-inclusive) leg, e.g. 512; 0 = skip (default, so that "
                         "every kernel launch of a default run has the benchmark's size)")
    return ap.parse_args()


def main():
    a = parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist

    import jpeg_decoder_amd as J
    from jpeg_decoder_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev = local_rank if (world > 1 and not a.one_device) else 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev)
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{dev}"))
        else:
            dist.init_process_group("gloo")
    torch.cuda.set_device(dev)

    hs, vs = {"420": (2, 2), "444": (1, 1), "422": (2, 1), "440": (1, 2)}[a.sampling]
    ndist = max(1, min(a.distinct, a.images))
    threads = min(16, os.cpu_count() or 1)
    with ThreadPoolExecutor(threads) as ex:
        files = list(ex.map(lambda s: synth.encode(a.width, a.height, hs, vs, a.quality, a.content, s, 0), range(ndist)))
    parsed = [J.ParsedImage(f) for f in files]
    images = [parsed[i % ndist] for i in range(a.images)]

    stream = torch.cuda.Stream(device=dev)
    ctx = J.Context(dev, stream=stream.cuda_stream)
    batch = J.Batch(ctx, images, a.subseq_bits)
    bc = batch.byte_counts()
    stride = (a.width * 3 + 15) // 16 * 16
    out_all = torch.empty((a.images, a.height, stride), dtype=torch.uint8, device=f"cuda:{dev}")
    outs = [out_all[i] for i in range(a.images)]

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(a.warmup):
        batch.decode(outs)
    st = batch.finish()
    assert not st.any(), f"decode reported status {np.unique(st)}"
    sync_all()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(a.steps):
        batch.decode(outs)
    ev1.record(stream)
    stream.synchronize()
    sync_all()
    t1 = time.perf_counter()
    wall = t1 - t0
    dev_ms = ev0.elapsed_time(ev1)
    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64, device=f"cuda:{dev}" if a.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    st = batch.finish()
    assert not st.any()

    # per-kernel durations, measured with HIP events on the launch stream around every kernel
    batch.set_profiling(True)
    ktimes = {}
    nprof = 3
    for _ in range(nprof):
        batch.decode(outs)
        for k, v in batch.kernel_times().items():
            ktimes[k] = ktimes.get(k, 0.0) + v / nprof
    batch.set_profiling(False)

    result = None
    if rank == 0:
        # correctness of what was timed: image 0 against the CPU oracle (outside the timed region)
        from oracle import oracle
        ref = oracle.decode(files[0])
        got = outs[0][:, : a.width * 3].reshape(a.height, a.width, 3).cpu().numpy()
        verified = bool(np.array_equal(got, ref.rgb) and np.array_equal(batch.coefficients(0), ref.qcoef))

        S, Cb, P, px = bc["scan"], bc["coef"], bc["rgb"], bc["pixels"]
        alg = {"k0_unstuff": S, "k1a_sync": S, "k1b_fix": 0, "k1s_scan": 0, "k1p_positions": S, "k1d_dcscan": 0,
               "k1w_write": S + Cb, "k2_idct_rgb": Cb + P}
        dom = max((k for k in alg if k in ktimes), key=lambda k: ktimes[k])
        achieved = alg[dom] / (ktimes[dom] * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:  # PMC-derived HBM bytes per launch of the dominant kernel, valid for the profiled batch size only
                t = json.load(open(tpath))
                if t.get("images") == a.images and a.sampling == "420" and a.content == 0:
                    traffic = t["kernels"].get(dom)
            except Exception:
                traffic = None
        ms_per_step = wall / a.steps * 1e3
        value = px * world / (wall / a.steps) / 1e6
        e2e = (S + 2 * Cb + P) / (dev_ms / a.steps * 1e-3) / 1e9
        result = {
            "metric": "decoded megapixels/sec (whole node), 1080p 4:2:0 q90 batch; % HBM roofline",
            "value": round(value, 1), "unit": "MP/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/i16 entropy decode, f32 IDCT+colour", "data": "synthetic",
            "config": {"workload": f"synthetic {a.width}x{a.height} {a.sampling} q{a.quality} baseline JPEG batch, "
                                   f"{a.images} images/GPU ({ndist} distinct seeds cycled), content class {a.content}",
                       "images_per_gpu": a.images, "subseq_bits": a.subseq_bits or "auto", "timing": "device-resident"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg[dom], "kernel_ms": round(ktimes[dom], 4)},
            "roofline_end_to_end": {"achieved": round(e2e, 1), "unit": "GB/s", "frac": round(e2e / HBM_PEAK_GBS, 4),
                                    "algorithmic_bytes_per_step": S + 2 * Cb + P, "device_ms_per_step": round(dev_ms / a.steps, 3)},
            "kernel_ms": {k: round(v, 4) for k, v in ktimes.items()},
            "verified_vs_oracle": verified,
        }
        if world == 1 and not a.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(files, a, threads)
    batch.close()
    ctx.close()
    if rank == 0 and world == 1 and a.h2h_sample > 0 and not a.no_cpu_baseline:
        del out_all, outs
        torch.cuda.empty_cache()
        result["host_to_host"] = host_to_host(parsed, a, dev, stride)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


def host_to_host(parsed, a, dev, stride):
    """PCIe-inclusive rate, reported beside the device-resident figure and never as `value`: parsed images in
    host memory -> plan + upload (zj_batch_create) -> decode -> RGB copied to pinned host memory."""
    import torch
    import jpeg_decoder_amd as J
    n = min(a.h2h_sample, a.images)
    images = [parsed[i % len(parsed)] for i in range(n)]
    stream = torch.cuda.Stream(device=dev)
    ctx = J.Context(dev, stream=stream.cuda_stream)
    host = torch.empty((n, a.height, stride), dtype=torch.uint8).pin_memory()
    out = torch.empty((n, a.height, stride), dtype=torch.uint8, device=f"cuda:{dev}")
    outs = [out[i] for i in range(n)]
    best = None
    for _ in range(2):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        b = J.Batch(ctx, images, a.subseq_bits)
        b.decode(outs)
        st = b.finish()
        with torch.cuda.stream(stream):
            host.copy_(out, non_blocking=True)
        stream.synchronize()
        dt = time.perf_counter() - t0
        b.close()
        assert not st.any()
        best = dt if best is None else min(best, dt)
    ctx.close()
    return {"value": round(a.width * a.height * n / best / 1e6, 1), "unit": "MP/s", "images": n,
            "includes": "batch planning, scan upload, decode, RGB download to pinned host memory (no overlap between them)"}


def cpu_baseline(files, a, threads):
    """The CPU restatement of the reference algorithm (oracle/zj_oracle.c: byte-at-a-time refill,
    linear code search, 384 B/MCU intermediate, naive f32 IDCT) on a bounded sample of the same
    workload, one image per thread over `threads` host threads, plus a single-thread figure."""
    from oracle import oracle
    sample = [files[i % len(files)] for i in range(a.cpu_sample)]
    t0 = time.perf_counter()
    oracle.decode(sample[0])
    t_one = time.perf_counter() - t0
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(lambda f: oracle.decode(f, want_coef=True).status, sample))
    t_all = time.perf_counter() - t0
    px = a.width * a.height
    return {"value": round(px * len(sample) / t_all / 1e6, 2), "unit": "MP/s", "cores": threads, "kind": "port",
            "sample": f"{len(sample)} of the benchmark's {a.width}x{a.height} images, one image per thread; "
                      "reference-algorithm CPU restatement (oracle/zj_oracle.c), not the Rust crate itself",
            "single_thread_value": round(px / t_one / 1e6, 2)}


if __name__ == "__main__":
    main()
