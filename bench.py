#!/usr/bin/env python3
"""Benchmark of the decode hot path, one process per GPU.

    python bench.py --gpus N --steps K --warmup W [--config c3|c4|c5]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

With --gpus N > 1 and no WORLD_SIZE in the environment the script itself starts N fresh child processes (before
anything touches a GPU) and prints rank 0's line.  Workloads (the configurations the headline metric names):

  c3 (default)  4096 synthetic 1920x1080 4:2:0 q90 baseline JPEGs per GPU (64 distinct seeds, each uploaded to its
                own HBM location).  Batches shard by image: no collective, weak scaling.
  c2            single-image latency: tests/golden/large.jpg (7680x4320 4:4:4, no restart markers: one self-synchronising
                stream) as the timed workload, porsche.jpg beside it.  Replicas only: an image without restart markers
                does not shard.
  c4            a mixed job of 1920x1080 / 3840x2160 x 4:4:4 / 4:2:0 images (index bits pick the class), 8192 per
                GPU by default, partitioned over the ranks by scan bytes (longest first), decoded in sub-batches
                that share one workspace (coefficient arena reused); all plans and pixels stay resident.
  c5            one 16384x16384 4:2:0 image with a restart interval of 16 MCUs, cut into one band of MCU rows per
                rank; the bands are gathered on rank 0 (RCCL point-to-point).  Strong scaling.

  c6            tests/golden/large.jpg, which has no restart markers, in one band of MCU rows per rank: every rank
                repeats the entropy front over the whole stream (nothing says where a row begins until everything before
                it has been walked) and produces the coefficients and pixels of its band; gather as in c5.

A step = one decode of the rank's whole share (all stage-1 kernels + the IDCT/colour kernel) with the scans already
resident in HBM and the pixels left in HBM.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, spec
METRIC = "decoded megapixels/sec (whole node), 1080p 4:2:0 q90 batch; % HBM roofline"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c3", choices=["c2", "c3", "c4", "c5", "c6"])
    ap.add_argument("--images", type=int, default=0, help="images per GPU (weak scaling); 0 = 4096 for c3, 8192 for c4")
    ap.add_argument("--sub-batch", type=int, default=1024, help="c4: images per sub-batch of the shared workspace")
    ap.add_argument("--distinct", type=int, default=64, help="distinct synthetic images (seeds), cycled")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--sampling", default="420", choices=["420", "444", "422", "440"])
    ap.add_argument("--no-overlap", action="store_true", help="c4: one workspace and one stream (no overlap of a sub-batch's pixel pass with the next one's entropy passes)")
    ap.add_argument("--quality", type=int, default=90)
    ap.add_argument("--content", type=int, default=0, help="0 natural, 1 smooth, 2 white noise")
    ap.add_argument("--subseq-bits", type=int, default=0)
    ap.add_argument("--c5-size", type=int, default=16384, help="c5: width and height of the image")
    ap.add_argument("--lead-bytes", type=int, default=16384, help="c6 on several ranks: bytes every rank walks in front of its part of the stream")
    ap.add_argument("--c6-rows", action="store_true", help="c6 on several ranks: repeat the entropy front on every rank and split by rows only (the fallback of the split stream)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sensitivity", action="store_true", help="c3: skip the smooth and white-noise content points")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --one-device: rehearsal of the multi-rank path on a box with a single GPU")
    ap.add_argument("--one-device", action="store_true", help="every rank uses GPU 0 (rehearsal only)")
    ap.add_argument("--cpu-sample", type=int, default=48, help="images decoded by the CPU baseline")
    ap.add_argument("--h2h-sample", type=int, default=-1,
                    help="images of the host-to-host (PCIe-inclusive) leg; -1 = 1024 for the default c3 run on one GPU, 0 = skip")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------ launching N ranks
def launch_children(a):
    """N fresh processes, one per GPU, started before this process has touched a GPU; rank 0's stdout is passed on."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    deadline = time.time() + 120
    for p in procs[1:]:
        try:
            p.wait(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
        rc = rc or p.returncode
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return rc


class Dist:
    """Rank bookkeeping and the barrier / max-over-ranks the timing protocol needs."""

    def __init__(self, a):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dev = local_rank if (self.world > 1 and not a.one_device) else 0
        self.backend = a.backend
        torch.cuda.set_device(self.dev)
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if a.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device(f"cuda:{self.dev}"))
            else:
                dist.init_process_group("gloo")

    def sync_all(self):
        self.torch.cuda.synchronize(self.dev)
        if self.world > 1:
            self.dist.barrier()
            self.torch.cuda.synchronize(self.dev)

    def max_over_ranks(self, v):
        if self.world == 1:
            return v
        t = self.torch.tensor([v], dtype=self.torch.float64, device=f"cuda:{self.dev}" if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, v):
        if self.world == 1:
            return v
        t = self.torch.tensor([v], dtype=self.torch.float64, device=f"cuda:{self.dev}" if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self.world > 1:
            self.dist.barrier()
            self.dist.destroy_process_group()


def host_threads():
    """Host threads of this process's CPU share: the box exports it as OMP_NUM_THREADS (os.cpu_count() shows the
    whole machine there)."""
    try:
        return max(1, min(int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1), os.cpu_count() or 1))
    except ValueError:
        return os.cpu_count() or 1


STEP_MS = {}  # this rank's device milliseconds per step of the last timed region (median, min, max), from HIP events between the steps


def timed_steps(D, stream, step, steps, warmup, check):
    """W untimed steps, then exactly K steps between barrier + synchronize on both sides; returns (max-over-ranks
    wall seconds, this rank's device milliseconds from HIP events on the launch stream) and leaves the per-step
    device times in STEP_MS."""
    torch = D.torch
    for _ in range(warmup):
        step()
    check()
    D.sync_all()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    evs[0].record(stream)
    for k in range(steps):
        step()
        evs[k + 1].record(stream)
    stream.synchronize()
    D.sync_all()
    wall = time.perf_counter() - t0
    dev_ms = evs[0].elapsed_time(evs[-1])
    per_step = sorted(evs[k].elapsed_time(evs[k + 1]) for k in range(steps))
    STEP_MS.clear()
    STEP_MS.update({"median": round(per_step[len(per_step) // 2], 4), "min": round(per_step[0], 4), "max": round(per_step[-1], 4)})
    return D.max_over_ranks(wall), dev_ms


def profile_kernels(batches, outs_of, nprof=3):
    """Per-kernel durations from HIP events on the launch stream around every kernel, summed over the batches of
    one step and averaged over `nprof` steps."""
    kt = {}
    for b in batches:
        b.set_profiling(True)
    for _ in range(nprof):
        for b in batches:
            b.decode(outs_of(b))
            for k, v in b.kernel_times().items():
                kt[k] = kt.get(k, 0.0) + v / nprof
    for b in batches:
        b.set_profiling(False)
    return kt


def kernels_sha256():
    """Hash of the kernel sources of this tree (the same files tools/pmc_traffic.py hashes when it records counters)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("zj_kernels.hip", "zj_core.h", "zj_dev.h"):
        h.update(open(os.path.join(ROOT, "jpeg-decoder_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def roofline_fields(kt, S, Cb, P, traffic_key=None):
    """Roofline object of the dominant kernel.  `achieved` is measured in this run (HIP events).  `traffic` is NOT: it
    is the PMC-derived HBM bytes per launch from the committed table profiles/traffic.json, attached only when that
    table was taken on exactly these kernel sources (hash) and holds this workload; `traffic_source` says so."""
    alg = {"k0_unstuff": S, "k1a_sync": S, "k1b_fix": 0, "k1s_scan": 0, "k1p_positions": S, "k1d_dcscan": 0,
           "k1w_write": S + Cb, "k2_idct_rgb": Cb + P}
    dom = max((k for k in alg if k in kt), key=lambda k: kt[k])
    achieved = alg[dom] / (kt[dom] * 1e-3) / 1e9
    traffic, source = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if traffic_key and os.path.exists(tpath):
        try:
            t = json.load(open(tpath))
            if t.get("kernels_sha256") != kernels_sha256():
                source = "profiles/traffic.json was taken on other kernel sources than this tree's: not attached"
            elif traffic_key not in t.get("workloads", {}):
                source = f"profiles/traffic.json holds no counters for workload {traffic_key}"
            else:
                traffic = t["workloads"][traffic_key].get(dom)
                source = (f"profiles/traffic.json (committed; rocprofv3 PMC run of commit {t.get('commit')}, kernel sources "
                          f"sha256 {t['kernels_sha256'][:12]} = this tree's), workload {traffic_key}; not measured in this run")
        except Exception as e:
            traffic, source = None, f"profiles/traffic.json unreadable: {e}"
    return {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": source,
            "algorithmic_bytes_per_launch": alg[dom], "kernel_ms": round(kt[dom], 4)}


def verify_against_oracle(files, fetch, threads):
    """Every distinct file through the CPU oracle (outside the timed region), pixels and coefficients compared with
    what `fetch(k)` returns for file k: (rgb HxWx3 uint8 numpy, qcoef or None)."""
    import numpy as np
    from oracle import oracle
    with ThreadPoolExecutor(threads) as ex:
        refs = list(ex.map(lambda f: oracle.decode(f), files))
    bad = []
    for k, ref in enumerate(refs):
        rgb, q = fetch(k)
        if not np.array_equal(rgb, ref.rgb) or (q is not None and not np.array_equal(q, ref.qcoef)):
            bad.append(k)
    return len(refs), bad


def fast_path_fields(batches):
    c = [0, 0, 0]
    for b in batches:
        for i, v in enumerate(b.counters()):
            c[i] += v
    return {"fix_ran_through_a_chunk": c[0], "blocks_off_their_position": c[1], "finish_calls_on_the_slow_path": c[2]}


# ------------------------------------------------------------------------------------------ c3
def content_point(a, ctx, stream, D, out_all, hs, vs, content, threads, nimg=1024, ndist=8):
    """The c3 measurement on 1 024 images of another content class, decoded into the first 1 024 output buffers;
    all distinct images are checked against the oracle."""
    import jpeg_decoder_amd as J
    from jpeg_decoder_amd import synth
    with ThreadPoolExecutor(threads) as ex:
        files = list(ex.map(lambda s: synth.encode(a.width, a.height, hs, vs, a.quality, content, 500 + s, 0), range(ndist)))
    parsed = [J.ParsedImage(f) for f in files]
    b = J.Batch(ctx, [parsed[i % ndist] for i in range(nimg)], a.subseq_bits)
    tables = b.output_tables([out_all[i] for i in range(nimg)])

    def check():
        assert not b.finish().any()
    wall, dev_ms = timed_steps(D, stream, lambda: b.decode(tables), 5, 2, check)
    check()
    fast = fast_path_fields([b])
    nver, bad = verify_against_oracle(files, lambda k: (out_all[k][:, : a.width * 3].reshape(a.height, a.width, 3).cpu().numpy(),
                                                        b.coefficients(k)), threads)
    assert not bad and not any(fast.values()), f"content class {content}: mismatches {bad}, fast path {fast}"
    bc = b.byte_counts()
    b.close()
    return {"value": round(bc["pixels"] / (dev_ms / 5 * 1e-3) / 1e6, 1), "unit": "MP/s", "images": nimg,
            "scan_bytes_per_pixel": round(bc["scan"] / bc["pixels"], 4), "ms_per_step": round(dev_ms / 5, 3),
            "roofline_end_to_end_frac": round((bc["scan"] + 2 * bc["coef"] + bc["rgb"]) / (dev_ms / 5 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "distinct_images_checked": nver}


def run_c3(a, D):
    import numpy as np
    import jpeg_decoder_amd as J
    from jpeg_decoder_amd import synth
    torch, dev = D.torch, D.dev
    nimg = a.images or 4096
    hs, vs = {"420": (2, 2), "444": (1, 1), "422": (2, 1), "440": (1, 2)}[a.sampling]
    ndist = max(1, min(a.distinct, nimg))
    threads = host_threads()
    with ThreadPoolExecutor(threads) as ex:
        files = list(ex.map(lambda s: synth.encode(a.width, a.height, hs, vs, a.quality, a.content, s, 0), range(ndist)))
    parsed = [J.ParsedImage(f) for f in files]
    images = [parsed[i % ndist] for i in range(nimg)]

    stream = torch.cuda.Stream(device=dev)
    ctx = J.Context(dev, stream=stream.cuda_stream)
    batch = J.Batch(ctx, images, a.subseq_bits)
    bc = batch.byte_counts()
    stride = (a.width * 3 + 15) // 16 * 16
    out_all = torch.empty((nimg, a.height, stride), dtype=torch.uint8, device=f"cuda:{dev}")
    outs = [out_all[i] for i in range(nimg)]

    def check():
        st = batch.finish()
        assert not st.any(), f"decode reported status {np.unique(st)}"

    tables = batch.output_tables(outs)  # the tensors are checked once, not on every step
    wall, dev_ms = timed_steps(D, stream, lambda: batch.decode(tables), a.steps, a.warmup, check)
    check()
    fast = fast_path_fields([batch])
    assert not any(fast.values()), f"a timed step left the fast path: {fast}"
    kt = profile_kernels([batch], lambda b: tables)

    result = None
    if D.rank == 0:
        # correctness of what was timed: every distinct image against the CPU oracle, and every other image against
        # the first copy of its source
        def fetch(k):
            return out_all[k][:, : a.width * 3].reshape(a.height, a.width, 3).cpu().numpy(), batch.coefficients(k)
        nver, bad = verify_against_oracle(files, fetch, threads)
        copies_equal = all(torch.equal(out_all[g: g + ndist][: min(ndist, nimg - g)], out_all[: min(ndist, nimg - g)])
                           for g in range(ndist, nimg, ndist))
        S, Cb, P, px = bc["scan"], bc["coef"], bc["rgb"], bc["pixels"]
        e2e = (S + 2 * Cb + P) / (dev_ms / a.steps * 1e-3) / 1e9
        result = {
            "metric": METRIC, "value": round(px * D.world / (wall / a.steps) / 1e6, 1), "unit": "MP/s", "n_gpus": D.world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(wall / a.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8/i16 entropy decode, f32 IDCT+colour", "data": "synthetic",
            "config": {"workload": f"c3: synthetic {a.width}x{a.height} {a.sampling} q{a.quality} baseline JPEG batch, "
                                   f"{nimg} images/GPU ({ndist} distinct seeds cycled), content class {a.content}",
                       "images_per_gpu": nimg, "subseq_bits": a.subseq_bits or "auto", "timing": "device-resident",
                       "sharding": "by image, no collective"},
            "roofline": roofline_fields(kt, S, Cb, P, f"c3:{nimg}" if a.sampling == "420" and a.content == 0 else None),
            "roofline_end_to_end": {"achieved": round(e2e, 1), "unit": "GB/s", "frac": round(e2e / HBM_PEAK_GBS, 4),
                                    "algorithmic_bytes_per_step": S + 2 * Cb + P, "device_ms_per_step": round(dev_ms / a.steps, 3)},
            "step_device_ms": dict(STEP_MS),
            "kernel_ms": {k: round(v, 4) for k, v in kt.items()},
            "verified_vs_oracle": {"distinct_images_checked": nver, "mismatches": len(bad), "copies_equal_their_source": bool(copies_equal)},
            "fast_path": fast,
        }
        assert not bad and copies_equal, f"GPU output differs from the oracle for distinct images {bad}"
        if D.world == 1 and not a.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(files, a.width * a.height, a.cpu_sample, threads,
                                                  f"of the benchmark's {a.width}x{a.height} images")
    batch.close()
    if D.rank == 0 and D.world == 1 and nimg >= 1024 and a.content == 0 and not a.no_sensitivity:
        # the two other content classes as sensitivity points: a smooth image (few bits per block) and white noise
        result["content_sensitivity"] = {name: content_point(a, ctx, stream, D, out_all, hs, vs, c, threads)
                                         for name, c in (("smooth", 1), ("white_noise", 2))}
    h2h = a.h2h_sample if a.h2h_sample >= 0 else (1024 if (D.world == 1 and nimg >= 1024) else 0)
    if D.rank == 0 and D.world == 1 and h2h > 0:
        del out_all, outs, tables
        torch.cuda.empty_cache()
        result["host_to_host"] = host_to_host(ctx, stream, parsed, a, dev, stride, min(h2h, nimg))
    ctx.close()
    return result


def host_to_host(ctx, stream, parsed, a, dev, stride, n):
    """PCIe-inclusive rate, reported beside the device-resident figure and never as `value`: parsed images in host
    memory -> plan + upload (pinned staging, copy stream) -> decode -> RGB to pinned host memory, in sub-batches of
    256 images that share one workspace, so that the upload of sub-batch k+1 and the download of sub-batch k-1
    overlap the decode of sub-batch k and nothing is allocated between sub-batches."""
    import torch
    import jpeg_decoder_amd as J
    sub = 256
    groups = [[parsed[i % len(parsed)] for i in range(g, min(g + sub, n))] for g in range(0, n, sub)]
    plan, scratch = J.Workspace.estimate(groups[0])
    ws = J.Workspace(ctx, scratch, plan, 2)
    host = torch.empty((n, a.height, stride), dtype=torch.uint8).pin_memory()
    dbuf = [torch.empty((sub, a.height, stride), dtype=torch.uint8, device=f"cuda:{dev}") for _ in range(2)]
    d2h = torch.cuda.Stream(device=dev)
    best = None
    for _ in range(3):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        done = [None, None]  # per output buffer: event behind the download that reads it
        b = J.Batch(ctx, groups[0], a.subseq_bits, workspace=ws, slot=0)
        b.decode([dbuf[0][i] for i in range(len(groups[0]))])
        for k, g in enumerate(groups):
            nxt = None
            if k + 1 < len(groups):  # planned and uploaded while sub-batch k decodes
                nxt = J.Batch(ctx, groups[k + 1], a.subseq_bits, workspace=ws, slot=(k + 1) % 2)
            st = b.finish()
            assert not st.any()
            d2h.wait_stream(stream)
            with torch.cuda.stream(d2h):
                host[k * sub: k * sub + len(g)].copy_(dbuf[k % 2][: len(g)], non_blocking=True)
                done[k % 2] = torch.cuda.Event()
                done[k % 2].record(d2h)
            b.close()
            if nxt is not None:
                if done[(k + 1) % 2] is not None:
                    done[(k + 1) % 2].synchronize()
                nxt.decode([dbuf[(k + 1) % 2][i] for i in range(len(groups[k + 1]))])
            b = nxt
        d2h.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    ws.close()
    return {"value": round(a.width * a.height * n / best / 1e6, 1), "unit": "MP/s", "images": n, "sub_batch": sub,
            "includes": "batch planning, scan upload through pinned staging, decode, RGB download to pinned host memory; "
                        "sub-batches share one workspace and overlap (no allocation between them)",
            "pcie_note": "3 output bytes per pixel cross PCIe, which bounds this figure near 18 000 MP/s whatever the decode rate"}


def cpu_baseline(files, pixels, nsample, threads, what):
    """The CPU restatement of the reference algorithm (oracle/zj_oracle.c, built -O3 -ffp-contract=off: byte-at-a-time
    refill, linear code search, 384 B/MCU intermediate, naive f32 IDCT) on a bounded sample of the same workload:
    one image per thread over the host threads of this process's CPU share, median of three passes after a warm-up
    pass, plus a single-thread figure (the reference is single-threaded), median of three after one warm-up."""
    from oracle import oracle
    if not isinstance(pixels, (list, tuple)):
        pixels = [pixels] * len(files)
    sample = [files[i % len(files)] for i in range(nsample)]
    px_total = sum(pixels[i % len(files)] for i in range(nsample))
    oracle.decode(sample[0])  # warm-up: library load, page-in
    one = []
    for _ in range(3):
        t0 = time.perf_counter()
        oracle.decode(sample[0])
        one.append(time.perf_counter() - t0)
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(lambda f: oracle.decode(f, want_coef=True).status, sample[:threads]))  # warm-up pass
        alls = []
        for _ in range(3):
            t0 = time.perf_counter()
            list(ex.map(lambda f: oracle.decode(f, want_coef=True).status, sample))
            alls.append(time.perf_counter() - t0)
    t_one, t_all = sorted(one)[1], sorted(alls)[1]
    return {"value": round(px_total / t_all / 1e6, 2), "unit": "MP/s", "cores": threads, "kind": "port",
            "sample": f"{len(sample)} {what}, one image per thread over {threads} threads, median of 3 passes after a warm-up; "
                      "reference-algorithm CPU restatement (oracle/zj_oracle.c, -O3), not the Rust crate itself",
            "single_thread_value": round(pixels[0] / t_one / 1e6, 2)}


# ------------------------------------------------------------------------------------------ c2
def run_c2(a, D):
    import numpy as np
    import jpeg_decoder_amd as J
    torch, dev = D.torch, D.dev
    stream = torch.cuda.Stream(device=dev)
    ctx = J.Context(dev, stream=stream.cuda_stream)
    golden = os.path.join(ROOT, "tests", "golden")
    info = json.load(open(os.path.join(golden, "golden.json")))["fixtures"]
    res, main = {}, None
    for name in ("porsche.jpg", "large.jpg"):
        data = open(os.path.join(golden, name), "rb").read()
        p = J.ParsedImage(data)
        b = J.Batch(ctx, [p], a.subseq_bits)
        outs = b.alloc_rgb()

        def check():
            assert not b.finish().any()
        wall, dev_ms = timed_steps(D, stream, lambda: b.decode(outs), a.steps, a.warmup, check)
        check()
        kt = profile_kernels([b], lambda _b: outs)
        bc = b.byte_counts()
        crc = zlib.crc32(np.ascontiguousarray(outs[0][:, : p.width * 3].cpu().numpy()).tobytes())
        ok = crc == info[name]["rgb_crc"] and zlib.crc32(b.coefficients(0).tobytes()) == info[name]["qcoef_crc"]
        res[name] = {"pixels": p.width * p.height, "ms": round(dev_ms / a.steps, 4), "wall_ms": round(wall / a.steps * 1e3, 4),
                     "step_device_ms": dict(STEP_MS), "kernel_ms": {k: round(v, 4) for k, v in kt.items()}, "equals_committed_oracle_hashes": bool(ok),
                     "fast_path": fast_path_fields([b]),
                     "roofline_end_to_end_frac": round((bc["scan"] + 2 * bc["coef"] + bc["rgb"]) / (dev_ms / a.steps * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        if name == "large.jpg":
            main = (wall, dev_ms, kt, bc, p)
        b.close()
    ctx.close()
    if D.rank != 0:
        return None
    wall, dev_ms, kt, bc, p = main
    assert all(r["equals_committed_oracle_hashes"] for r in res.values()), res
    return {"metric": METRIC, "value": round(p.width * p.height * D.world / (wall / a.steps) / 1e6, 1), "unit": "MP/s", "n_gpus": D.world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(wall / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/i16 entropy decode, f32 IDCT+colour", "data": "fixture",
            "config": {"workload": "c2: single-image latency, tests/golden/large.jpg (7680x4320 4:4:4, one stream without restart markers); "
                                   "porsche.jpg beside it", "timing": "device-resident", "sharding": "replicas only"},
            "roofline": roofline_fields(kt, bc["scan"], bc["coef"], bc["rgb"], "c2"), "images": res}


# ------------------------------------------------------------------------------------------ c4
C4_CLASSES = [(1920, 1080, 2, 2), (3840, 2160, 2, 2), (1920, 1080, 1, 1), (3840, 2160, 1, 1)]  # index bit 0: size, bit 1: sampling


def run_c4(a, D):
    import numpy as np
    import jpeg_decoder_amd as J
    from jpeg_decoder_amd import shard, synth
    torch, dev = D.torch, D.dev
    per_gpu = a.images or 8192
    total = per_gpu * D.world
    nseed = max(1, min(a.distinct // 4, 16))
    threads = host_threads()
    keys = [(c, s) for c in range(4) for s in range(nseed)]
    with ThreadPoolExecutor(threads) as ex:
        blobs = list(ex.map(lambda k: synth.encode(C4_CLASSES[k[0]][0], C4_CLASSES[k[0]][1], C4_CLASSES[k[0]][2], C4_CLASSES[k[0]][3],
                                                   a.quality, a.content, 1000 * k[0] + k[1], 0), keys))
    files = dict(zip(keys, blobs))
    parsed = {k: J.ParsedImage(f) for k, f in files.items()}
    key_of = [(i & 3, (i >> 2) % nseed) for i in range(total)]  # the job: image i is of class i & 3
    # static partition of the whole job over the ranks by scan bytes, longest first; no data-path collective
    costs = [len(files[k]) for k in key_of]
    mine = shard.partition_batch(costs, D.world)[D.rank]
    my_keys = [key_of[i] for i in mine]

    stream = torch.cuda.Stream(device=dev)
    ctx = J.Context(dev, stream=stream.cuda_stream)
    sub = max(1, a.sub_batch)
    groups = [my_keys[g: g + sub] for g in range(0, len(my_keys), sub)]
    est = [J.Workspace.estimate([parsed[k] for k in g]) for g in groups]
    # Sub-batches alternate between two workspaces (two scratch ranges), and the IDCT/colour pass runs on a stream
    # of its own: the pixel pass of one sub-batch overlaps the entropy passes of the next, which hides the tails
    # of the kernels of either.  One workspace and one stream with --no-overlap or a single sub-batch.
    overlap = len(groups) > 1 and not a.no_overlap
    pix = torch.cuda.Stream(device=dev) if overlap else None
    nws = 2 if overlap else 1
    wss = [J.Workspace(ctx, max(e[1] for e in est), max(e[0] for e in est), (len(groups) + nws - 1) // nws) for _ in range(nws)]
    batches, outs, tables = [], {}, {}
    S = Cb = P = px = 0
    for gi, g in enumerate(groups):
        b = J.Batch(ctx, [parsed[k] for k in g], a.subseq_bits, workspace=wss[gi % nws], slot=gi // nws)
        batches.append(b)
        outs[id(b)] = b.alloc_rgb()
        tables[id(b)] = b.output_tables(outs[id(b)])
        bc = b.byte_counts()
        S, Cb, P, px = S + bc["scan"], Cb + bc["coef"], P + bc["rgb"], px + bc["pixels"]

    def step():
        if not overlap:
            for b in batches:
                b.decode(tables[id(b)])
            return
        for b in batches:
            b.decode_entropy()
            b.decode_pixels(tables[id(b)], pix.cuda_stream)
        stream.wait_stream(pix)  # the step is over when its last pixels are out

    def check():
        for b in batches:
            st = b.finish()
            assert not st.any(), f"decode reported status {np.unique(st)}"

    wall, dev_ms = timed_steps(D, stream, step, a.steps, a.warmup, check)
    check()
    fast = fast_path_fields(batches)
    assert not any(fast.values()), f"a timed step left the fast path: {fast}"
    kt = profile_kernels(batches, lambda b: tables[id(b)])
    px_all = D.sum_over_ranks(float(px))
    bytes_all = D.sum_over_ranks(float(S + 2 * Cb + P))
    # every image of this rank against the first decoded copy of its source, and those first copies against the oracle
    first, copies_equal = {}, True
    for b, g in zip(batches, groups):
        for t, k in zip(outs[id(b)], g):
            if k not in first:
                first[k] = t
            elif copies_equal and not torch.equal(t[:, : parsed[k].width * 3], first[k][:, : parsed[k].width * 3]):
                copies_equal = False
    ks = sorted(first)
    nver, bad = verify_against_oracle([files[k] for k in ks],
                                      lambda j: (first[ks[j]][:, : parsed[ks[j]].width * 3].reshape(parsed[ks[j]].height, parsed[ks[j]].width, 3).cpu().numpy(), None),
                                      threads)
    ok = D.sum_over_ranks(0.0 if (not bad and copies_equal) else 1.0) == 0.0
    result = None
    if D.rank == 0:
        e2e = (S + 2 * Cb + P) / (dev_ms / a.steps * 1e-3) / 1e9
        result = {
            "metric": METRIC, "value": round(px_all / (wall / a.steps) / 1e6, 1), "unit": "MP/s", "n_gpus": D.world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(wall / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/i16 entropy decode, f32 IDCT+colour", "data": "synthetic",
            "config": {"workload": f"c4: mixed job of {total} images ({per_gpu} per GPU), 1920x1080 / 3840x2160 x 4:2:0 / 4:4:4 in equal "
                                   f"parts, q{a.quality}, {4 * nseed} distinct files cycled",
                       "images_per_gpu": len(my_keys), "sub_batches": len(groups), "sub_batch_images": sub, "pixel_pass_on_its_own_stream": overlap, "timing": "device-resident",
                       "sharding": "by image, longest-processing-time-first on scan bytes, no collective",
                       "workspace_bytes": {"scratch_shared": max(e[1] for e in est), "plan_per_sub_batch": max(e[0] for e in est)}},
            "roofline": roofline_fields(kt, S, Cb, P, f"c4:{per_gpu}/{sub}" if D.world == 1 else None),
            "roofline_end_to_end": {"achieved": round(e2e, 1), "unit": "GB/s", "frac": round(e2e / HBM_PEAK_GBS, 4),
                                    "algorithmic_bytes_per_step_all_ranks": bytes_all, "device_ms_per_step": round(dev_ms / a.steps, 3)},
            "step_device_ms": dict(STEP_MS), "kernel_ms": {k: round(v, 4) for k, v in kt.items()},
            "verified_vs_oracle": {"distinct_images_checked_on_rank0": nver, "mismatches_on_rank0": len(bad), "all_ranks_ok": bool(ok)},
            "fast_path": fast,
        }
        if D.world == 1 and not a.no_cpu_baseline:
            smp = [files[(c, s_ % nseed)] for s_ in range(8) for c in range(4)]
            pxs = [C4_CLASSES[c][0] * C4_CLASSES[c][1] for s_ in range(8) for c in range(4)]
            result["cpu_baseline"] = cpu_baseline(smp, pxs, len(smp), threads, "images, eight of each class of the job")
    assert ok, "GPU output differs from the oracle"
    for b in batches:
        b.close()
    for ws in wss:
        ws.close()
    ctx.close()
    return result


# ------------------------------------------------------------------------------------------ c5
def run_c5(a, D, restart_free=False):
    import numpy as np
    import jpeg_decoder_amd as J
    from jpeg_decoder_amd import shard, synth
    torch, dev = D.torch, D.dev
    if restart_free:
        data = open(os.path.join(ROOT, "tests", "golden", "large.jpg"), "rb").read()
        p = J.ParsedImage(data)
        mcux, mcuy, _, mh = shard.mcu_geometry(p.width, p.height, p.image.hsamp[0], p.image.vsamp[0])
        per = (mcuy + D.world - 1) // D.world
        bands = [(min(r * per, mcuy), max(0, min(per, mcuy - r * per))) for r in range(D.world)]
    else:
        size = a.c5_size
        strip_h = min(1024, size)
        reps = size // strip_h
        data = synth.encode_tiled(size, strip_h, reps, 2, 2, a.quality, a.content, 77, 16)
        p = J.ParsedImage(data)
        mcux, mcuy, _, mh = shard.mcu_geometry(p.width, p.height, 2, 2)
        bands = shard.band_rows(mcux, mcuy, 16, D.world)
    row0, nrows = bands[D.rank]
    stream = torch.cuda.Stream(device=dev)
    ctx = J.Context(dev, stream=stream.cuda_stream)
    row_bytes = (p.width * 3 + 15) // 16 * 16
    shapes = [(min(n * mh, p.height - r * mh) if n else 0, row_bytes) for r, n in bands]
    # the root decodes straight into its rows of the assembled image and receives the other bands into theirs
    full = torch.empty((p.height, row_bytes), dtype=torch.uint8, device=f"cuda:{dev}") if D.rank == 0 else None
    mine = full[row0 * mh: row0 * mh + shapes[D.rank][0]] if D.rank == 0 else torch.empty(shapes[D.rank], dtype=torch.uint8, device=f"cuda:{dev}")
    if restart_free:  # the whole stream on every rank, a window of rows out
        batch = J.Batch(ctx, [p], a.subseq_bits, rows=[(row0, nrows)] if D.world > 1 else None)
    else:
        batch = J.Batch(ctx, [p.band(row0, nrows) if D.world > 1 else p], a.subseq_bits)
    bc = batch.byte_counts()
    recv = None
    if D.rank == 0 and D.world > 1:
        recv = [full[r * mh: r * mh + shapes[k][0]] for k, (r, n) in enumerate(bands)]
    use_cpu = D.backend == "gloo"

    def step():
        with torch.cuda.stream(stream):
            batch.decode([mine])
            if D.world > 1:
                gather_into(D, mine, recv, shapes, use_cpu)

    def check():
        st = batch.finish()
        assert not st.any(), f"decode reported status {np.unique(st)}"

    wall, dev_ms = timed_steps(D, stream, step, a.steps, a.warmup, check)
    check()
    fast = fast_path_fields([batch])
    assert not any(fast.values()), f"a timed step left the fast path: {fast}"
    kt = profile_kernels([batch], lambda b: [mine])
    result = None
    if D.rank == 0:
        # the assembled image against a decode of the whole image on this GPU alone, and every repetition of the
        # strip against the CPU oracle's decode of the strip (restart intervals are independent, so the image is the
        # strip's pixels repeated)
        torch.cuda.synchronize(dev)
        crc_gathered = zlib.crc32(full[:, : p.width * 3].cpu().numpy().tobytes())
        if D.world > 1:
            wb = J.Batch(ctx, [p], a.subseq_bits)
            whole = torch.empty((p.height, row_bytes), dtype=torch.uint8, device=f"cuda:{dev}")
            wb.decode([whole])
            assert not wb.finish().any()
            crc_whole = zlib.crc32(whole[:, : p.width * 3].cpu().numpy().tobytes())
            wb.close()
            del whole
        else:
            crc_whole = crc_gathered
        if restart_free:  # the committed hash of the oracle's decode of the fixture
            want = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))["fixtures"]["large.jpg"]["rgb_crc"]
            strips_ok = crc_gathered == want
        else:
            from oracle import oracle
            ref = torch.from_numpy(oracle.decode(synth.encode(size, strip_h, 2, 2, a.quality, a.content, 77, 16)).rgb).to(f"cuda:{dev}")
            strips_ok = all(torch.equal(full[k * strip_h: (k + 1) * strip_h, : p.width * 3].reshape(strip_h, p.width, 3), ref) for k in range(reps))
        S, Cb, P = bc["scan"], bc["coef"], bc["rgb"]  # this rank's band
        px = p.width * p.height
        e2e_band = (S + 2 * Cb + P) / (dev_ms / a.steps * 1e-3) / 1e9
        result = {
            "metric": METRIC, "value": round(px / (wall / a.steps) / 1e6, 1), "unit": "MP/s", "n_gpus": D.world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(wall / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u8/i16 entropy decode, f32 IDCT+colour", "data": "fixture" if restart_free else "synthetic",
            "config": {"workload": ("c6: tests/golden/large.jpg (7680x4320 4:4:4, no restart markers), entropy front repeated on every "
                                    "rank, one band of MCU rows per rank, bands gathered on rank 0") if restart_free else
                                   (f"c5: one {size}x{size} 4:2:0 q{a.quality} image, restart interval 16 MCUs ({len(data)} bytes), "
                                    f"one band of MCU rows per rank, bands gathered on rank 0"),
                       "bands_mcu_rows": bands, "timing": "device-resident, gather included",
                       "collective": "none" if D.world == 1 else f"point-to-point gather to rank 0 ({D.backend})"},
            "roofline": roofline_fields(kt, S, Cb, P, ("c6" if restart_free else "c5") if D.world == 1 else None),
            "roofline_end_to_end": {"achieved": round(e2e_band, 1), "unit": "GB/s", "frac": round(e2e_band / HBM_PEAK_GBS, 4),
                                    "algorithmic_bytes_per_step_rank0_band": S + 2 * Cb + P, "device_ms_per_step": round(dev_ms / a.steps, 3),
                                    "note": "rank 0's band and its stream time, which includes waiting for the gather"},
            "step_device_ms": dict(STEP_MS), "kernel_ms": {k: round(v, 4) for k, v in kt.items()},
            "verified": {"crc32_gathered": crc_gathered, "crc32_whole_image_one_gpu": crc_whole, "equal": crc_gathered == crc_whole,
                         "equals_oracle": bool(strips_ok)},
            "fast_path": fast,
        }
        assert crc_gathered == crc_whole and strips_ok, "gathered image differs"
        if D.world == 1 and not a.no_cpu_baseline and restart_free:
            result["cpu_baseline"] = cpu_baseline([data], p.width * p.height, 3, 1, "decodes of the image itself, one thread")
        elif D.world == 1 and not a.no_cpu_baseline:
            strip = synth.encode(size, strip_h, 2, 2, a.quality, a.content, 77, 16)
            result["cpu_baseline"] = cpu_baseline([strip], size * strip_h, min(host_threads(), reps), host_threads(),
                                                  f"strips of {size}x{strip_h} pixels of the image (restart intervals are independent)")
    batch.close()
    ctx.close()
    return result


# ------------------------------------------------------------------------------------------ c6
def run_c6(a, D):
    """tests/golden/large.jpg, one stream without restart markers.  One rank: the plain decode.  Several ranks: the
    stream is split by bytes (RangeBatch) -- every rank unstuffs and walks its part plus a lead-in, one all-gather of
    32-byte reports joins the parts up, every rank writes the coefficients and pixels of the MCUs that begin in its
    part, rank 0 gathers the pixels.  If the parts do not join up every rank learns so from the same reports and the
    run falls back to the repeated front with row windows (run_c5's restart-free form), and says so."""
    if D.world == 1 or a.c6_rows:
        return run_c5(a, D, True)
    import numpy as np
    import torch
    import jpeg_decoder_amd as J
    from jpeg_decoder_amd import shard
    torch_dist = D.dist
    dev = D.dev
    data = open(os.path.join(ROOT, "tests", "golden", "large.jpg"), "rb").read()
    p = J.ParsedImage(data)
    hs, vs = p.image.hsamp[0], p.image.vsamp[0]
    mcux, mcuy, mw, mh = shard.mcu_geometry(p.width, p.height, hs, vs)
    stream = torch.cuda.Stream(device=dev)
    ctx = J.Context(dev, stream=stream.cuda_stream)
    pitch = (p.width * 3 + 15) // 16 * 16
    part = J.RangeBatch(ctx, p, D.rank, D.world, a.lead_bytes, a.subseq_bits)
    window = torch.empty((p.height, pitch), dtype=torch.uint8, device=f"cuda:{dev}")  # this rank's MCU rows begin at row 0 of it
    full = torch.empty((p.height, pitch), dtype=torch.uint8, device=f"cuda:{dev}") if D.rank == 0 else None
    use_cpu = D.backend == "gloo"
    RB = J.RangeBatch.REPORT_BYTES
    bad_report = (np.array([-1, 0, -2, 0, 0, 0, 0, 0], np.int32)).tobytes()  # joins up with no neighbour
    state = {"ok": True, "wins": None, "ms": {"front": 0.0, "exchange": 0.0, "write_pixels_gather": 0.0}}

    def exchange(rep):
        t = torch.frombuffer(bytearray(rep), dtype=torch.uint8)
        if not use_cpu:
            t = t.to(f"cuda:{dev}")
        out = [torch.empty_like(t) for _ in range(D.world)]
        torch_dist.all_gather(out, t)
        return [bytes(o.cpu().numpy().tobytes()) for o in out]

    def step():
        with torch.cuda.stream(stream):
            t0 = time.perf_counter()
            try:
                rep = part.front()
            except J.ChainError:
                rep = bad_report
            t1 = time.perf_counter()
            reports = exchange(rep)
            try:
                part.resolve(reports)
            except (J.ChainError, J.DecodeError):
                state["ok"] = False  # the same verdict on every rank: all of them check the whole chain
                return
            t2 = time.perf_counter()
            wins = shard.split_windows(reports, mcux * mcuy)
            state["wins"] = wins
            part.decode(window)
            row0 = part.window[2]
            pieces = [shard.mcu_range_pieces(m0, n, mcux, mw, mh, p.width, p.height) for (m0, n) in wins]
            mine = shard.pack_pieces(window, row0 * mh, pieces[D.rank])
            if D.rank == 0:
                shard.unpack_pieces(full, mine, pieces[0])
                bufs = {r: torch.empty(shard.pieces_bytes(pieces[r]), dtype=torch.uint8, device="cpu" if use_cpu else f"cuda:{dev}")
                        for r in range(1, D.world) if shard.pieces_bytes(pieces[r])}
                if use_cpu:
                    stream.synchronize()
                    for q in [torch_dist.irecv(bufs[r], r) for r in bufs]:
                        q.wait()
                elif bufs:
                    for q in torch_dist.batch_isend_irecv([torch_dist.P2POp(torch_dist.irecv, bufs[r], r) for r in bufs]):
                        q.wait()
                for r in bufs:
                    shard.unpack_pieces(full, bufs[r].to(f"cuda:{dev}"), pieces[r])
            elif mine.numel():
                if use_cpu:
                    stream.synchronize()
                    torch_dist.send(mine.cpu(), 0)
                else:
                    for q in torch_dist.batch_isend_irecv([torch_dist.P2POp(torch_dist.isend, mine, 0)]):
                        q.wait()
            stream.synchronize()
            t3 = time.perf_counter()
            for k, v in (("front", t1 - t0), ("exchange", t2 - t1), ("write_pixels_gather", t3 - t2)):
                state["ms"][k] = v * 1e3

    def check():
        if state["ok"]:
            st = part.finish()
            assert not st.any(), f"decode reported status {np.unique(st)}"

    step()
    check()
    if not state["ok"]:
        part.close()
        ctx.close()
        res = run_c5(a, D, True)
        if res is not None:
            res["config"]["workload"] += f"; FELL BACK from the split stream: the parts did not join up with a lead-in of {a.lead_bytes} bytes"
        return res
    wall, dev_ms = timed_steps(D, stream, step, a.steps, a.warmup, check)
    check()
    assert state["ok"]
    lo, hi = part.segments()
    my_bytes = D.torch.tensor([hi - lo], dtype=D.torch.float64)
    result = None
    if D.rank == 0:
        torch.cuda.synchronize(dev)
        crc_gathered = zlib.crc32(full[:, : p.width * 3].cpu().numpy().tobytes())
        wb = J.Batch(ctx, [p], a.subseq_bits)
        whole = torch.empty((p.height, pitch), dtype=torch.uint8, device=f"cuda:{dev}")
        wb.decode([whole])
        assert not wb.finish().any()
        crc_whole = zlib.crc32(whole[:, : p.width * 3].cpu().numpy().tobytes())
        kt = profile_kernels([wb], lambda b: [whole])
        bc = wb.byte_counts()
        wb.close()
        want = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))["fixtures"]["large.jpg"]["rgb_crc"]
        px = p.width * p.height
        rf = roofline_fields(kt, bc["scan"], bc["coef"], bc["rgb"], None)
        rf["note"] = "kernels of the whole image decoded on rank 0 alone, for reference: the split path is timed as phases, not kernels"
        result = {
            "metric": METRIC, "value": round(px / (wall / a.steps) / 1e6, 1), "unit": "MP/s", "n_gpus": D.world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(wall / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u8/i16 entropy decode, f32 IDCT+colour", "data": "fixture",
            "config": {"workload": (f"c6: tests/golden/large.jpg (7680x4320 4:4:4, no restart markers), stream split by bytes over the "
                                    f"ranks (lead-in {a.lead_bytes} bytes): every rank unstuffs and walks its part only, one all-gather "
                                    f"of {RB}-byte reports joins the parts up, MCUs owned per rank gathered on rank 0"),
                       "mcus_first_count_per_rank": state["wins"], "scan_bytes_read_by_rank0": hi - lo, "scan_bytes": len(data),
                       "timing": "device-resident, exchange and gather included",
                       "collective": f"all-gather of one {RB}-byte report per rank + point-to-point gather of pixels to rank 0 ({D.backend})"},
            "roofline": rf,
            "phase_ms_rank0_last_step": {k: round(v, 3) for k, v in state["ms"].items()},
            "step_device_ms": dict(STEP_MS),
            "verified": {"crc32_gathered": crc_gathered, "crc32_whole_image_one_gpu": crc_whole, "equal": crc_gathered == crc_whole,
                         "equals_oracle": crc_gathered == want},
        }
        assert crc_gathered == crc_whole and crc_gathered == want, "gathered image differs"
    del my_bytes
    part.close()
    ctx.close()
    return result


def gather_into(D, mine, recv, shapes, use_cpu):
    """One batched set of point-to-point transfers: every peer sends its band, the root receives each straight into
    its rows of the assembled image.  With the gloo rehearsal backend the bands travel through host memory."""
    torch, dist = D.torch, D.dist
    if use_cpu:
        torch.cuda.current_stream().synchronize()
        if D.rank == 0:
            bufs = [None] + [torch.empty(shapes[r], dtype=torch.uint8) for r in range(1, D.world)]
            reqs = [dist.irecv(bufs[r], r) for r in range(1, D.world) if shapes[r][0]]
            for q in reqs:
                q.wait()
            for r in range(1, D.world):
                if shapes[r][0]:
                    recv[r].copy_(bufs[r])
        elif mine.numel():
            dist.send(mine.cpu(), 0)
        return
    ops = []
    if D.rank == 0:
        for r in range(1, D.world):
            if shapes[r][0]:
                ops.append(dist.P2POp(dist.irecv, recv[r], r))
    elif mine.numel():
        ops.append(dist.P2POp(dist.isend, mine, 0))
    if ops:
        for q in dist.batch_isend_irecv(ops):
            q.wait()


def main():
    a = parse_args()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        return launch_children(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        sys.stderr.write(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; start it as `python bench.py --gpus N` or through "
                         "torch.distributed.run with --nproc-per-node equal to --gpus\n")
        return 2
    D = Dist(a)
    result = {"c2": run_c2, "c3": run_c3, "c4": run_c4, "c5": run_c5, "c6": run_c6}[a.config](a, D)
    D.close()
    if D.rank == 0:
        print(json.dumps(result), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
