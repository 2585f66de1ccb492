"""Multi-GPU work distribution (pure host logic; RCCL only ever moves finished pixels).

* A batch of images shards with no collective: images are independent units, assigned to ranks by
  a static, cost-balanced partition (longest-processing-time first on the scan size, which is what
  the entropy stage's time follows).
* One huge restart-marked image shards by contiguous bands of MCU rows that begin and end on
  restart-interval boundaries; each rank decodes its band as an ordinary image and the bands are
  gathered on the root rank.
* An image without restart markers does not shard (serial DC chain, unknown bit alignment at any
  cut): replicas only.
"""
import heapq


def partition_batch(costs, world_size):
    """Greedy LPT: returns a list (per rank) of sorted image indices.  costs[i] ~ scan bytes of image i."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    heap = [(0, r) for r in range(world_size)]
    heapq.heapify(heap)
    parts = [[] for _ in range(world_size)]
    for i in order:
        load, r = heapq.heappop(heap)
        parts[r].append(i)
        heapq.heappush(heap, (load + costs[i], r))
    return [sorted(p) for p in parts]


def mcu_geometry(width, height, hs, vs):
    mw, mh = 8 * hs, 8 * vs
    return (width + mw - 1) // mw, (height + mh - 1) // mh, mw, mh


def band_rows(mcux, mcuy, restart_interval, world_size):
    """Split mcuy MCU rows into <= world_size contiguous bands (row0, nrows) whose boundaries fall on
    restart-interval boundaries.  Raises ValueError when the image cannot be cut that way."""
    if restart_interval <= 0:
        raise ValueError("image has no restart markers: replicas only")
    # smallest number of MCU rows after which an interval boundary coincides with a row boundary
    step = None
    for r in range(1, mcuy + 1):
        if (r * mcux) % restart_interval == 0:
            step = r
            break
    if step is None:
        raise ValueError("restart intervals never align with MCU rows")
    units = (mcuy + step - 1) // step
    bands, row = [], 0
    for r in range(world_size):
        u = units // world_size + (1 if r < units % world_size else 0)
        nrows = min(u * step, mcuy - row)
        if nrows > 0:
            bands.append((row, nrows))
        else:
            bands.append((row, 0))
        row += nrows
    return bands


def gather_bands(local_band, band_shapes, rank, world_size, root=0, group=None):
    """Gather per-rank bands (torch uint8 tensors [rows, row_bytes]) on `root` with one batched set of
    point-to-point transfers: every peer sends its band over its own link, the root posts all receives
    at once (no ring, no all-gather).  Returns the list of bands on root, None elsewhere.
    band_shapes: list of (rows, row_bytes) for all ranks (known to everyone from the geometry)."""
    import torch
    import torch.distributed as dist

    if world_size == 1:
        return [local_band]
    ops, out = [], None
    if rank == root:
        out = []
        for r in range(world_size):
            if r == root:
                out.append(local_band)
                continue
            buf = torch.empty(band_shapes[r], dtype=torch.uint8, device=local_band.device)
            out.append(buf)
            if buf.numel():
                ops.append(dist.P2POp(dist.irecv, buf, r, group=group))
    else:
        if local_band.numel():
            ops.append(dist.P2POp(dist.isend, local_band.contiguous(), root, group=group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out
