"""Multi-GPU work distribution (pure host logic; RCCL only ever moves finished pixels).

* A batch of images shards with no collective: images are independent units, assigned to ranks by
  a static, cost-balanced partition (longest-processing-time first on the scan size, which is what
  the entropy stage's time follows).
* One huge restart-marked image shards by contiguous bands of MCU rows that begin and end on
  restart-interval boundaries; each rank decodes its band as an ordinary image and the bands are
  gathered on the root rank.
* An image without restart markers is one serial chain (unknown bit alignment and DC predictors at any
  cut).  It is split by bytes: every rank walks its part of the stream from a lead-in, one all-gather of a
  32-byte report per rank joins the parts up (RangeBatch), and every rank then owns a run of MCUs in raster
  order -- not whole rows, so its pixels are gathered as rectangles (mcu_range_pieces).  When the parts do not
  join up (a lead-in shorter than the stream needs to synchronise), every rank repeats the entropy front and
  takes a window of rows instead (Batch(rows=...)).
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


def mcu_range_pieces(m0, n, mcux, mw, mh, width, height):
    """Pixel rectangles (y0, y1, xb0, xb1) -- pixel rows [y0, y1), byte columns [xb0, xb1) of an RGB8 row -- that
    the MCUs [m0, m0 + n) in raster order cover, clipped to the image: at most a partial first MCU row, a run of
    whole rows and a partial last row."""
    out = []

    def rect(row, x0, x1):
        y0, y1 = row * mh, min((row + 1) * mh, height)
        xb0, xb1 = min(x0 * mw, width) * 3, min(x1 * mw, width) * 3
        if y1 > y0 and xb1 > xb0:
            out.append((y0, y1, xb0, xb1))

    m, end = m0, m0 + n
    if m < end and m % mcux:
        stop = min(end, (m // mcux + 1) * mcux)
        rect(m // mcux, m % mcux, stop - (m // mcux) * mcux)
        m = stop
    if end - m >= mcux:
        r0, r1 = m // mcux, end // mcux
        y0, y1 = r0 * mh, min(r1 * mh, height)
        if y1 > y0:
            out.append((y0, y1, 0, width * 3))
        m = r1 * mcux
    if m < end:
        rect(m // mcux, 0, end - m)
    return out


def pack_pieces(window, y_origin, pieces):
    """The rectangles `pieces` of `window` (a 2-D uint8 tensor whose row 0 is pixel row y_origin) as one flat tensor."""
    import torch
    parts = [window[y0 - y_origin: y1 - y_origin, xb0:xb1].reshape(-1) for (y0, y1, xb0, xb1) in pieces]
    return torch.cat(parts) if parts else window.new_empty(0)


def pieces_bytes(pieces):
    return sum((y1 - y0) * (xb1 - xb0) for (y0, y1, xb0, xb1) in pieces)


def unpack_pieces(full, flat, pieces):
    """Inverse of pack_pieces into the assembled image `full` (row 0 = pixel row 0)."""
    at = 0
    for (y0, y1, xb0, xb1) in pieces:
        nb = (y1 - y0) * (xb1 - xb0)
        full[y0:y1, xb0:xb1].copy_(flat[at: at + nb].reshape(y1 - y0, xb1 - xb0))
        at += nb


def split_windows(reports, frame_mcus):
    """(first MCU, MCUs) of every part from the gathered reports (bytes each), as zj_batch_range_resolve assigns
    them: every part the MCUs it reports, the last part what the frame still lacks."""
    import struct
    mcus = [struct.unpack_from("<I", bytes(r), 16)[0] for r in reports]
    out, at = [], 0
    for k, n in enumerate(mcus):
        if k == len(mcus) - 1:
            n = frame_mcus - at
        out.append((at, n))
        at += n
    return out
