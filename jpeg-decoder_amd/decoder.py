"""Host-side mirror of the reference's decoder API on top of the C ABI.

Reference API (src/lib.rs:5, src/decoder.rs:322-330, src/error.rs:3-12):
    Decoder::new(file: File) -> Decoder
    Decoder::decode(&mut self) -> Result<(), DecodeError>      # writes ./out.ppm
Here:
    Decoder(file).decode()            # same side effect, raises DecodeError instead of returning Err
    Decoder(file).decode_rgb()        # returns the pixels as a numpy array
plus the batch entry points the GPU path is built for (ParsedImage / Batch).
All device memory is handled through torch tensors (plumbing only); every byte of decode work is
done by the HIP kernels behind libzj.so.
"""
import ctypes as C

import numpy as np

from . import binding as B


class DecodeError(Exception):
    """Counterpart of `enum DecodeError` (src/error.rs:4-6); `kind` is one of
    'io', 'format', 'unsupported', 'device', 'corrupt'."""

    def __init__(self, kind, message, status=0):
        super().__init__(f"{kind}: {message}")
        self.kind, self.status = kind, status


class ParsedImage:
    """A JPEG file parsed on the host (zj_parse): tables, geometry and the scan extent."""

    def __init__(self, data: bytes):
        self._data = bytes(data)  # keeps image.scan alive
        self._buf = C.create_string_buffer(self._data, len(self._data))
        self.parsed = B.zj_parsed()
        rc = B.lib().zj_parse(C.cast(self._buf, C.c_void_p), len(self._data), C.byref(self.parsed))
        if rc == B.ZJ_E_UNSUPPORTED:
            raise DecodeError("unsupported", "JPEG variant outside the baseline 3-component path")
        if rc != B.ZJ_OK:
            raise DecodeError("format", "malformed JPEG container")

    @property
    def image(self):
        return self.parsed.image

    @property
    def width(self):
        return self.parsed.image.width

    @property
    def height(self):
        return self.parsed.image.height

    @property
    def nblocks(self):
        return B.lib().zj_image_blocks(C.byref(self.parsed.image))

    def restart_offsets(self):
        """Byte offsets inside the scan at which restart intervals begin (zj_restart_offsets, the host loop)."""
        n = B.lib().zj_restart_offsets(C.byref(self.parsed.image), None, 0)
        buf = (C.c_size_t * max(n, 1))()
        B.lib().zj_restart_offsets(C.byref(self.parsed.image), buf, n)
        return [int(buf[k]) for k in range(n)]

    def band(self, row0, nrows):
        """Sub-image of MCU rows [row0, row0+nrows) of a restart-marked image (zj_image_band)."""
        out = BandImage(self)
        rc = B.lib().zj_image_band(C.byref(self.parsed.image), row0, nrows, C.byref(out.img))
        if rc != B.ZJ_OK:
            raise DecodeError("format", "band does not start and end on restart-interval boundaries")
        return out


class BandImage:
    """A horizontal band of a restart-marked image; borrows the parent's tables and bytes."""

    def __init__(self, parent):
        self.parent = parent
        self.img = B.zj_image()

    @property
    def image(self):
        return self.img

    @property
    def width(self):
        return self.img.width

    @property
    def height(self):
        return self.img.height

    @property
    def nblocks(self):
        return B.lib().zj_image_blocks(C.byref(self.img))


class Context:
    """One GPU, one stream (zj_create).  stream=None lets the library create its own."""

    def __init__(self, device=0, stream=None):
        import torch
        if not torch.cuda.is_available():
            raise DecodeError("device", "no GPU visible: the decode path has no CPU fallback")
        self.device = device
        self._torch = torch
        torch.cuda.set_device(device)
        handle = None
        if stream == "torch":
            handle = torch.cuda.current_stream(device).cuda_stream
        elif stream is not None:
            handle = stream
        self.h = B.lib().zj_create(device, handle)
        if not self.h:
            raise DecodeError("device", "zj_create failed")

    def last_error(self):
        return B.lib().zj_last_error(self.h).decode()

    def close(self):
        if self.h:
            B.lib().zj_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def image_array(images):
    """ctypes array of zj_image for a list of ParsedImage / BandImage."""
    arr = (B.zj_image * max(len(images), 1))()
    for i, im in enumerate(images):
        arr[i] = im.image
    return arr


class Workspace:
    """Device memory shared by batches that decode one after the other (zj_workspace_create): one scratch
    range (coefficient arena and all intermediates) and `nslots` plan ranges.  Sizes in bytes; use
    Workspace.estimate(images) for the upper bounds of one batch."""

    def __init__(self, ctx: Context, scratch_bytes, plan_bytes, nslots=2):
        self.ctx = ctx
        self.nslots = nslots
        self.h = B.lib().zj_workspace_create(ctx.h, scratch_bytes, plan_bytes, nslots)
        if not self.h:
            raise DecodeError("device", "zj_workspace_create failed: " + ctx.last_error())

    @staticmethod
    def estimate(images):
        plan, scratch = C.c_size_t(), C.c_size_t()
        arr = image_array(list(images))
        rc = B.lib().zj_workspace_estimate(arr, len(images), C.byref(plan), C.byref(scratch))
        if rc != B.ZJ_OK:
            raise DecodeError("format", "zj_workspace_estimate failed")
        return plan.value, scratch.value

    def close(self):
        if self.h:
            B.lib().zj_workspace_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class OutputTables:
    """Addresses and row pitches of a batch's output tensors, checked against the batch's images."""

    def __init__(self, batch, rgb_tensors):
        n = batch.n
        self.batch, self.tensors = batch, list(rgb_tensors)
        self.ptrs = (C.c_void_p * max(n, 1))()
        self.strides = (C.c_size_t * max(n, 1))()
        for i, t in enumerate(self.tensors):
            if t is not None:
                check_output_tensor(t, batch.images[i], batch.ctx.device, batch.window_height(i))
                self.ptrs[i] = t.data_ptr()
                self.strides[i] = t.stride(0)


class Batch:
    """A planned batch of parsed images, resident on the GPU (zj_batch_create, or zj_batch_create_in
    when a workspace and one of its plan slots are given)."""

    def __init__(self, ctx: Context, images, subseq_bits=0, workspace=None, slot=0, rows=None):
        self.ctx = ctx
        self.images = list(images)
        n = len(self.images)
        self.n = n
        arr = image_array(self.images)
        self._arr = arr
        self.workspace = workspace
        self.rows = rows  # per image (first MCU row, MCU rows) of the output window, or None
        if rows is not None:
            if workspace is not None:
                raise DecodeError("device", "row windows and workspaces are separate entry points")
            r0 = (C.c_uint32 * max(n, 1))(*[r[0] if r else 0 for r in rows])
            nr = (C.c_uint32 * max(n, 1))(*[r[1] if r else 0 for r in rows])
            self.h = B.lib().zj_batch_create_rows(ctx.h, arr, n, subseq_bits, r0, nr)
        elif workspace is None:
            self.h = B.lib().zj_batch_create(ctx.h, arr, n, subseq_bits)
        else:
            self.h = B.lib().zj_batch_create_in(workspace.h, slot, arr, n, subseq_bits)
        if not self.h:
            raise DecodeError("device", "zj_batch_create failed: " + ctx.last_error())
        self._rgb = None

    def alloc_rgb(self):
        """Device output buffers (torch uint8 tensors, rows padded to 16 bytes)."""
        torch = self.ctx._torch
        outs = []
        for i, im in enumerate(self.images):
            stride = (im.width * 3 + 15) // 16 * 16
            outs.append(torch.empty((self.window_height(i), stride), dtype=torch.uint8, device=f"cuda:{self.ctx.device}"))
        return outs

    def window_height(self, i):
        """Pixel rows image i produces: all of them, or those of its window of MCU rows."""
        im = self.images[i]
        if not self.rows or not self.rows[i] or not self.rows[i][1]:
            return im.height
        mh = 8 * max(im.image.vsamp[0], 1)
        return min(self.rows[i][1] * mh, im.height - self.rows[i][0] * mh)

    def output_tables(self, rgb_tensors):
        """Check the output tensors once and keep their addresses and pitches: the result can be passed to
        decode() / decode_pixels() in place of the list, which saves the per-tensor checks on every call."""
        return OutputTables(self, rgb_tensors)

    def _output_tables(self, rgb_tensors):
        if rgb_tensors is None:
            self._rgb = None
            return None, (C.c_size_t * max(self.n, 1))()
        t = rgb_tensors if isinstance(rgb_tensors, OutputTables) else OutputTables(self, rgb_tensors)
        if t.batch is not self:
            raise ValueError("output tables of another batch")
        self._rgb = t.tensors  # keeps them alive while the GPU writes
        return t.ptrs, t.strides

    def decode(self, rgb_tensors=None):
        """Launch the hot path (asynchronous).  rgb_tensors: list of 2-D uint8 CUDA tensors or None."""
        ptrs, strides = self._output_tables(rgb_tensors)
        rc = B.lib().zj_batch_decode(self.h, ptrs, strides)
        if rc != B.ZJ_OK:
            raise DecodeError("device", self.ctx.last_error())

    def decode_entropy(self):
        """First half of decode(): the entropy passes, on the context's stream (zj_batch_decode_entropy)."""
        if B.lib().zj_batch_decode_entropy(self.h) != B.ZJ_OK:
            raise DecodeError("device", self.ctx.last_error())

    def decode_pixels(self, rgb_tensors=None, stream=None):
        """Second half of decode(): IDCT and colour conversion, on `stream` (a raw HIP stream handle; None: the
        context's stream), which is made to wait for the entropy passes (zj_batch_decode_pixels)."""
        ptrs, strides = self._output_tables(rgb_tensors)
        rc = B.lib().zj_batch_decode_pixels(self.h, ptrs, strides, stream)
        if rc != B.ZJ_OK:
            raise DecodeError("device", self.ctx.last_error())

    def finish(self):
        """Wait and return the per-image status words (numpy uint32)."""
        st = np.zeros(max(self.n, 1), np.uint32)
        rc = B.lib().zj_batch_finish(self.h, st.ctypes.data_as(C.POINTER(C.c_uint32)))
        if rc != B.ZJ_OK:
            raise DecodeError("device", self.ctx.last_error())
        return st[: self.n]

    def coefficients(self, i, dequantised=False):
        """int16 [nblocks, 64]: zig-zag quantised (stage seam) or natural-order dequantised
        (the reference's `blocks`, src/decoder.rs:387)."""
        nb = self.images[i].nblocks
        out = np.empty((nb, 64), np.int16)
        p = out.ctypes.data_as(C.c_void_p)
        rc = B.lib().zj_batch_export_coef(self.h, i, None if dequantised else p, p if dequantised else None, B.ZJ_MEM_HOST)
        if rc != B.ZJ_OK:
            raise DecodeError("device", self.ctx.last_error())
        return out

    def set_profiling(self, on=True):
        B.lib().zj_batch_set_profiling(self.h, 1 if on else 0)

    def kernel_times(self):
        ms = (C.c_float * B.ZJ_NUM_KERNELS)()
        rc = B.lib().zj_batch_kernel_times(self.h, ms)
        if rc != B.ZJ_OK:
            raise DecodeError("device", "no timing available")
        return {B.lib().zj_kernel_name(k).decode(): float(ms[k]) for k in range(B.ZJ_NUM_KERNELS)}

    def byte_counts(self):
        v = [C.c_uint64() for _ in range(4)]
        B.lib().zj_batch_bytes(self.h, *[C.byref(x) for x in v])
        return {"scan": v[0].value, "coef": v[1].value, "rgb": v[2].value, "pixels": v[3].value}

    def counters(self):
        """(fix-up lanes that ran through a whole chunk, blocks that ended off their successor's position,
        finish calls that ran the exhaustive fix-up loop) over the batch's life; all 0 on the fast path."""
        c = (C.c_uint32 * 3)()
        rc = B.lib().zj_batch_counters(self.h, c)
        if rc != B.ZJ_OK:
            raise DecodeError("device", self.ctx.last_error())
        return int(c[0]), int(c[1]), int(c[2])

    def memory(self):
        """(plan bytes, scratch bytes) taken from a workspace; (0, 0) for a batch that owns its memory."""
        p, s_ = C.c_size_t(), C.c_size_t()
        B.lib().zj_batch_memory(self.h, C.byref(p), C.byref(s_))
        return p.value, s_.value

    def segments(self, i):
        """(begin, end) byte offsets inside image i's scan of the entropy-coded segments it was cut into."""
        n = B.lib().zj_batch_segments(self.h, i, None, 0)
        buf = (C.c_uint64 * max(2 * n, 1))()
        B.lib().zj_batch_segments(self.h, i, buf, n)
        return [(int(buf[2 * k]), int(buf[2 * k + 1])) for k in range(n)]

    def close(self):
        if self.h:
            B.lib().zj_batch_destroy(self.h)
            self.h = None
        self._rgb = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ChainError(DecodeError):
    """The parts of a split stream do not join up (ZJ_E_CHAIN): decode the stream whole, or with row windows."""

    def __init__(self, message):
        super().__init__("chain", message)


class RangeBatch:
    """Part `rank` of `nranks` of one image without restart markers (zj_batch_create_range): unstuffing, walks, block
    positions and DC differences cover this part's bytes only (plus `lead_bytes` in front of its boundary).  Use:
        report = part.front()                  # bytes, to be all-gathered by the caller
        first, count, row0, nrows = part.resolve(reports_of_all_parts)    # ChainError: fall back to Batch(rows=...)
        part.decode(window)                    # window: uint8 tensor that begins at MCU row row0
        part.finish()
    """

    REPORT_BYTES = C.sizeof(B.zj_range_report)

    def __init__(self, ctx: Context, image, rank, nranks, lead_bytes=16384, subseq_bits=0):
        self.ctx, self.image, self.rank, self.nranks = ctx, image, rank, nranks
        self.n = 1
        self._img = image.image
        self.h = B.lib().zj_batch_create_range(ctx.h, C.byref(self._img), rank, nranks, lead_bytes, subseq_bits)
        if not self.h:
            raise DecodeError("device", "zj_batch_create_range failed: " + ctx.last_error())
        self.window = None  # (first MCU, MCUs, first MCU row, MCU rows) once resolved
        self._rgb = None

    def front(self) -> bytes:
        rep = B.zj_range_report()
        rc = B.lib().zj_batch_range_front(self.h, C.byref(rep))
        if rc == B.ZJ_E_CHAIN:
            raise ChainError(self.ctx.last_error())
        if rc != B.ZJ_OK:
            raise DecodeError("device", self.ctx.last_error())
        return bytes(rep)

    def resolve(self, reports):
        """reports: the `nranks` reports in rank order (bytes each, or one bytes object of all of them)."""
        blob = reports if isinstance(reports, (bytes, bytearray)) else b"".join(bytes(r) for r in reports)
        if len(blob) != self.nranks * self.REPORT_BYTES:
            raise ValueError("one report per part")
        arr = (B.zj_range_report * self.nranks).from_buffer_copy(blob)
        v = [C.c_uint32() for _ in range(4)]
        rc = B.lib().zj_batch_range_resolve(self.h, arr, self.nranks, *[C.byref(x) for x in v])
        if rc == B.ZJ_E_CHAIN:
            raise ChainError(self.ctx.last_error())
        if rc != B.ZJ_OK:
            raise DecodeError("device", self.ctx.last_error())
        self.window = tuple(x.value for x in v)
        return self.window

    def window_height(self):
        im = self.image
        mh = 8 * max(im.image.vsamp[0], 1)
        _, _, row0, nrows = self.window
        return min(nrows * mh, im.height - row0 * mh) if nrows else 0

    def decode(self, rgb):
        """rgb: 2-D uint8 CUDA tensor of at least window_height() rows; its row 0 is the first pixel row of MCU row row0."""
        if self.window is None:
            raise DecodeError("device", "resolve() first")
        ptr, stride = None, 0
        if self.window[3]:
            check_output_tensor(rgb, self.image, self.ctx.device, self.window_height())
            ptr, stride = rgb.data_ptr(), rgb.stride(0)
        self._rgb = rgb
        rc = B.lib().zj_batch_range_decode(self.h, ptr, stride)
        if rc != B.ZJ_OK:
            raise DecodeError("device", self.ctx.last_error())

    def finish(self):
        st = np.zeros(1, np.uint32)
        rc = B.lib().zj_batch_finish(self.h, st.ctypes.data_as(C.POINTER(C.c_uint32)))
        if rc == B.ZJ_E_CHAIN:
            raise ChainError(self.ctx.last_error())
        if rc != B.ZJ_OK:
            raise DecodeError("device", self.ctx.last_error())
        return st

    def coefficients(self):
        """int16 [MCUs owned * blocks per MCU, 64], zig-zag quantised: the coefficients of the MCUs this part owns."""
        nb = self.image.nblocks
        out = np.empty((nb, 64), np.int16)
        rc = B.lib().zj_batch_export_coef(self.h, 0, out.ctypes.data_as(C.c_void_p), None, B.ZJ_MEM_HOST)
        if rc != B.ZJ_OK:
            raise DecodeError("device", self.ctx.last_error())
        bpm = nb // B.lib().zj_image_mcus(C.byref(self._img))
        return out[self.window[0] * bpm: (self.window[0] + self.window[1]) * bpm]

    def segments(self):
        """(begin, end) byte offsets inside the scan of the bytes this part reads."""
        buf = (C.c_uint64 * 2)()
        B.lib().zj_batch_segments(self.h, 0, buf, 1)
        return int(buf[0]), int(buf[1])

    def close(self):
        if self.h:
            B.lib().zj_batch_destroy(self.h)
            self.h = None
        self._rgb = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def check_output_tensor(t, im, device, height=None):
    """A pixel output must be a 2-D uint8 tensor on the context's GPU with unit inner stride, at least
    `height` rows and a row pitch of at least 3 * width bytes: the kernels write exactly that region."""
    if str(t.dtype) != "torch.uint8":
        raise DecodeError("device", f"output tensor must be uint8, got {t.dtype}")
    if t.dim() != 2 or not t.is_cuda or t.device.index != device:
        raise DecodeError("device", f"output tensor must be 2-D on cuda:{device}")
    if t.shape[1] > 1 and t.stride(1) != 1:
        raise DecodeError("device", "output tensor rows must be contiguous")
    height = im.height if height is None else height
    if t.shape[0] < height or t.shape[1] < 3 * im.width or (height > 1 and t.stride(0) < 3 * im.width):
        raise DecodeError("device", f"output tensor {tuple(t.shape)} (pitch {t.stride(0)}) is too small for {im.width}x{im.height} RGB")


def decode_noalloc(ctx: Context, images, rgb_tensors):
    """zj_decode_batch_noalloc: one call, device outputs, no device allocation.  Returns the status words."""
    n = len(images)
    arr = image_array(list(images))
    ptrs = (C.c_void_p * max(n, 1))()
    strides = (C.c_size_t * max(n, 1))()
    for i, t in enumerate(rgb_tensors):
        if t is not None:
            check_output_tensor(t, images[i], ctx.device)
            ptrs[i] = t.data_ptr()
            strides[i] = t.stride(0)
    st = np.zeros(max(n, 1), np.uint32)
    rc = B.lib().zj_decode_batch_noalloc(ctx.h, arr, n, ptrs, strides, st.ctypes.data_as(C.POINTER(C.c_uint32)))
    if rc != B.ZJ_OK:
        raise DecodeError("nomem" if rc == B.ZJ_E_NOMEM else "device", ctx.last_error())
    return st[:n]


def decode_images(ctx: Context, images, subseq_bits=0, want_coef=False):
    """Decode parsed images; returns (list of HxWx3 uint8 numpy arrays, status array[, coefficient list])."""
    b = Batch(ctx, images, subseq_bits)
    try:
        outs = b.alloc_rgb()
        b.decode(outs)
        st = b.finish()
        rgb = [t[:, : im.width * 3].reshape(im.height, im.width, 3).cpu().numpy() for t, im in zip(outs, images)]
        if want_coef:
            return rgb, st, [b.coefficients(i) for i in range(b.n)]
        return rgb, st
    finally:
        b.close()


class Decoder:
    """`zen_jpeg::Decoder` (src/decoder.rs:148-151, :322-330) with the entropy decode, IDCT and
    colour conversion running on the GPU."""

    def __init__(self, file, device=0):  # Decoder::new(file)
        self.file = file
        self.device = device
        self.w = self.h = 0  # Dimensions, src/decoder.rs:153-156

    def decode_rgb(self):
        try:
            data = self.file.read()
        except OSError as e:  # DecodeError::Io
            raise DecodeError("io", str(e))
        img = ParsedImage(data)
        self.w, self.h = img.width, img.height
        ctx = Context(self.device)
        try:
            rgb, st = decode_images(ctx, [img])
        finally:
            ctx.close()
        if st[0]:
            raise DecodeError("corrupt", "entropy-coded data is corrupt or truncated", int(st[0]))
        return rgb[0]

    def decode(self, out_path="out.ppm"):  # Decoder::decode: writes out.ppm (src/decoder.rs:334,580,648)
        rgb = np.ascontiguousarray(self.decode_rgb())
        rc = B.lib().zj_write_ppm(out_path.encode(), self.w, self.h, rgb.ctypes.data_as(C.c_void_p), 0)
        if rc != B.ZJ_OK:
            raise DecodeError("io", f"cannot write {out_path}")
