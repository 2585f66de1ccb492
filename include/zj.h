/*
 * zj.h -- C ABI of the MI355X baseline-JPEG decode path.
 *
 * The reference (crate `zen-jpeg`) has no FFI boundary of its own: its whole product is
 * `Decoder::decode()` (src/decoder.rs:330-651).  This header is the boundary a maintainer would
 * bind from Rust: the host keeps the marker walk and the DQT / DHT / SOF0 / SOS parsing
 * (src/decoder.rs:347-642) and hands the parsed state to the device at the two places where the
 * reference starts its hot loops -- the SOS arm (src/decoder.rs:366-390) and `to_rgb`
 * (src/decoder.rs:644-646).  Plain pointers and sizes only; nothing here unwinds or aborts.
 *
 * Threading: a zj_ctx is bound to one GPU and must be used from one thread at a time; one ctx
 * (and one process or host thread) per GPU.  Calls on distinct contexts are independent.
 * Ownership: the caller owns every pointer it passes; inputs are borrowed for the duration of
 * the call that takes them (zj_batch_create copies what it needs to the device); outputs are
 * caller-allocated.  The library owns all device scratch inside zj_ctx / zj_batch.
 */
#ifndef ZJ_H
#define ZJ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- return codes of the functions below (0 = success, negative = failure) ---- */
#define ZJ_OK 0
#define ZJ_E_INVALID (-1)     /* bad argument                                                       */
#define ZJ_E_FORMAT (-2)      /* malformed container (the reference panics: src/decoder.rs:358 ...) */
#define ZJ_E_UNSUPPORTED (-3) /* valid JPEG outside the path (progressive, 12-bit, 4 components)     */
#define ZJ_E_DEVICE (-4)      /* HIP runtime error or no usable GPU; see zj_last_error              */
#define ZJ_E_NOMEM (-5)
#define ZJ_E_IO (-6)          /* counterpart of DecodeError::Io (src/error.rs:4-6)                  */
#define ZJ_E_CHAIN (-7)       /* a stream split over several GPUs: the parts' walks do not join up; decode it whole */

/* ---- per-image status word (bit set; 0 = decoded cleanly).  Each bit replaces one of the
 *      reference's panics on malformed entropy data. ---- */
#define ZJ_ST_BAD_CODE 1u      /* bit pattern that is no code of the table (src/ec.rs:27,47 OOB)   */
#define ZJ_ST_COEF_OVERFLOW 2u /* run past coefficient 63            (src/decoder.rs:228 OOB)      */
#define ZJ_ST_TRUNCATED 4u     /* entropy data ended early           (src/decoder.rs:197,222)      */
#define ZJ_ST_UNSUPPORTED 8u   /* image rejected at plan time, nothing decoded                      */

/* One DHT table as it appears in the file: BITS and HUFFVAL (src/decoder.rs:497-513). */
typedef struct zj_huff {
    uint8_t counts[16];
    uint8_t symbols[256];
} zj_huff;

/*
 * One parsed image = the state the reference holds when it reaches the SOS arm:
 * dimensions (src/decoder.rs:570-583), quantisation tables in file (zig-zag) order (:453),
 * Huffman tables indexed [table id][0 = AC, 1 = DC] (:553), plus what the reference reads and
 * drops: sampling factors (:612-613), SOS selectors (:368-370) and the restart interval.
 * `scan` is the entropy-coded segment exactly as stored: stuffed bytes and RSTn markers
 * included, from the first byte after the SOS header up to (not including) the terminating
 * marker.  HOST memory.
 */
typedef struct zj_image {
    const uint8_t *scan;
    size_t scan_len;
    uint16_t width, height;
    uint8_t ncomp;       /* 3, or 1 (grayscale: entries [0] only; hsamp[0] = vsamp[0] = 1) */
    uint8_t hsamp[3];    /* luma 1 or 2, chroma 1 */
    uint8_t vsamp[3];
    uint8_t tq[3];       /* component -> quantisation table id */
    uint8_t td[3];       /* component -> DC table id (0 or 1)  */
    uint8_t ta[3];       /* component -> AC table id (0 or 1)  */
    uint16_t restart_interval; /* MCUs per restart interval, 0 = no restart markers */
    const uint8_t *qt[4];      /* 64 bytes each, zig-zag order; NULL if undefined    */
    const zj_huff *ht[2][2];   /* [id][0 = AC, 1 = DC]; NULL if undefined            */
} zj_image;

/* Result of the host-side container parse; owns copies of the tables, borrows the file bytes. */
typedef struct zj_parsed {
    zj_image image; /* image.scan points into the caller's file buffer */
    uint8_t qt_store[4][64];
    zj_huff ht_store[2][2];
    uint32_t n_restart_markers; /* RSTn markers found inside the scan */
} zj_parsed;

/* Where an output pointer lives. */
#define ZJ_MEM_HOST 0
#define ZJ_MEM_DEVICE 1

/* Outputs of one image.  Any pointer may be NULL.
 *   rgb    interleaved RGB8, rows `rgb_stride` bytes apart (0 = 3*width): the payload of the
 *          reference's out.ppm (src/decoder.rs:644-648)
 *   qcoef  int16 [nblocks][64], zig-zag order, before dequantisation, DC after prediction;
 *          blocks in scan order (MCU-major; inside an MCU luma blocks row by row, then Cb, Cr)
 *   dqcoef int16 [nblocks][64], natural order, dequantised: the reference's `blocks`
 *          (src/decoder.rs:387) */
typedef struct zj_output {
    uint8_t *rgb;
    size_t rgb_stride;
    int16_t *qcoef;
    int16_t *dqcoef;
    int mem; /* ZJ_MEM_HOST or ZJ_MEM_DEVICE, applies to all three pointers */
} zj_output;

typedef struct zj_ctx zj_ctx;
typedef struct zj_batch zj_batch;

/* ---------------------------------------------------------------- host side (no GPU needed) */

/* Container parse: replaces the marker loop of Decoder::decode (src/decoder.rs:347-642) for
 * everything except the SOS hot loop.  Accepts every APPn / COM (skipped by length), several
 * tables per DQT / DHT segment, DRI.  Returns ZJ_OK, ZJ_E_FORMAT or ZJ_E_UNSUPPORTED. */
int zj_parse(const uint8_t *file, size_t len, zj_parsed *out);

/* Geometry helpers derived from a parsed image. */
uint32_t zj_image_blocks(const zj_image *img);       /* number of 8x8 blocks in scan order */
uint32_t zj_image_mcus(const zj_image *img);

/* Byte offsets (relative to img->scan) at which restart intervals begin: entry 0 is 0, entry k
 * is the byte after the k-th RSTn marker.  Writes at most `cap` entries, returns the number of
 * intervals found.  Used to cut one huge restart-marked image into per-GPU bands. */
size_t zj_restart_offsets(const zj_image *img, size_t *offsets, size_t cap);

/* Build the sub-image that covers MCU rows [row0, row0 + nrows) of `img`.  Requires restart
 * intervals that tile MCU rows exactly ((mcus per row) % restart_interval == 0, or an interval
 * of whole rows).  `band` borrows img's tables and points into img's scan; its height is the band's pixel
 * height (clipped to the image).  Returns ZJ_OK or ZJ_E_INVALID. */
int zj_image_band(const zj_image *img, uint32_t row0, uint32_t nrows, zj_image *band);

/* PPM writer: header as src/decoder.rs:580, payload as :648. */
int zj_write_ppm(const char *path, uint32_t width, uint32_t height, const uint8_t *rgb, size_t stride);

/* ---------------------------------------------------------------- device side */

int zj_device_count(void);

/* Create a context on GPU `device`.  `stream` is a hipStream_t to launch on (NULL = a stream
 * the context creates and owns).  Returns NULL on failure. */
zj_ctx *zj_create(int device, void *stream);
void zj_destroy(zj_ctx *ctx);
const char *zj_last_error(zj_ctx *ctx);

/* Tuning knob: bits per bitstream subsequence handed to one lane of the Huffman stage
 * (512..1048576, rounded per segment to whole 512-bit windows; 0 = chosen from the batch size). */
#define ZJ_FLAG_SUBSEQ_BITS(b) ((uint32_t)(b))
/* Testing aid: build the lookup tables without their second level, so that every code longer than the
 * first level goes through the canonical search (the path that otherwise only unusual tables reach). */
#define ZJ_FLAG_NO_SECOND_LEVEL 0x1000000u
/* Testing aid: list the restart intervals with the host loop (zj_restart_offsets) instead of the
 * marker search on the device that zj_batch_create otherwise runs over the uploaded scans. */
#define ZJ_FLAG_HOST_MARKER_SCAN 0x2000000u

/* Plan a batch: validates the images, builds the device tables, uploads the scans, allocates all
 * scratch and the coefficient arena.  Everything the decode call needs is resident in HBM when
 * this returns.  Images that are unsupported keep a slot and report ZJ_ST_UNSUPPORTED.  A scan of
 * 0xF0000000 bytes or more is refused (ZJ_E_INVALID): positions inside a scan are 32-bit. */
zj_batch *zj_batch_create(zj_ctx *ctx, const zj_image *imgs, size_t n, uint32_t flags);
void zj_batch_destroy(zj_batch *b);

/* As zj_batch_create, with a window of MCU rows per image: coefficients and pixels are produced for MCU rows
 * [row0[i], row0[i] + nrows[i]) only (nrows[i] == 0, or NULL arrays: the whole image); rgb[i] of the decode call
 * then begins at the window's first pixel row.  The entropy front -- unstuffing, synchronisation, block positions,
 * DC prediction -- still runs over the whole image: in a stream without restart markers nothing says where a row
 * begins or what its predictors are until everything before it has been walked (the serial chain of
 * src/decoder.rs:376-389).  This is how such an image is spread over several GPUs: every rank repeats the front
 * (about a third of the work) and takes its band of rows of the rest, and the bands are gathered like those of a
 * restart-marked image.  Coefficients outside the window are not produced (zj_batch_export_coef returns the
 * arena as is there). */
zj_batch *zj_batch_create_rows(zj_ctx *ctx, const zj_image *imgs, size_t n, uint32_t flags, const uint32_t *row0, const uint32_t *nrows);

/* ---- one image without restart markers over several GPUs, entropy front included ----
 * The reference decodes such a stream as one serial chain (src/decoder.rs:376-389: one BitReader, predictors carried
 * from MCU to MCU).  Here part `rank` of `nranks` takes the bytes from `lead_bytes` in front of its boundary (rank *
 * scan_len / nranks, moved off FF 00 pairs) to a little behind the next boundary, walks them from a guessed state as
 * the subsequences of one GPU do, and owns the MCUs that begin between the two boundaries.  What it cannot know alone
 * comes from one all-gather of the parts' reports (struct zj_range_report, 32 bytes each), which the caller performs:
 *   zj_batch_create_range   plan; unstuffing, walks, block positions and DC differences will cover this part only
 *   zj_batch_range_front    run them; fills this part's report                                   (synchronous)
 *   -- all-gather of the reports, by the caller (RCCL / gloo / MPI ...) --
 *   zj_batch_range_resolve  every part checks the whole chain the same way: part r's walk is the true one if part
 *                           r-1's is and both find the same block, at the same place in its MCU, first behind their
 *                           boundary.  Gives this part the raster index of its first MCU, its MCU count, the MCU rows
 *                           its pixels touch, and (internally) its DC predictors = the sums of the parts before it.
 *                           ZJ_E_CHAIN: some part had not synchronised inside its lead-in; every part gets the same
 *                           answer, and the caller falls back to zj_batch_create_rows (front repeated on every GPU).
 *   zj_batch_range_decode   coefficients of the MCUs owned, then IDCT + colour of MCU rows [row0, row0 + nrows) into
 *                           `rgb` (which begins at MCU row row0).  Pixels of MCUs of those rows that other parts own
 *                           are not meaningful: the first and the last row are shared with the neighbours, who write
 *                           the rest of them.  Asynchronous; zj_batch_finish as usual (ZJ_E_CHAIN there too if a
 *                           block did not end where the next begins).  May be called again (same result). */
typedef struct zj_range_report {
    int32_t start_delta;  /* bits from this part's boundary to the first block that begins at or behind it */
    uint32_t start_k;     /* that block's place in its MCU according to this part's own walk */
    int32_t end_delta;    /* the same at the next part's boundary (0 for the last part) */
    uint32_t end_k;
    uint32_t mcus;        /* MCUs that begin between the two boundaries */
    int16_t dc_sum[3];    /* their DC differences summed per component (wrapping i16, src/decoder.rs:199) */
    uint16_t reserved;
    uint32_t reserved2;
} zj_range_report;
zj_batch *zj_batch_create_range(zj_ctx *ctx, const zj_image *img, uint32_t rank, uint32_t nranks, uint32_t lead_bytes, uint32_t flags);
int zj_batch_range_front(zj_batch *b, zj_range_report *report);
int zj_batch_range_resolve(zj_batch *b, const zj_range_report *all, uint32_t nranks, uint32_t *mcu_first, uint32_t *mcu_count, uint32_t *row0, uint32_t *nrows);
int zj_batch_range_decode(zj_batch *b, uint8_t *rgb, size_t stride);

/* ---- workspaces: device memory shared by batches that are decoded one after the other ----
 * A batch made by zj_batch_create owns every byte it uses, the coefficient arena (128 bytes per block)
 * included, so a job too large for one batch would have to allocate and free per sub-batch.  A workspace
 * holds one scratch range -- everything a decode call produces on the way to the pixels: unstuffed streams,
 * bitmaps, window records, block positions, DC values, quantised coefficients -- that all its batches
 * share, and `nslots` plan ranges (scans, descriptors, tables, status words), one per batch that is alive at
 * the same time.  zj_batch_create_in allocates nothing: it uploads through pinned staging on a copy stream of
 * the workspace's own, so planning and uploading batch k+1 overlaps the decode of batch k, and the first
 * decode call of a batch makes the decode stream wait for its upload.
 * Rules: batches of one workspace decode on the context's stream, one after the other (any order); the
 * coefficients of a batch (zj_batch_qcoef, zj_batch_export_coef) are valid until the next decode call of
 * any batch of the workspace; a slot may be used again when its previous batch has been destroyed. */
typedef struct zj_workspace zj_workspace;
/* Upper bounds of the plan and scratch bytes a batch of these images takes (host only, no GPU needed). */
int zj_workspace_estimate(const zj_image *imgs, size_t n, size_t *plan_bytes, size_t *scratch_bytes);
zj_workspace *zj_workspace_create(zj_ctx *ctx, size_t scratch_bytes, size_t plan_bytes, uint32_t nslots);
void zj_workspace_destroy(zj_workspace *ws); /* after its batches */
/* As zj_batch_create, in plan slot `slot`.  NULL when the batch does not fit (zj_last_error says which range). */
zj_batch *zj_batch_create_in(zj_workspace *ws, uint32_t slot, const zj_image *imgs, size_t n, uint32_t flags);
/* Bytes the batch took from its plan slot and from the shared scratch (0, 0 for zj_batch_create). */
int zj_batch_memory(zj_batch *b, size_t *plan_bytes, size_t *scratch_bytes);

/* The hot path: stage 1 (unstuff, self-synchronising parallel Huffman decode, DC prefix scan)
 * -> quantised coefficients in the batch's arena; stage 2+3 (dequantise, 8x8 IDCT, chroma
 * upsample, YCbCr->RGB, store) -> rgb[i].  rgb[i] are DEVICE pointers (NULL entries skip stage
 * 2+3 for that image); stride[i] in bytes (NULL or 0 = 3*width).  Asynchronous on the context's
 * stream; allocates nothing. */
int zj_batch_decode(zj_batch *b, uint8_t *const *rgb, const size_t *stride);

/* The two halves of zj_batch_decode on their own: the entropy passes (the SOS loop, src/decoder.rs:366-390, up
 * to the coefficients) on the context's stream, and IDCT + colour conversion (`to_rgb`, src/decoder.rs:282-320,
 * called at :644-646) on `stream`
 * -- NULL for the context's, or another stream of the caller's on the same device, which is made to wait
 * for the entropy passes.  The entropy passes keep the arithmetic units busy and the pixel pass the memory
 * system, so a caller with several batches gains by giving the pixel passes a stream of their own: the
 * pixel pass of one batch then runs beside the entropy passes of the next.  With a stream of its own the
 * pixels are ready when THAT stream reaches the point behind the call; zj_batch_finish and zj_batch_destroy
 * wait for it as well, and the batch's next entropy pass is ordered behind it.  When the scratch belongs
 * to a workspace, the next entropy pass of ANY batch of that workspace is ordered behind these pixels (it
 * would overwrite the coefficients they are made from): to overlap, alternate between two workspaces. */
int zj_batch_decode_entropy(zj_batch *b);
int zj_batch_decode_pixels(zj_batch *b, uint8_t *const *rgb, const size_t *stride, void *stream);

/* Wait for the stream, then copy out the per-image status words (n entries).  If the device
 * reported that subsequence synchronisation had not converged, this call reruns the decode with
 * the exhaustive fix-up loop before returning (never observed on real streams). */
int zj_batch_finish(zj_batch *b, uint32_t *status);

/* Evidence that decode calls stayed on the fast path, counted over the batch's life:
 *   counters[0]  times a fix-up lane ran through a whole chunk of subsequences without meeting a record
 *                that agreed with its state (the condition that sends zj_batch_finish into its loop)
 *   counters[1]  blocks that did not end where the position list says the next one begins
 *   counters[2]  zj_batch_finish calls that ran the exhaustive fix-up loop
 * All three are 0 for every stream seen so far; waits for the stream. */
int zj_batch_counters(zj_batch *b, uint32_t counters[3]);

/* Device pointer to image i's quantised coefficients inside the arena (valid until destroy).  The decode call keeps
 * the DC values in an array of their own (the arena's coefficient 0 holds the DC difference of src/decoder.rs:198 until
 * someone asks): the first call after a decode call queues one small kernel on the context's stream that writes them
 * into the arena; what the pointer shows is complete once that stream has been waited for. */
const int16_t *zj_batch_qcoef(zj_batch *b, size_t i);
/* Copy image i's coefficients out: zig-zag quantised and/or natural-order dequantised. */
int zj_batch_export_coef(zj_batch *b, size_t i, int16_t *qcoef, int16_t *dqcoef, int mem);

/* The entropy-coded segments (restart intervals, the defined extension for FFDD/RSTn; one segment
 * when the image has no restart interval) the planner cut image i into: ranges[2k], ranges[2k+1] =
 * begin and end byte offset of segment k relative to img->scan, RSTn markers excluded.  Writes at
 * most `cap` pairs, returns the number of segments (0 for an unsupported image). */
size_t zj_batch_segments(zj_batch *b, size_t i, uint64_t *ranges, size_t cap);

/* Per-kernel timing: when enabled, zj_batch_decode brackets every kernel with events on the
 * context's stream; zj_batch_kernel_times waits and returns milliseconds of the last decode call
 * in the order of zj_kernel_name(0..ZJ_NUM_KERNELS-1). */
#define ZJ_NUM_KERNELS 9
int zj_batch_set_profiling(zj_batch *b, int enable);
int zj_batch_kernel_times(zj_batch *b, float *ms /* ZJ_NUM_KERNELS */);
const char *zj_kernel_name(int k);

/* Byte counts of the batch for roofline accounting: scan bytes as stored (S), coefficient bytes
 * (C = blocks * 128), RGB bytes (P = 3 * w * h), summed over the supported images. */
int zj_batch_bytes(zj_batch *b, uint64_t *scan_bytes, uint64_t *coef_bytes, uint64_t *rgb_bytes, uint64_t *pixels);

/* One-shot convenience with the reference's call shape (parse -> decode -> pixels):
 * plan + decode + finish + copy-out + destroy.  outs[i].mem selects host or device outputs. */
int zj_decode_batch(zj_ctx *ctx, const zj_image *imgs, size_t n, const zj_output *outs, uint32_t *status);

/* zj_decode_batch for callers that cannot allocate device memory -- tight serving loops, code that runs between
 * the captures of HIP graphs, harnesses that forbid hipMalloc.  It has no counterpart in the reference.  The call
 * is synchronous (it waits for the stream and copies the status words back), so it cannot itself be captured into
 * a graph.  Same work as
 * zj_decode_batch with ZJ_MEM_DEVICE pixel outputs (rgb[i] device pointers, NULL = skip the pixels of
 * image i; stride[i] in bytes, NULL or 0 = 3 * width), but every device buffer the decode needs -- uploaded
 * scans, tables, unstuffed streams, bitmaps, window records, block positions, quantised coefficients --
 * lies in one fixed array of ZJ_NOALLOC_MAX_BYTES that the library holds on each GPU it runs on until the
 * process ends.  The call makes no hipMalloc / hipFree / hipHostMalloc and creates no stream or event.
 * Inputs that need more than the array return ZJ_E_NOMEM (zj_last_error names the limit); as a guide, one
 * 1920x1080 4:4:4 image takes about 19 MB, one 3840x2160 4:2:0 image about 40 MB.  All calls on one GPU
 * share the array: calls that overlap in time on one GPU (from several contexts or threads) race. */
#define ZJ_NOALLOC_MAX_BYTES (64u << 20)
int zj_decode_batch_noalloc(zj_ctx *ctx, const zj_image *imgs, size_t n, uint8_t *const *rgb, const size_t *stride, uint32_t *status);

#ifdef __cplusplus
}
#endif
#endif /* ZJ_H */
