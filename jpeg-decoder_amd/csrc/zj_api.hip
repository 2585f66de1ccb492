// zj_api.hip -- the C ABI of include/zj.h on top of the kernels: context, batch planning (all
// host-side layout decisions, table construction and uploads), the decode call that launches the
// kernels, result/status retrieval.  The decode call itself allocates nothing.
//
// Where a batch's device memory comes from:
//   * zj_batch_create          every buffer is a hipMalloc of its own, freed by zj_batch_destroy
//   * zj_batch_create_in       plan memory from one slot of a workspace, scratch from the workspace's
//                              shared scratch range: no allocation per batch, the coefficient arena and all
//                              intermediates are reused by the next batch
//   * zj_decode_batch_noalloc  plan and scratch from a file-scope __device__ array: no allocation at all
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/zj.h"
#include "zj_dev.h"
#include "zj_hostutil.h"

#define ZJ_NONE 0xFFFFFFFFu

struct zj_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    std::string err;
    int err_code; // ZJ_E_* behind the last NULL returned by zj_batch_create / zj_batch_create_in / zj_workspace_create
};

// Device memory a batch takes its buffers from instead of allocating them: a bump allocator over a range
// the batch does not own.  Offsets advance in multiples of 256 bytes.
struct ZjPool {
    uint8_t *base;
    size_t cap, off;
};

struct zj_workspace {
    zj_ctx *ctx;
    uint8_t *scratch; // shared by every batch of the workspace
    size_t scratch_bytes;
    uint8_t *plan; // nslots ranges of plan_bytes each
    size_t plan_bytes;
    uint32_t nslots;
    hipStream_t copy_stream;  // uploads and the marker search of zj_batch_create_in run here
    uint8_t *stage[2];        // pinned staging chunks for the scan upload
    hipEvent_t stage_free[2]; // recorded behind the copy that reads stage[k]
    bool stage_used[2];
    size_t stage_bytes;
    // recorded behind the last pixel pass that a batch of this workspace ran on a stream other than the
    // context's: the next entropy pass on the shared scratch waits for it
    hipEvent_t pix_ev;
    bool pix_ev_valid, pix_pending;
    // one live batch per plan slot; the workspace outlives its batches (a destroy while batches are alive is deferred to
    // the last of them)
    std::vector<zj_batch *> slot_batch;
    zj_batch *scratch_owner; // the batch whose decode call wrote the shared scratch last: only its coefficients are there
    uint32_t live;
    bool dying;
    // per slot: pinned staging for the output pointer and pitch tables of the slot's batch (grown on demand), and the
    // event behind the copies that read them
    struct SlotTables {
        uint8_t **h_rgb;
        size_t *h_stride;
        uint32_t *h_flags;
        size_t cap;
        hipEvent_t copied;
        bool ev_valid, ev_pending;
    };
    std::vector<SlotTables> tables;
};
static void workspace_free(zj_workspace *ws);

struct zj_batch {
    zj_ctx *ctx;
    size_t n;
    std::vector<ZjImgDev> imgs;
    std::vector<ZjSegDev> segs;
    std::vector<uint32_t> unsupported; // per image: 1 = rejected at plan time
    std::vector<uint64_t> scan_at;     // per image: where its scan begins in the raw arena
    ZjParams P;
    ZjLaunch L;
    std::vector<void *> allocs;
    // Plan memory (scans, descriptors, tables, status words) belongs to one batch for its whole life; scratch
    // (everything a decode call produces on the way to the pixels, the coefficient arena included) may be
    // shared by batches that decode one after the other on one stream.  Either comes from hipMalloc (flag
    // false) or from a range the batch does not own.
    bool plan_pooled, scratch_pooled;
    ZjPool plan_pool, scratch_pool;
    zj_workspace *ws;  // upload stream and staging of zj_batch_create_in, else NULL
    uint32_t ws_slot;
    hipStream_t up;    // stream of the plan-time uploads and kernels
    hipEvent_t ready;  // recorded on the workspace's copy stream behind the uploads
    bool ready_valid, ready_waited;
    // staging for the per-call pointer tables: pinned, or pageable when the batch may not allocate
    bool pinned;
    uint8_t **h_rgb;
    size_t *h_stride;
    uint8_t **d_rgb;
    size_t *d_stride;
    uint32_t *h_flags;        // 4 words
    hipEvent_t tables_copied; // recorded behind the copies that read h_rgb / h_stride
    bool tables_ev_valid, tables_ev_pending;
    // zj_batch_decode_entropy / zj_batch_decode_pixels: the two halves of a decode call, the second possibly on
    // another stream; created on first use of such a stream
    hipEvent_t entropy_done, pixels_done;
    bool entropy_launched;
    bool dc_patched; // coefficient 0 of the arena's blocks has been filled in from the DC array since the last decode call
    bool split_ev_valid, pixels_pending; // pixels_pending: pixels_done was recorded on a stream other than the context's
    uint32_t slow_path_calls; // zj_batch_finish calls that had to run the exhaustive fix-up loop
    uint64_t scan_bytes, coef_bytes, rgb_bytes, pixels;
    // a part of a stream that is split over several GPUs (zj_batch_create_range), else range.on == false
    struct {
        bool on;
        uint32_t rank, nranks;
        uint64_t x_bits, y_bits;  // where this part's ownership begins and ends, in bits of its own unstuffed stream
        bool last;                // the part that ends with the stream
        int stage;                // 0 planned, 1 front done (report filled), 2 resolved
        zj_range_report rep;
        uint32_t m_first_local;   // position-list MCU of the first MCU owned
        uint32_t g_first, g_end;  // position-list block range of the MCUs owned
        uint32_t before[3];       // DC prefix per component in front of the first MCU owned (in this part's own scan)
        uint32_t off[3];          // what turns that scan's values into predicted DC values
        uint32_t per_wg, cls, strips_x; // launch geometry to redo once the MCUs owned are known
    } range;
    bool profiling;
    hipEvent_t ev[ZJ_NUM_KERNELS + 1];
    bool ev_valid;
    bool decoded;
};

static bool hip_ok(zj_ctx *c, hipError_t e, const char *what)
{
    if (e == hipSuccess) return true;
    if (c) { c->err = std::string(what) + ": " + hipGetErrorString(e); c->err_code = (e == hipErrorOutOfMemory) ? ZJ_E_NOMEM : ZJ_E_DEVICE; }
    return false;
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---------------------------------------------------------------------------------- the no-allocation scratch
// Plan and scratch memory of zj_decode_batch_noalloc.  The array's name is a host-side handle, not a device
// address: every use on the host goes through the pointer hipGetSymbolAddress returned for the device in
// question (each GPU holds its own copy of the array at its own address).
static __device__ __align__(256) uint8_t zj_noalloc_scratch[ZJ_NOALLOC_MAX_BYTES];
#define ZJ_MAX_DEVICES 64
static uint8_t *g_noalloc_ptr[ZJ_MAX_DEVICES];
static std::mutex g_noalloc_mutex;

// Address of the array on `device`, which must be the current device; looked up once per device.
static uint8_t *noalloc_ptr(zj_ctx *c, int device)
{
    if (device < 0 || device >= ZJ_MAX_DEVICES) return nullptr;
    std::lock_guard<std::mutex> lock(g_noalloc_mutex);
    if (!g_noalloc_ptr[device]) {
        void *p = nullptr;
        if (!hip_ok(c, hipGetSymbolAddress(&p, HIP_SYMBOL(zj_noalloc_scratch)), "hipGetSymbolAddress")) return nullptr;
        g_noalloc_ptr[device] = (uint8_t *)p;
    }
    return g_noalloc_ptr[device];
}

extern "C" int zj_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" zj_ctx *zj_create(int device, void *stream)
{
    int n = zj_device_count();
    if (device < 0 || device >= n) return nullptr;
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    zj_ctx *c = new zj_ctx();
    c->device = device;
    c->err_code = ZJ_OK;
    c->own_stream = (stream == nullptr);
    if (stream) c->stream = (hipStream_t)stream;
    else if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return nullptr; }
    // the no-allocation entry point's memory on this device: its address is looked up here, at set-up, so
    // that the entry point itself only reads a table
    if (!noalloc_ptr(c, device)) {
        if (c->own_stream) (void)hipStreamDestroy(c->stream);
        delete c;
        return nullptr;
    }
    return c;
}

extern "C" void zj_destroy(zj_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" const char *zj_last_error(zj_ctx *c) { return c ? c->err.c_str() : "no context"; }

static const char *k_names[ZJ_NUM_KERNELS] = {"k0_unstuff", "k1a_sync", "k1b_fix",   "k1s_scan", "k1p_positions",
                                              "k1d_dcscan", "k1w_write", "k2_idct_rgb", "total"};
extern "C" const char *zj_kernel_name(int k) { return (k >= 0 && k < ZJ_NUM_KERNELS) ? k_names[k] : ""; }

enum { ZJ_MEM_PLAN = 0, ZJ_MEM_SCRATCH = 1 };

static void *pool_take(ZjPool &p, size_t bytes)
{
    const size_t o = align_up(p.off, 256);
    if (o > p.cap || bytes > p.cap - o) return nullptr;
    p.off = o + bytes;
    return p.base + o;
}

template <typename T>
static T *dev_alloc(zj_batch *b, size_t count, bool zero = false, int kind = ZJ_MEM_PLAN)
{
    void *p = nullptr;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = 16;
    const bool pooled = kind == ZJ_MEM_SCRATCH ? b->scratch_pooled : b->plan_pooled;
    if (pooled) {
        ZjPool &pool = kind == ZJ_MEM_SCRATCH ? b->scratch_pool : b->plan_pool;
        p = pool_take(pool, bytes);
        if (!p) {
            char msg[200];
            snprintf(msg, sizeof msg, "batch does not fit: %s memory of %zu bytes is exhausted (%zu in use, %zu more wanted)",
                     kind == ZJ_MEM_SCRATCH ? "scratch" : "plan", pool.cap, pool.off, bytes);
            b->ctx->err = msg;
            b->ctx->err_code = ZJ_E_NOMEM;
            return nullptr;
        }
        // Shared scratch is not cleared here: another batch may be using it on the decode stream right now.
        // Nothing in it has to start at zero (the kernels write every word they later read).
        if (kind == ZJ_MEM_SCRATCH) zero = false;
    } else {
        if (!hip_ok(b->ctx, hipMalloc(&p, bytes), "hipMalloc")) return nullptr;
        b->allocs.push_back(p);
    }
    if (zero && !hip_ok(b->ctx, hipMemsetAsync(p, 0, bytes, b->up), "hipMemset")) return nullptr;
    return (T *)p;
}

template <typename T>
static T *dev_upload(zj_batch *b, const std::vector<T> &v)
{
    T *p = dev_alloc<T>(b, v.size());
    if (!p) return nullptr;
    // synchronous with respect to the host vector (pageable memory is staged before the call returns)
    if (!v.empty() && !hip_ok(b->ctx, hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, b->up), "hipMemcpy H2D")) return nullptr;
    return p;
}

// Restart markers of every image that has a restart interval, searched on the device in the uploaded
// scans.  mark_first[i] = index of image i in the lists (ZJ_NONE: no restart interval); per listed image
// the number of markers found and where its offsets begin in `offs` (at most one per restart interval of
// the frame is kept).  The device buffers used live only for the duration of the call.
static bool find_markers(zj_batch *b, const zj_image *images, size_t n, const uint8_t *raw, const std::vector<size_t> &raw_off,
                         std::vector<uint32_t> &mark_first, std::vector<uint32_t> &count, std::vector<uint32_t> &outfirst,
                         std::vector<uint32_t> &offs)
{
    zj_ctx *ctx = b->ctx;
    std::vector<ZjMarkImg> mi;
    std::vector<uint32_t> tile_img;
    uint64_t nout = 0;
    for (size_t i = 0; i < n; i++) {
        const zj_image &im = images[i];
        if (!zj_image_supported(&im) || im.restart_interval == 0) continue;
        ZjMarkImg d;
        memset(&d, 0, sizeof d);
        d.raw_off = raw_off[i];
        d.scan_len = (uint32_t)im.scan_len;
        d.first_tile = (uint32_t)tile_img.size();
        d.ntiles = (uint32_t)((im.scan_len + ZJ_MARK_TILE - 1) / ZJ_MARK_TILE);
        d.out_first = (uint32_t)nout;
        d.cap = (zj_image_mcus(&im) + im.restart_interval - 1) / im.restart_interval;
        nout += d.cap;
        if (nout > 0xF0000000ull || tile_img.size() + d.ntiles > 0xF0000000ull) { ctx->err = "batch too large for 32-bit indices"; ctx->err_code = ZJ_E_INVALID; return false; }
        mark_first[i] = (uint32_t)mi.size();
        for (uint32_t t = 0; t < d.ntiles; t++) tile_img.push_back((uint32_t)mi.size());
        outfirst.push_back(d.out_first);
        mi.push_back(d);
    }
    count.assign(mi.size(), 0);
    offs.assign((size_t)nout, 0);
    if (mi.empty() || tile_img.empty()) return true;
    ZjMarkParams M;
    memset(&M, 0, sizeof M);
    M.raw = raw;
    M.nimgs = (uint32_t)mi.size();
    M.ntiles = (uint32_t)tile_img.size();
    void *d_imgs = nullptr, *d_tile_img = nullptr, *d_tile_count = nullptr, *d_img_count = nullptr, *d_offs = nullptr;
    hipStream_t st = b->up;
    const size_t sizes[5] = {mi.size() * sizeof(ZjMarkImg), tile_img.size() * 4, tile_img.size() * 4, mi.size() * 4, (size_t)(nout + 1) * 4};
    void **ptrs[5] = {&d_imgs, &d_tile_img, &d_tile_count, &d_img_count, &d_offs};
    bool ok = true;
    const size_t pool_mark = b->plan_pool.off; // pooled: the five buffers are taken from the plan range and given back below
    for (int k = 0; ok && k < 5; k++) {
        if (b->plan_pooled) {
            *ptrs[k] = pool_take(b->plan_pool, sizes[k]);
            if (!*ptrs[k]) { ctx->err = "batch does not fit: plan memory is exhausted by the restart-marker search"; ctx->err_code = ZJ_E_NOMEM; ok = false; }
        } else {
            ok = hip_ok(ctx, hipMalloc(ptrs[k], sizes[k]), "hipMalloc");
        }
    }
    if (ok) ok = hip_ok(ctx, hipMemcpyAsync(d_imgs, mi.data(), mi.size() * sizeof(ZjMarkImg), hipMemcpyHostToDevice, st), "H2D marker images") &&
                 hip_ok(ctx, hipMemcpyAsync(d_tile_img, tile_img.data(), tile_img.size() * 4, hipMemcpyHostToDevice, st), "H2D marker tiles");
    if (ok) {
        M.imgs = (const ZjMarkImg *)d_imgs;
        M.tile_img = (const uint32_t *)d_tile_img;
        M.tile_count = (uint32_t *)d_tile_count;
        M.img_count = (uint32_t *)d_img_count;
        M.offs = (uint32_t *)d_offs;
        zj_launch_markers(M, st);
        ok = hip_ok(ctx, hipGetLastError(), "marker kernels") &&
             hip_ok(ctx, hipMemcpyAsync(count.data(), d_img_count, mi.size() * 4, hipMemcpyDeviceToHost, st), "D2H marker counts") &&
             (nout == 0 || hip_ok(ctx, hipMemcpyAsync(offs.data(), d_offs, (size_t)nout * 4, hipMemcpyDeviceToHost, st), "D2H marker offsets")) &&
             hip_ok(ctx, hipStreamSynchronize(st), "sync after marker search");
    }
    if (b->plan_pooled) b->plan_pool.off = pool_mark;
    else { (void)hipFree(d_imgs); (void)hipFree(d_tile_img); (void)hipFree(d_tile_count); (void)hipFree(d_img_count); (void)hipFree(d_offs); }
    return ok;
}

extern "C" void zj_batch_destroy(zj_batch *b)
{
    if (!b) return;
    (void)hipSetDevice(b->ctx->device);
    (void)hipStreamSynchronize(b->ctx->stream);
    if (b->pixels_pending) (void)hipEventSynchronize(b->pixels_done);
    if (b->ws) (void)hipStreamSynchronize(b->ws->copy_stream);
    for (void *p : b->allocs) (void)hipFree(p);
    if (b->ws) {
        // the tables belong to the workspace's slot
    } else if (b->pinned) {
        if (b->h_rgb) (void)hipHostFree(b->h_rgb);
        if (b->h_stride) (void)hipHostFree(b->h_stride);
        if (b->h_flags) (void)hipHostFree(b->h_flags);
    } else {
        free(b->h_rgb); free(b->h_stride); free(b->h_flags);
    }
    if (b->ev_valid)
        for (auto &e : b->ev) (void)hipEventDestroy(e);
    if (b->ready_valid) (void)hipEventDestroy(b->ready);
    if (b->split_ev_valid) { (void)hipEventDestroy(b->entropy_done); (void)hipEventDestroy(b->pixels_done); }
    if (b->tables_ev_valid && !b->ws) (void)hipEventDestroy(b->tables_copied);
    zj_workspace *ws = b->ws;
    if (ws) {
        if (b->ws_slot < ws->slot_batch.size() && ws->slot_batch[b->ws_slot] == b) {
            ws->slot_batch[b->ws_slot] = nullptr;
            ws->tables[b->ws_slot].ev_pending = b->tables_ev_pending;
        }
        if (ws->live) ws->live--;
        if (ws->scratch_owner == b) ws->scratch_owner = nullptr;
    }
    delete b;
    if (ws && ws->dying && ws->live == 0) workspace_free(ws);
}

// Upload of the scans through the workspace's two pinned chunks: the raw arena is built chunk by chunk in
// pinned memory (scans at their 16-byte aligned offsets, zeros between them) and each chunk goes over in one
// asynchronous copy while the host fills the other.
static bool upload_scans_staged(zj_batch *b, const zj_image *images, size_t n, uint8_t *raw, const std::vector<size_t> &raw_off, size_t raw_total)
{
    zj_workspace *ws = b->ws;
    zj_ctx *ctx = b->ctx;
    const size_t CH = ws->stage_bytes;
    size_t i = 0; // first image that may still have bytes at or behind the chunk's start
    int k = 0;
    for (size_t c0 = 0; c0 < raw_total; c0 += CH, k ^= 1) {
        const size_t c1 = std::min(raw_total, c0 + CH);
        if (ws->stage_used[k] && !hip_ok(ctx, hipEventSynchronize(ws->stage_free[k]), "staging wait")) return false;
        uint8_t *dst = ws->stage[k];
        size_t pos = c0; // everything of the chunk before `pos` is filled
        while (i < n && pos < c1) {
            const zj_image &im = images[i];
            if (!zj_image_supported(&im) || im.scan_len == 0) { i++; continue; }
            const size_t s0 = raw_off[i], s1 = raw_off[i] + im.scan_len;
            if (s1 <= pos) { i++; continue; }
            if (s0 >= c1) break;
            if (s0 > pos) { memset(dst + (pos - c0), 0, s0 - pos); pos = s0; }
            const size_t e = std::min(s1, c1);
            memcpy(dst + (pos - c0), im.scan + (pos - s0), e - pos);
            pos = e;
            if (e == s1) i++;
        }
        if (pos < c1) memset(dst + (pos - c0), 0, c1 - pos);
        if (!hip_ok(ctx, hipMemcpyAsync(raw + c0, dst, c1 - c0, hipMemcpyHostToDevice, b->up), "H2D scans")) return false;
        if (!hip_ok(ctx, hipEventRecord(ws->stage_free[k], b->up), "event record")) return false;
        ws->stage_used[k] = true;
    }
    return hip_ok(ctx, hipMemsetAsync(raw + raw_total, 0, 1040, b->up), "hipMemset");
}

enum { ZJ_ALLOC_OWN = 0, ZJ_ALLOC_WORKSPACE, ZJ_ALLOC_NONE };

// The planner behind zj_batch_create, zj_batch_create_in and zj_decode_batch_noalloc.
// ZJ_ALLOC_NONE: plan and scratch memory both come out of the range [fixed_base, fixed_base + fixed_cap).
static zj_batch *batch_create_impl(zj_ctx *ctx, const zj_image *images, size_t n, uint32_t flags, int mode, zj_workspace *ws, uint32_t slot,
                                   uint8_t *fixed_base, size_t fixed_cap, const uint32_t *win_row0 = nullptr, const uint32_t *win_nrows = nullptr,
                                   const uint64_t *seg_range = nullptr)
{
    if (!ctx) return nullptr;
    ctx->err_code = ZJ_E_INVALID;
    if (!images && n) { ctx->err = "no images"; return nullptr; }
    if (!hip_ok(ctx, hipSetDevice(ctx->device), "hipSetDevice")) return nullptr;
    zj_batch *b = new zj_batch();
    b->ctx = ctx;
    b->n = n;
    b->h_rgb = nullptr; b->h_stride = nullptr; b->h_flags = nullptr;
    b->ev_valid = false; b->profiling = false; b->decoded = false;
    b->ready_valid = false; b->ready_waited = false; b->tables_ev_valid = false; b->tables_ev_pending = false;
    b->slow_path_calls = 0;
    b->entropy_launched = false; b->split_ev_valid = false; b->pixels_pending = false; b->dc_patched = false;
    b->scan_bytes = b->coef_bytes = b->rgb_bytes = b->pixels = 0;
    memset(&b->range, 0, sizeof b->range);
    b->ws = nullptr;
    b->up = ctx->stream;
    b->plan_pooled = b->scratch_pooled = false;
    b->pinned = true;
    memset(&b->plan_pool, 0, sizeof b->plan_pool);
    memset(&b->scratch_pool, 0, sizeof b->scratch_pool);
    if (mode == ZJ_ALLOC_WORKSPACE) {
        b->ws = ws;
        b->ws_slot = slot;
        ws->live++; // from here on zj_batch_destroy gives the slot back
        ws->slot_batch[slot] = b;
        b->up = ws->copy_stream;
        b->plan_pooled = b->scratch_pooled = true;
        b->plan_pool = ZjPool{ws->plan + (size_t)slot * ws->plan_bytes, ws->plan_bytes, 0};
        b->scratch_pool = ZjPool{ws->scratch, ws->scratch_bytes, 0};
        b->pinned = false;
    } else if (mode == ZJ_ALLOC_NONE) {
        // one range for both kinds: plan memory first, scratch behind it (the scratch pool is set up below,
        // when the plan part is complete)
        b->plan_pooled = b->scratch_pooled = true;
        b->plan_pool = ZjPool{fixed_base, fixed_cap, 0};
        b->pinned = false;
    }
    memset(&b->P, 0, sizeof b->P);
    memset(&b->L, 0, sizeof b->L);

    const uint32_t l2_cap = (flags & ZJ_FLAG_NO_SECOND_LEVEL) ? 0u : ZJ_L2_SIZE;
    uint32_t subbits = flags & 0xFFFFFFu;
    const bool subbits_forced = subbits != 0;
    if (subbits == 0) {
        // Enough subsequences to give every SIMD of the GPU a few wavefronts, and no more: the longer a
        // subsequence, the smaller the share of the synchronisation rounds.
        uint64_t bits = 0;
        for (size_t i = 0; i < n; i++) bits += (uint64_t)images[i].scan_len * 8u;
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
        const uint64_t lanes = (uint64_t)cus * 16u * 64u;
        // One window for a very small input (what counts there is the chain of window steps of a lone wavefront, and a
        // second window of its own is one more step for every lane; measured: 0.167 -> 0.154 ms for a 600x200 image).
        // Not beyond: a chunk of 64 subsequences has to be longer than the distance a wrong guess survives, or the fix pass
        // runs through whole chunks and the decode call is repeated with the exhaustive loop -- 32 Kbit chunks are enough
        // for photographs, not for large flat images; and from sixteen 1080p images on, longer subsequences do less work twice.
        subbits = bits < (4u << 20) ? ZJ_WIN_BITS : 1024u;
        while (subbits < 131072u && (uint64_t)subbits * lanes < bits) subbits *= 2u; // (the 4 096-image batch: one chunk per 1080p image at 128 Kbit, and nothing left for the fix pass to do: -0.2 ms)
    }
    if (subbits < ZJ_WIN_BITS || subbits > (1u << 20)) { ctx->err = "subsequence bits must be in 512..1048576"; zj_batch_destroy(b); return nullptr; }

    b->imgs.resize(n);
    b->unsupported.assign(n, 0);
    std::vector<uint32_t> chunk_seg, blk256_img, k1w_img, ptile_seg, tile_seg, small_seg, dtile_seg, dsmall_seg;
    uint64_t dc_longest = 0;  // MCUs of the longest segment
    size_t unstuff_tiles = 0; // tiles of the unstuff pass's 128-thread form over all segments: one look-back word each
    uint64_t nwin_total = 0;
    std::vector<uint32_t> k2_grp[ZJ_K2_CLASSES], k2_img[ZJ_K2_CLASSES]; // stage-2 workgroups per sampling layout
    std::vector<uint16_t> luts;
    std::vector<ZjCanon> canons;
    std::vector<uint16_t> qts;
    std::vector<uint8_t> set_partner; // [table set][DC id] AC table id that follows, or 0xFF
    std::map<std::string, uint32_t> lut_ids, qt_ids;
    std::vector<size_t> raw_off(n, 0);
    size_t raw_total = 0, ust_total = 0;
    uint64_t nsub_total = 0, nblk_total = 0, nstrip_total = 0;
    uint32_t prev_lut_set = ZJ_NONE;
    std::vector<size_t> offs;

    // The scans go to the device first: the restart markers are searched there.
    uint64_t mcus_total = 0;
    for (size_t i = 0; i < n; i++) {
        if (!zj_image_supported(&images[i])) continue;
        if (images[i].scan_len >= 0xF0000000ull) { ctx->err = "scan of 4 GiB or more"; ctx->err_code = ZJ_E_INVALID; zj_batch_destroy(b); return nullptr; } // the kernels count bytes inside a segment in 32 bits
        mcus_total += zj_image_mcus(&images[i]);
        raw_off[i] = raw_total;
        raw_total += align_up(images[i].scan_len, 16) + 16;
    }
    b->scan_at.assign(raw_off.begin(), raw_off.end());
    // Tiles per workgroup of the unstuff pass (a workgroup takes tiles of its segment until none is left, asking for the next
    // one's bytes while it works on this one): up to ZJ_UNSTUFF_REP when the batch has several times more tiles than the GPU
    // holds at once (ten per CU), one when it has fewer -- a single image wants every tile in flight at the same moment.
    const uint32_t unstuff_rep = (uint32_t)std::min<uint64_t>(ZJ_UNSTUFF_REP, std::max<uint64_t>(1, raw_total / ZJ_UNSTUFF_TILE / 2560u));
    uint8_t *raw = dev_alloc<uint8_t>(b, raw_total + 1040, false);
    if (!raw) { zj_batch_destroy(b); return nullptr; }
    {
        // the entropy-coded bytes as stored, each image at a 16-byte aligned offset
        bool up;
        if (b->ws) {
            up = upload_scans_staged(b, images, n, raw, raw_off, raw_total);
        } else {
            up = hip_ok(ctx, hipMemsetAsync(raw, 0, raw_total + 1040, b->up), "hipMemset");
            for (size_t i = 0; up && i < n; i++) {
                if (!zj_image_supported(&images[i]) || images[i].scan_len == 0) continue;
                up = hip_ok(ctx, hipMemcpyAsync(raw + raw_off[i], images[i].scan, images[i].scan_len, hipMemcpyHostToDevice, b->up), "hipMemcpy scan");
            }
            if (up) up = hip_ok(ctx, hipStreamSynchronize(b->up), "sync after upload");
        }
        if (!up) { zj_batch_destroy(b); return nullptr; }
    }
    std::vector<uint32_t> mark_first(n, ZJ_NONE), mark_count, mark_outfirst, mark_offs; // device-found restart offsets per image
    if (!(flags & ZJ_FLAG_HOST_MARKER_SCAN) && !find_markers(b, images, n, raw, raw_off, mark_first, mark_count, mark_outfirst, mark_offs)) {
        zj_batch_destroy(b);
        return nullptr;
    }

    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
    // MCUs per wavefront of the write pass: 32 (64 for 4:4:4) times a power of two up to 8, as long as the batch
    // still gives every CU several sets of wavefronts (measured: x8 is 7 % faster than x1 on a large batch).
    uint32_t k1w_mult = 1;
    while (k1w_mult < 16u && mcus_total / (64u * k1w_mult) >= (uint64_t)cus * 12u * 4u) k1w_mult *= 2u;
    // Strips per workgroup of the IDCT/colour kernel: 4, doubled up to 16 while the batch keeps every CU
    // supplied with workgroups (a strip is about 16 MCUs; measured: 16 is 4 % faster than 4 on a large batch).
    uint32_t k2_ns = ZJ_K2_NS;
    while (k2_ns < 16u && mcus_total / 16u / (k2_ns * 2u) >= (uint64_t)cus * 16u) k2_ns *= 2u;
#ifdef ZJ_K2_NS_FORCE // experiments only (tools/exp/build_variant.sh)
    k2_ns = ZJ_K2_NS_FORCE;
#endif
    b->P.k2_ns = k2_ns;
    b->P.dc_in_write = (win_row0 == nullptr && seg_range == nullptr) ? 1u : 0u; // (a window of rows or a part of a stream: the DC pass of its own)
    for (size_t i = 0; i < n; i++) {
        const zj_image &im = images[i];
        ZjImgDev &d = b->imgs[i];
        memset(&d, 0, sizeof d);
        d.first_seg = (uint32_t)b->segs.size();
        d.first_block = (uint32_t)nblk_total;
        d.first_strip = (uint32_t)nstrip_total;
        if (!zj_image_supported(&im)) { b->unsupported[i] = 1; continue; }
        d.flags = 1;
        d.w = im.width; d.h = im.height;
        d.hs = im.hsamp[0]; d.vs = im.vsamp[0];
        const uint32_t ncomp = im.ncomp;
        d.bpm = ncomp == 1 ? 1u : d.hs * d.vs + 2;
        d.mcux = (d.w + 8 * d.hs - 1) / (8 * d.hs);
        d.mcuy = (d.h + 8 * d.vs - 1) / (8 * d.vs);
        const uint64_t nmcu = (uint64_t)d.mcux * d.mcuy;
        d.nblocks = (uint32_t)(nmcu * d.bpm);
        // the MCU rows that are wanted (zj_batch_create_rows); by default all of them
        d.row0 = 0;
        d.nrows = d.mcuy;
        if (win_row0 && win_nrows && win_nrows[i]) {
            if (win_row0[i] >= d.mcuy || win_nrows[i] > d.mcuy - win_row0[i]) {
                ctx->err = "rows outside the image";
                ctx->err_code = ZJ_E_INVALID;
                zj_batch_destroy(b);
                return nullptr;
            }
            d.row0 = win_row0[i];
            d.nrows = win_nrows[i];
        }
        const uint64_t wmcu = (uint64_t)d.mcux * d.nrows; // MCUs of those rows
        d.mcu_lo = d.row0 * d.mcux;
        d.mcu_hi = (uint32_t)std::min<uint64_t>(nmcu, (uint64_t)(d.row0 + d.nrows) * d.mcux);
        d.mcu_shift = 0;
        for (uint32_t k = 0; k < d.bpm; k++) {
            const uint32_t c = k < d.hs * d.vs ? 0u : (k == d.hs * d.vs ? 1u : 2u);
            d.compsel |= c << (2 * k);
            d.dcsel |= (uint32_t)im.td[c] << k;
            d.acsel |= (uint32_t)im.ta[c] << k;
        }
        // table set (deduplicated by content over the batch)
        {
            const zj_huff *t4[4] = {im.ht[0][1], im.ht[1][1], im.ht[0][0], im.ht[1][0]}; // DC0 DC1 AC0 AC1
            std::string key;
            for (int t = 0; t < 4; t++) {
                if (t4[t]) key.append((const char *)t4[t], sizeof(zj_huff));
                else key.append(sizeof(zj_huff), '\0');
            }
            // Which AC table follows each DC table in this image's blocks (0, 1, or 0xFF: none or not the same for all
            // components that use it): the DC pair entries hold the block's first AC symbol, so the pairing is part of
            // what a table set is.
            uint8_t partner[2] = {0xFF, 0xFF};
            for (int dcid = 0; dcid < 2; dcid++) {
                int a = -1;
                for (uint32_t c = 0; c < ncomp; c++)
                    if (im.td[c] == dcid) a = (a == -1 || a == im.ta[c]) ? im.ta[c] : -2;
                if (a >= 0) partner[dcid] = (uint8_t)a;
            }
            key.append((const char *)partner, 2);
            auto it = lut_ids.find(key);
            if (it == lut_ids.end()) {
                const uint32_t id = (uint32_t)(luts.size() / ZJ_SET_ENTRIES);
                luts.resize(luts.size() + ZJ_SET_ENTRIES);
                canons.resize(canons.size() + 4);
                bool ok = true;
                for (int t = 0; t < 4; t++) {
                    uint16_t *tbl = &luts[(size_t)id * ZJ_SET_ENTRIES + (size_t)t * ZJ_TBL_ENTRIES];
                    if (t4[t]) ok = zj_build_table(t4[t], tbl, &canons[(size_t)id * 4 + t], l2_cap) && ok;
                    else { memset(tbl, 0, sizeof(uint16_t) * ZJ_TBL_ENTRIES); memset(&canons[(size_t)id * 4 + t], 0, sizeof(ZjCanon)); }
                }
                if (ok) { set_partner.push_back(partner[0]); set_partner.push_back(partner[1]); }
                else { luts.resize((size_t)id * ZJ_SET_ENTRIES); canons.resize((size_t)id * 4); }
                it = lut_ids.emplace(key, ok ? id : ZJ_NONE).first;
            }
            if (it->second == ZJ_NONE) { b->unsupported[i] = 1; d.flags = 0; d.nblocks = 0; continue; }
            d.lut_set = it->second;
            for (uint32_t t = 0; t < 4u; t++) {
                const uint16_t *tbl = &luts[(size_t)d.lut_set * ZJ_SET_ENTRIES + (size_t)t * ZJ_TBL_ENTRIES];
                for (uint32_t e = 0; e < ZJ_L1_SIZE; e++)
                    if ((tbl[e] & 0x8000u) && ((tbl[e] >> 12) & 7u) == 0) d.flags |= 2u;
            }
        }
        for (int c = 0; c < 3; c++) {
            std::string key((const char *)im.qt[im.tq[c < (int)ncomp ? c : 0]], 64);
            auto it = qt_ids.find(key);
            if (it == qt_ids.end()) {
                it = qt_ids.emplace(key, (uint32_t)qts.size()).first;
                for (int z = 0; z < 64; z++) qts.push_back((uint16_t)(uint8_t)key[z]);
            }
            d.qt_off[c] = it->second;
        }
        // segments: one per restart interval (src/bitstream.rs:41-47 ends the reference's stream at
        // the first RSTn; here every interval is an independent segment)
        const uint32_t ri = im.restart_interval;
        const uint64_t nseg = ri ? (nmcu + ri - 1) / ri : 1;
        d.ri_blocks = ri ? ri * d.bpm : 0;
        d.nseg = (uint32_t)nseg;
        size_t nint = 1;
        offs.assign(1, 0);
        if (ri && mark_first[i] != ZJ_NONE) {
            // found on the device: entry k is the byte behind the k-th RSTn, as zj_restart_offsets lists them
            const uint32_t mi = mark_first[i];
            const size_t have = std::min<size_t>(mark_count[mi], (size_t)nseg);
            offs.resize(1 + have);
            for (size_t k = 0; k < have; k++) offs[1 + k] = mark_offs[mark_outfirst[mi] + k];
            nint = 1 + have;
        } else if (ri) {
            offs.resize((size_t)nseg + 1);
            nint = zj_restart_offsets(&im, offs.data(), offs.size());
            if (nint > offs.size()) nint = offs.size();
        }
        // The workgroups of the sync and fix passes take four chunks each and load one table set: an image
        // starts a new workgroup only when its table set differs from its predecessor's.  (Padding every
        // image to whole workgroups left half of all wavefront slots empty for images of two chunks.)
        if (d.lut_set != prev_lut_set) {
            while (chunk_seg.size() % 4u) chunk_seg.push_back(ZJ_NONE);
            nsub_total = align_up(nsub_total, 256);
            prev_lut_set = d.lut_set;
        }
        d.first_sub = (uint32_t)nsub_total;
        uint32_t sub_in_img = 0;
        uint32_t open_lanes = 0, open_used = 0; // the packed chunk under construction: its group size, lanes taken
        // Bits per lane wanted in the groups of short segments.  A lone wavefront needs about 20 us per
        // window whatever the group size, and the walk into the neighbours costs about nine windows more, so
        // groups of few long subsequences finish as early as many short ones and do a fraction of the work.
        const uint32_t small_target = subbits_forced ? subbits : std::max(subbits, 16384u);
        for (uint64_t k = 0; k < nseg; k++) {
            ZjSegDev s;
            memset(&s, 0, sizeof s);
            size_t beg = im.scan_len, end = im.scan_len;
            if (k < nint) {
                beg = offs[(size_t)k];
                end = (k + 1 < nint) ? offs[(size_t)k + 1] - 2 : im.scan_len;
            }
            if (seg_range) { beg = (size_t)seg_range[0]; end = (size_t)seg_range[1]; } // a part of the stream (one image, no restart interval)
            s.raw_begin = raw_off[i] + beg;
            s.raw_end = raw_off[i] + end;
            const size_t rl = end - beg;
            s.out_off = ust_total;
            s.out_cap = (uint32_t)align_up(align_up(rl, 16) + 96, 128); // segments start on 128-byte lines: the sync pass loads whole lines
            ust_total += s.out_cap;
            s.img = (uint32_t)i;
            zj_seg_layout((uint64_t)rl * 8u, subbits, small_target, &s.subbits, &s.nsub, &s.lanes);
            s.nc = s.subbits / ZJ_WIN_BITS;
            s.first_win = (uint32_t)nwin_total;
            if (s.lanes == 64u) {
                // whole chunks of its own
                sub_in_img = (uint32_t)align_up(sub_in_img, 64);
                open_lanes = 0;
                s.first_sub = d.first_sub + sub_in_img;
                const uint32_t nchunk = (s.nsub + 63) / 64;
                nwin_total += (uint64_t)nchunk * 64u * s.nc;
                for (uint32_t c = 0; c < nchunk; c++) chunk_seg.push_back((uint32_t)b->segs.size());
                sub_in_img += nchunk * 64;
            } else {
                // a group of lanes in a packed chunk, which holds groups of one size only
                if (open_lanes != s.lanes || open_used + s.lanes > 64u) {
                    sub_in_img = (uint32_t)align_up(sub_in_img, 64);
                    chunk_seg.push_back((uint32_t)b->segs.size());
                    b->L.npacked++;
                    open_lanes = s.lanes;
                    open_used = 0;
                }
                s.first_sub = d.first_sub + sub_in_img;
                nwin_total += (uint64_t)s.lanes * s.nc;
                sub_in_img += s.lanes;
                open_used += s.lanes;
            }
            if (s.raw_end - (s.raw_begin & ~15ull) <= ZJ_UNSTUFF_SMALL) { // also an empty segment, which still publishes its payload length
                s.first_tile = 0;
                s.ntiles = 1;
                small_seg.push_back((uint32_t)b->segs.size());
            } else {
                s.first_tile = (uint32_t)unstuff_tiles;
                s.ntiles = (uint32_t)((s.raw_end - (s.raw_begin & ~15ull) + ZJ_UNSTUFF_TILE - 1u) / ZJ_UNSTUFF_TILE);
                unstuff_tiles += s.ntiles;
                // a workgroup of the unstuff pass takes tiles of its segment until none is left
                for (uint32_t t = 0; t < s.ntiles; t += unstuff_rep) tile_seg.push_back((uint32_t)b->segs.size());
            }
            s.first_ptile = (uint32_t)ptile_seg.size();
            for (uint64_t w = 0; w < (uint64_t)s.nsub * s.nc; w += 256u) ptile_seg.push_back((uint32_t)b->segs.size());
            const uint64_t m0 = ri ? k * ri : 0;
            const uint64_t mc = ri ? std::min<uint64_t>(ri, nmcu - m0) : nmcu;
            s.first_block = d.first_block + (uint32_t)(m0 * d.bpm);
            s.nblocks = (uint32_t)(mc * d.bpm);
            if (mc <= 64u) {
                s.first_dtile = 0;
                s.ndtiles = 1;
                dsmall_seg.push_back((uint32_t)b->segs.size());
            } else {
                s.first_dtile = (uint32_t)dtile_seg.size();
                s.ndtiles = (uint32_t)((mc + 255u) / 256u);
                dc_longest = std::max<uint64_t>(dc_longest, mc);
                if (s.ndtiles > 1u) b->L.ndmulti++;
                for (uint32_t t = 0; t < s.ndtiles; t++) dtile_seg.push_back((uint32_t)b->segs.size());
            }
            b->segs.push_back(s);
        }
        sub_in_img = (uint32_t)align_up(sub_in_img, 64);
        d.nsub = sub_in_img;
        nsub_total += d.nsub;
        const uint32_t nb_pad = (uint32_t)align_up(d.nblocks, 256);
        for (uint32_t g = 0; g < nb_pad / 256; g++) blk256_img.push_back((uint32_t)i);
        nblk_total += nb_pad;
        {
            d.k1w_mcus = (d.hs * d.vs == 1 ? 64u : 32u) * k1w_mult;
            const uint32_t per_wg = 4u * d.k1w_mcus; // MCUs per write-pass workgroup
            b->range.per_wg = per_wg;
            d.first_k1w = (uint32_t)k1w_img.size();
            for (uint64_t m = 0; m < wmcu; m += per_wg) k1w_img.push_back((uint32_t)i);
        }
        const uint32_t T = ncomp == 1 ? 96u : 4u * zj_k2_wave_mcus(d.hs, d.vs); // MCUs per strip of stage 2
        d.strips_x = (d.mcux + T - 1) / T;
        const uint32_t ns = (d.strips_x * d.nrows + k2_ns - 1) / k2_ns; // workgroups of stage 2
        const int cls = ncomp == 1 ? 4 : (d.hs == 2 && d.vs == 2) ? 0 : (d.hs == 1 && d.vs == 1) ? 1 : (d.hs == 2) ? 2 : 3;
        for (uint32_t s = 0; s < ns; s++) { k2_grp[cls].push_back(s); k2_img[cls].push_back((uint32_t)i); }
        b->range.cls = (uint32_t)cls;
        b->range.strips_x = d.strips_x;
        nstrip_total += ns;
        b->scan_bytes += im.scan_len;
        b->coef_bytes += wmcu * d.bpm * 128;
        const uint32_t mh = 8u * d.vs, wy0 = d.row0 * mh, wh = std::min<uint32_t>(d.nrows * mh, d.h - wy0); // pixel rows produced
        b->rgb_bytes += (uint64_t)d.w * wh * 3;
        b->pixels += (uint64_t)d.w * wh;
        if (nsub_total > 0xF0000000ull || nblk_total > 0xF0000000ull || nwin_total > 0xF0000000ull || ust_total > 0xF0000000ull) {
            ctx->err = "batch too large for 32-bit indices";
            ctx->err_code = ZJ_E_INVALID;
            zj_batch_destroy(b);
            return nullptr;
        }
    }

    while (chunk_seg.size() % 4u) chunk_seg.push_back(ZJ_NONE); // whole workgroups
    ZjParams &P = b->P;
    bool ok = true;
    P.segs = dev_upload(b, b->segs);
    P.chunk_seg = dev_upload(b, chunk_seg);
    P.blk256_img = dev_upload(b, blk256_img);
    P.k1w_img = dev_upload(b, k1w_img);
    P.ptile_seg = dev_upload(b, ptile_seg);
    const size_t nbig = tile_seg.size(); // tiles of the 128-thread form; the short segments' entries follow them
    tile_seg.insert(tile_seg.end(), small_seg.begin(), small_seg.end());
    P.tile_seg = dev_upload(b, tile_seg);
    const size_t nbigd = dtile_seg.size();
    dtile_seg.insert(dtile_seg.end(), dsmall_seg.begin(), dsmall_seg.end());
    P.dtile_seg = dev_upload(b, dtile_seg);
    std::vector<uint32_t> strip_grp_all, strip_img_all;
    for (int c = 0; c < ZJ_K2_CLASSES; c++) {
        b->L.k2_first[c] = (uint32_t)strip_grp_all.size();
        b->L.k2_count[c] = (uint32_t)k2_grp[c].size();
        strip_grp_all.insert(strip_grp_all.end(), k2_grp[c].begin(), k2_grp[c].end());
        strip_img_all.insert(strip_img_all.end(), k2_img[c].begin(), k2_img[c].end());
    }
    P.strip_img = dev_upload(b, strip_grp_all);
    P.strip_img_img = dev_upload(b, strip_img_all);
    std::vector<uint16_t> wluts(luts.size());
    for (size_t t = 0; t < luts.size() / ZJ_TBL_ENTRIES; t++)
        zj_build_walk_table(&luts[t * ZJ_TBL_ENTRIES], (t & 3u) < 2u, &wluts[t * ZJ_TBL_ENTRIES]);
    std::vector<uint32_t> ptabs(luts.size() / ZJ_TBL_ENTRIES * ZJ_PT_ENTRIES);
    std::vector<uint8_t> pair_needs_canon(luts.size() / ZJ_SET_ENTRIES, 0);
    for (size_t t = 0; t < luts.size() / ZJ_TBL_ENTRIES; t++) {
        uint32_t *pt = &ptabs[t * ZJ_PT_ENTRIES];
        const uint8_t pa = (t & 3u) < 2u ? set_partner[(t / 4u) * 2u + (t & 1u)] : 0xFF;
        (void)zj_build_pair_table(&canons[t], (t & 3u) < 2u, pt, pa != 0xFF ? &canons[(t / 4u) * 4u + 2u + pa] : nullptr);
        for (uint32_t e = 0; e < (uint32_t)ZJ_PL1_SIZE; e++) {
            if (!(pt[e] & 0x80000000u)) continue;
            if (l2_cap == 0) pt[e] = 0x80000000u; // the testing aid that drops the second level applies here too
            if (((pt[e] >> 24) & 7u) == 0) pair_needs_canon[t / 4] = 1;
        }
    }
    for (size_t i = 0; i < n; i++)
        if (b->imgs[i].flags && pair_needs_canon[b->imgs[i].lut_set]) b->imgs[i].flags |= 4u;
    // the AC tables (third and fourth of a set) as write entries: in the direct form when their long codes fit it
    // (and the testing aid that drops the second level is off), else in the layout of the 16-bit tables
    const size_t nsets = luts.size() / ZJ_SET_ENTRIES;
    std::vector<uint32_t> wtabs(nsets * 2u * ZJ_WT_ENTRIES, 0u), wt_info(nsets * 2u + 2u, ZJ_WT_TWO_LEVEL);
    for (size_t set = 0; set < nsets; set++)
        for (uint32_t t = 0; t < 2u; t++) {
            uint32_t *wt = &wtabs[(set * 2u + t) * ZJ_WT_ENTRIES];
            const ZjCanon &cn = canons[set * 4u + 2u + t];
#ifndef ZJ_V_NODIRECT // experiments only
            if (l2_cap != 0 && cn.nsym != 0 && zj_build_direct_table(&cn, wt, &wt_info[set * 2u + t])) continue;
#endif
            wt_info[set * 2u + t] = ZJ_WT_TWO_LEVEL;
            for (uint32_t e = 0; e < ZJ_TBL_ENTRIES; e++) wt[e] = zj_write_entry(luts[set * ZJ_SET_ENTRIES + (2u + t) * ZJ_TBL_ENTRIES + e]);
        }
    // behind the entries of AC table t: the first level of DC table t, one byte per entry (zj_core.h, ZJ_WT_DIRECT)
    for (size_t set = 0; set < nsets; set++)
        for (uint32_t t = 0; t < 2u; t++) {
            uint8_t *dc8 = reinterpret_cast<uint8_t *>(&wtabs[(set * 2u + t) * ZJ_WT_ENTRIES + ZJ_WT_DIRECT]);
            for (uint32_t i = 0; i < ZJ_L1_SIZE; i++) {
                const uint32_t e = luts[set * ZJ_SET_ENTRIES + t * ZJ_TBL_ENTRIES + i];
                const uint32_t len = (e >> 8) & 31u, sz = e & 255u;
                dc8[i] = ((e & 0x8000u) || len == 0 || len > 15u || sz > 15u) ? 0u : (uint8_t)(len | (sz << 4));
            }
        }
    P.luts = dev_upload(b, luts);
    P.wluts = dev_upload(b, wluts);
    P.ptabs = dev_upload(b, ptabs);
    P.wtabs = dev_upload(b, wtabs);
    P.wt_info = dev_upload(b, wt_info);
    P.imgs = dev_upload(b, b->imgs); // after the table forms are known: the descriptors carry their flags
    P.canons = dev_upload(b, canons);
    P.qts = dev_upload(b, qts);
    P.raw = raw;
    // what every decode call clears, in one block so that it is one fill: flags, status words, the unstuff pass's look-back
    // words and tickets
    const size_t nbig_tiles = unstuff_tiles;
    const size_t status_words = align_up(n ? n : 1, 4);
    const size_t clear_bytes = 16 + status_words * 4 + (nbig_tiles + 1) * 8 + (b->segs.size() + 2) * 4;
    uint8_t *clear_block = dev_alloc<uint8_t>(b, clear_bytes, true);
    P.flags = reinterpret_cast<uint32_t *>(clear_block);
    P.status = clear_block ? reinterpret_cast<uint32_t *>(clear_block + 16) : nullptr;
    P.tile_status = clear_block ? reinterpret_cast<uint64_t *>(clear_block + 16 + status_words * 4) : nullptr;
    P.seg_ticket = clear_block ? reinterpret_cast<uint32_t *>(P.tile_status + nbig_tiles + 1) : nullptr;
    P.counters = dev_alloc<uint32_t>(b, 4, true);
    b->L.clear_bytes = clear_bytes;
    b->d_rgb = dev_alloc<uint8_t *>(b, n, true);
    b->d_stride = dev_alloc<size_t>(b, n, true);
    ok = P.imgs && P.segs && P.chunk_seg && P.blk256_img && P.k1w_img && P.strip_img && P.strip_img_img && P.luts && P.wluts && P.ptabs && P.wtabs && P.wt_info && P.canons && P.qts &&
         P.ptile_seg && P.tile_seg && P.dtile_seg && P.status && P.flags && P.counters && b->d_rgb && b->d_stride;
    if (ok && mode == ZJ_ALLOC_NONE) {
        // the rest of the range is this call's scratch
        const size_t used = align_up(b->plan_pool.off, 256);
        b->scratch_pool = ZjPool{b->plan_pool.base + used, b->plan_pool.cap - used, 0};
        b->plan_pool.cap = used;
    }
    if (ok) {
        const int S = ZJ_MEM_SCRATCH;
        P.dtile_sum = dev_alloc<uint32_t>(b, (nbigd + 1) * 4, true, S);
        P.ust = dev_alloc<uint8_t>(b, ust_total + 1040, true, S);
        P.seg_payload_bits = dev_alloc<uint32_t>(b, b->segs.size(), true, S);
        P.seg_found = dev_alloc<uint32_t>(b, b->segs.size(), true, S);
        P.winfo = dev_alloc<uint32_t>(b, nwin_total, true, S);
        P.wbase = dev_alloc<uint32_t>(b, nwin_total, true, S);
        P.bitmap = dev_alloc<uint8_t>(b, ust_total + 1040, true, S);
        P.blockpos = dev_alloc<uint32_t>(b, nblk_total, true, S);
        P.dcabs = dev_alloc<int16_t>(b, nblk_total, true, S);
        P.dcbits = dev_alloc<uint8_t>(b, nblk_total, true, S);
        P.qcoef = dev_alloc<int16_t>(b, nblk_total * 64, false, S);
        ok = P.dtile_sum && P.ust && P.seg_payload_bits && P.seg_found && P.winfo && P.wbase && P.bitmap && P.blockpos && P.dcabs &&
             P.dcbits && P.qcoef;
    }
    P.rgb = b->d_rgb;
    P.rgb_stride = b->d_stride;
    P.nchunks = (uint32_t)chunk_seg.size();
    P.nseg = (uint32_t)b->segs.size();
    const size_t nn = n ? n : 1;
    if (ok && b->pinned) {
        ok = hip_ok(ctx, hipHostMalloc((void **)&b->h_rgb, nn * sizeof(uint8_t *), hipHostMallocDefault), "hipHostMalloc") &&
             hip_ok(ctx, hipHostMalloc((void **)&b->h_stride, nn * sizeof(size_t), hipHostMallocDefault), "hipHostMalloc") &&
             hip_ok(ctx, hipHostMalloc((void **)&b->h_flags, 16, hipHostMallocDefault), "hipHostMalloc") &&
             hip_ok(ctx, hipEventCreateWithFlags(&b->tables_copied, hipEventDisableTiming), "hipEventCreate");
        b->tables_ev_valid = ok;
    } else if (ok && b->ws) {
        // pinned tables of the plan slot: the copies that read them are asynchronous, and the next decode call rewrites
        // them only behind the slot's event
        zj_workspace::SlotTables &t = ws->tables[slot];
        if (t.ev_pending) { ok = hip_ok(ctx, hipEventSynchronize(t.copied), "event sync"); t.ev_pending = false; }
        if (ok && t.cap < nn) {
            if (t.h_rgb) (void)hipHostFree(t.h_rgb);
            if (t.h_stride) (void)hipHostFree(t.h_stride);
            t.h_rgb = nullptr; t.h_stride = nullptr; t.cap = 0;
            const size_t cap = std::max<size_t>(nn, 256);
            ok = hip_ok(ctx, hipHostMalloc((void **)&t.h_rgb, cap * sizeof(uint8_t *), hipHostMallocDefault), "hipHostMalloc") &&
                 hip_ok(ctx, hipHostMalloc((void **)&t.h_stride, cap * sizeof(size_t), hipHostMallocDefault), "hipHostMalloc");
            if (ok) t.cap = cap;
        }
        if (ok && !t.h_flags) ok = hip_ok(ctx, hipHostMalloc((void **)&t.h_flags, 16, hipHostMallocDefault), "hipHostMalloc");
        if (ok && !t.ev_valid) { ok = hip_ok(ctx, hipEventCreateWithFlags(&t.copied, hipEventDisableTiming), "hipEventCreate"); t.ev_valid = ok; }
        if (ok) {
            b->h_rgb = t.h_rgb; b->h_stride = t.h_stride; b->h_flags = t.h_flags;
            b->tables_copied = t.copied;
            b->tables_ev_valid = true;
        }
    } else if (ok) {
        b->h_rgb = (uint8_t **)calloc(nn, sizeof(uint8_t *));
        b->h_stride = (size_t *)calloc(nn, sizeof(size_t));
        b->h_flags = (uint32_t *)calloc(4, sizeof(uint32_t));
        if (!b->h_rgb || !b->h_stride || !b->h_flags) { ctx->err = "out of host memory"; ctx->err_code = ZJ_E_NOMEM; ok = false; }
    }
    if (ok && b->ws) {
        // the decode stream waits for the uploads once, at the batch's first decode call
        ok = hip_ok(ctx, hipEventCreateWithFlags(&b->ready, hipEventDisableTiming), "hipEventCreate");
        b->ready_valid = ok;
        if (ok) ok = hip_ok(ctx, hipEventRecord(b->ready, b->up), "event record");
    }
    // The tables above were copied out of vectors that die with this call.  Copies from pageable memory are
    // staged before hipMemcpyAsync returns on this runtime, but nothing here depends on that: wait for the
    // upload stream (the decode stream, which may be busy with another batch of the workspace, is not touched).
    if (ok) ok = hip_ok(ctx, hipStreamSynchronize(b->up), "sync after plan upload");
    if (!ok) { zj_batch_destroy(b); return nullptr; }
    b->L.nseg = (uint32_t)b->segs.size();
    b->L.nchunks = P.nchunks;
    b->L.nblk256 = (uint32_t)blk256_img.size();
    b->L.nk1w = (uint32_t)k1w_img.size();
    b->L.nptiles = (uint32_t)ptile_seg.size();
    b->L.ntiles = (uint32_t)nbig;
    b->L.nsmall = (uint32_t)small_seg.size();
    b->L.ndtiles = (uint32_t)nbigd;
    b->L.ndsmall = (uint32_t)dsmall_seg.size();
    {
        // the DC scan by whole segments pays when the segments fill the GPU on their own and are long enough to occupy a
        // workgroup of 1024 lanes; huge segments, or many short ones, are better off in tiles
        const size_t nlong = b->segs.size() - dsmall_seg.size();
        b->L.dc_by_segment = nlong >= 256u && nbigd >= nlong * 8u && dc_longest <= 16384u; // (a longer segment is a chain of more than four steps of one workgroup)
    }
    b->L.nstrips = (uint32_t)strip_grp_all.size();
    b->L.fix_iters = 1;
    ctx->err_code = ZJ_OK;
    return b;
}

extern "C" zj_batch *zj_batch_create(zj_ctx *ctx, const zj_image *images, size_t n, uint32_t flags)
{
    return batch_create_impl(ctx, images, n, flags, ZJ_ALLOC_OWN, nullptr, 0, nullptr, 0);
}

extern "C" zj_batch *zj_batch_create_rows(zj_ctx *ctx, const zj_image *images, size_t n, uint32_t flags, const uint32_t *row0, const uint32_t *nrows)
{
    return batch_create_impl(ctx, images, n, flags, ZJ_ALLOC_OWN, nullptr, 0, nullptr, 0, row0, nrows);
}

// ---------------------------------------------------------------------------------- workspaces
extern "C" int zj_workspace_estimate(const zj_image *images, size_t n, size_t *plan_bytes, size_t *scratch_bytes)
{
    if (!images && n) return ZJ_E_INVALID;
    // upper bounds of what batch_create_impl takes, term by term (every allocation rounded up to 256 bytes)
    uint64_t plan = 1040 + 64 * 256, scr = 2 * 1040 + 64 * 256;
    for (size_t i = 0; i < n; i++) {
        const zj_image &im = images[i];
        plan += sizeof(ZjImgDev) + 4 + 16; // descriptor, status word, output pointer and pitch
        if (!zj_image_supported(&im)) continue;
        const uint64_t len = im.scan_len, M = zj_image_mcus(&im), B = align_up((size_t)zj_image_blocks(&im), 256);
        const uint64_t S = im.restart_interval ? (M + im.restart_interval - 1) / im.restart_interval : 1;
        const uint64_t win = len / 32 + 64 * S + 512;  // at most two windows per 512 bits (rounding per lane), whole chunks
        const uint64_t chunks = win / 64 + S + 8;
        const uint64_t utiles = len / ZJ_UNSTUFF_TILE + 2 * S, ptiles = win / 256 + S, dtiles = M / 256 + S;
        const uint64_t mtiles = len / ZJ_MARK_TILE + 1;
        plan += align_up((size_t)len, 16) + 16 + S * sizeof(ZjSegDev) + 4 * (chunks + B / 256 + (M / 128 + 1) + ptiles + utiles + dtiles) + 8 * (M / 16 + 2);
        plan += 2 * ZJ_SET_ENTRIES * 2 + ZJ_PSET_ENTRIES * 4 + 2 * ZJ_WT_ENTRIES * 4 + 512 + 4 * sizeof(ZjCanon) + 3 * 64 * 2;  // a table set (four forms) and quantisation tables of its own
        plan += sizeof(ZjMarkImg) + 8 * mtiles + 4 + 4 * (S + 1);           // restart-marker search
        plan += 8 * (utiles + S) + 64; // look-back words and tickets of the unstuff pass, cleared with the status words
        scr += 16 * dtiles + 2 * (len + 240 * S) + 8 * S + 8 * win + 7 * B + 128 * B; // a segment's slot: its bytes, 96 of slack, rounded up to 16 and then to a 128-byte line
    }
    if (plan_bytes) *plan_bytes = (size_t)(plan + plan / 64);
    if (scratch_bytes) *scratch_bytes = (size_t)(scr + scr / 64);
    return ZJ_OK;
}

extern "C" void zj_workspace_destroy(zj_workspace *ws)
{
    if (!ws) return;
    if (ws->live) { ws->dying = true; return; } // batches are still alive: the last of them frees the workspace
    workspace_free(ws);
}

static void workspace_free(zj_workspace *ws)
{
    (void)hipSetDevice(ws->ctx->device);
    if (ws->copy_stream) { (void)hipStreamSynchronize(ws->copy_stream); (void)hipStreamDestroy(ws->copy_stream); }
    (void)hipStreamSynchronize(ws->ctx->stream);
    if (ws->pix_ev_valid) {
        if (ws->pix_pending) (void)hipEventSynchronize(ws->pix_ev);
        (void)hipEventDestroy(ws->pix_ev);
    }
    for (int k = 0; k < 2; k++) {
        if (ws->stage[k]) (void)hipHostFree(ws->stage[k]);
        if (ws->stage_free[k]) (void)hipEventDestroy(ws->stage_free[k]);
    }
    for (auto &t : ws->tables) {
        if (t.ev_valid) { if (t.ev_pending) (void)hipEventSynchronize(t.copied); (void)hipEventDestroy(t.copied); }
        if (t.h_rgb) (void)hipHostFree(t.h_rgb);
        if (t.h_stride) (void)hipHostFree(t.h_stride);
        if (t.h_flags) (void)hipHostFree(t.h_flags);
    }
    if (ws->scratch) (void)hipFree(ws->scratch);
    if (ws->plan) (void)hipFree(ws->plan);
    delete ws;
}

extern "C" zj_workspace *zj_workspace_create(zj_ctx *ctx, size_t scratch_bytes, size_t plan_bytes, uint32_t nslots)
{
    if (!ctx) return nullptr;
    ctx->err_code = ZJ_E_INVALID;
    if (nslots == 0 || scratch_bytes == 0 || plan_bytes == 0) { ctx->err = "workspace sizes and slot count must be positive"; return nullptr; }
    if (!hip_ok(ctx, hipSetDevice(ctx->device), "hipSetDevice")) return nullptr;
    zj_workspace *ws = new zj_workspace();
    ws->ctx = ctx;
    ws->scratch = nullptr; ws->plan = nullptr; ws->copy_stream = nullptr;
    ws->stage[0] = ws->stage[1] = nullptr; ws->stage_free[0] = ws->stage_free[1] = nullptr;
    ws->stage_used[0] = ws->stage_used[1] = false;
    ws->pix_ev_valid = ws->pix_pending = false;
    ws->live = 0; ws->dying = false; ws->scratch_owner = nullptr;
    ws->slot_batch.assign(nslots, nullptr);
    ws->tables.assign(nslots, zj_workspace::SlotTables{nullptr, nullptr, nullptr, 0, nullptr, false, false});
    ws->scratch_bytes = align_up(scratch_bytes, 256);
    ws->plan_bytes = align_up(plan_bytes, 256);
    ws->nslots = nslots;
    ws->stage_bytes = std::min<size_t>(32u << 20, align_up(plan_bytes, 4096));
    bool ok = hip_ok(ctx, hipMalloc((void **)&ws->scratch, ws->scratch_bytes), "hipMalloc workspace scratch") &&
              hip_ok(ctx, hipMalloc((void **)&ws->plan, ws->plan_bytes * nslots), "hipMalloc workspace plan slots") &&
              hip_ok(ctx, hipStreamCreateWithFlags(&ws->copy_stream, hipStreamNonBlocking), "hipStreamCreate");
    for (int k = 0; ok && k < 2; k++)
        ok = hip_ok(ctx, hipHostMalloc((void **)&ws->stage[k], ws->stage_bytes, hipHostMallocDefault), "hipHostMalloc staging") &&
             hip_ok(ctx, hipEventCreateWithFlags(&ws->stage_free[k], hipEventDisableTiming), "hipEventCreate");
    if (!ok) { workspace_free(ws); return nullptr; }
    ctx->err_code = ZJ_OK;
    return ws;
}

extern "C" zj_batch *zj_batch_create_in(zj_workspace *ws, uint32_t slot, const zj_image *images, size_t n, uint32_t flags)
{
    if (!ws) return nullptr;
    if (slot >= ws->nslots) { ws->ctx->err = "no such plan slot"; ws->ctx->err_code = ZJ_E_INVALID; return nullptr; }
    if (ws->dying) { ws->ctx->err = "the workspace is being destroyed"; ws->ctx->err_code = ZJ_E_INVALID; return nullptr; }
    if (ws->slot_batch[slot]) { ws->ctx->err = "plan slot is still held by a live batch (destroy it first)"; ws->ctx->err_code = ZJ_E_INVALID; return nullptr; }
    return batch_create_impl(ws->ctx, images, n, flags, ZJ_ALLOC_WORKSPACE, ws, slot, nullptr, 0);
}

extern "C" int zj_batch_memory(zj_batch *b, size_t *plan_bytes, size_t *scratch_bytes)
{
    if (!b) return ZJ_E_INVALID;
    if (plan_bytes) *plan_bytes = b->plan_pooled ? b->plan_pool.off : 0;
    if (scratch_bytes) *scratch_bytes = b->scratch_pooled ? b->scratch_pool.off : 0;
    return ZJ_OK;
}

// The entropy passes (everything up to the coefficients) on the context's stream.
static int decode_entropy(zj_batch *b)
{
    zj_ctx *c = b->ctx;
    if (!hip_ok(c, hipSetDevice(c->device), "hipSetDevice")) return ZJ_E_DEVICE;
    hipStream_t st = c->stream;
    if (b->pixels_pending) {
        // the IDCT of the previous call may still be reading the coefficients on its own stream
        if (!hip_ok(c, hipStreamWaitEvent(st, b->pixels_done, 0), "stream wait")) return ZJ_E_DEVICE;
        b->pixels_pending = false;
    }
    if (b->scratch_pooled && b->ws && b->ws->pix_pending) {
        // so may the IDCT of another batch on the same scratch
        if (!hip_ok(c, hipStreamWaitEvent(st, b->ws->pix_ev, 0), "stream wait")) return ZJ_E_DEVICE;
        b->ws->pix_pending = false;
    }
    if (b->ready_valid && !b->ready_waited) {
        if (!hip_ok(c, hipStreamWaitEvent(st, b->ready, 0), "stream wait")) return ZJ_E_DEVICE;
        b->ready_waited = true;
    }
    b->entropy_launched = true;
    b->dc_patched = false;
    if (b->scratch_pooled && b->ws) b->ws->scratch_owner = b;
    if (b->L.nchunks == 0) { // nothing to decode: the status words are all this call has to clear (otherwise stage 1's one fill does)
        if (b->n && !hip_ok(c, hipMemsetAsync(b->P.status, 0, b->n * sizeof(uint32_t), st), "memset status")) return ZJ_E_DEVICE;
        return ZJ_OK;
    }
    hipEvent_t *ev = nullptr;
    if (b->profiling) {
        if (!b->ev_valid) {
            for (auto &e : b->ev)
                if (!hip_ok(c, hipEventCreate(&e), "hipEventCreate")) return ZJ_E_DEVICE;
            b->ev_valid = true;
        }
        ev = b->ev;
    }
    zj_launch_stage1(b->P, b->L, st, ev);
    if (!hip_ok(c, hipGetLastError(), "kernel launch")) return ZJ_E_DEVICE;
    return ZJ_OK;
}

// IDCT and colour conversion into the caller's buffers, on `ps` (the context's stream, or one of the caller's
// that is made to wait for the entropy passes).
static int decode_pixels(zj_batch *b, uint8_t *const *rgb, const size_t *stride, hipStream_t ps)
{
    zj_ctx *c = b->ctx;
    if (!hip_ok(c, hipSetDevice(c->device), "hipSetDevice")) return ZJ_E_DEVICE;
    hipStream_t st = c->stream;
    if (!b->entropy_launched) { c->err = "zj_batch_decode_pixels before zj_batch_decode_entropy"; return ZJ_E_INVALID; }
    for (size_t i = 0; i < b->n; i++) {
        if (rgb && rgb[i] && !b->unsupported[i] && stride && stride[i] && stride[i] < (size_t)b->imgs[i].w * 3u) {
            c->err = "output pitch is smaller than 3 * width";
            return ZJ_E_INVALID;
        }
    }
    // the copies of the previous call may still be reading the staging tables
    if (b->tables_ev_pending) {
        if (!hip_ok(c, hipEventSynchronize(b->tables_copied), "event sync")) return ZJ_E_DEVICE;
        b->tables_ev_pending = false;
    }
    bool any_rgb = false;
    for (size_t i = 0; i < b->n; i++) {
        b->h_rgb[i] = (rgb && !b->unsupported[i]) ? rgb[i] : nullptr;
        b->h_stride[i] = stride ? stride[i] : 0;
        any_rgb = any_rgb || b->h_rgb[i];
    }
    if (ps != st) {
        if (!b->split_ev_valid) {
            if (!hip_ok(c, hipEventCreateWithFlags(&b->entropy_done, hipEventDisableTiming), "hipEventCreate")) return ZJ_E_DEVICE;
            if (!hip_ok(c, hipEventCreateWithFlags(&b->pixels_done, hipEventDisableTiming), "hipEventCreate")) { (void)hipEventDestroy(b->entropy_done); return ZJ_E_DEVICE; }
            b->split_ev_valid = true;
        }
        if (!hip_ok(c, hipEventRecord(b->entropy_done, st), "event record")) return ZJ_E_DEVICE;
        if (!hip_ok(c, hipStreamWaitEvent(ps, b->entropy_done, 0), "stream wait")) return ZJ_E_DEVICE;
    }
    if (b->n) {
        if (!hip_ok(c, hipMemcpyAsync(b->d_rgb, b->h_rgb, b->n * sizeof(uint8_t *), hipMemcpyHostToDevice, ps), "H2D rgb table")) return ZJ_E_DEVICE;
        if (!hip_ok(c, hipMemcpyAsync(b->d_stride, b->h_stride, b->n * sizeof(size_t), hipMemcpyHostToDevice, ps), "H2D stride table")) return ZJ_E_DEVICE;
        if (b->tables_ev_valid) {
            if (!hip_ok(c, hipEventRecord(b->tables_copied, ps), "event record")) return ZJ_E_DEVICE;
            b->tables_ev_pending = true;
        }
    }
    if (b->L.nchunks) {
        ZjLaunch L2 = b->L;
        if (!any_rgb) { L2.nstrips = 0; for (int k = 0; k < ZJ_K2_CLASSES; k++) L2.k2_count[k] = 0; }
        zj_launch_stage2(b->P, L2, ps, (b->profiling && b->ev_valid) ? b->ev + 8 : nullptr);
        if (!hip_ok(c, hipGetLastError(), "kernel launch")) return ZJ_E_DEVICE;
    }
    if (ps != st) {
        if (!hip_ok(c, hipEventRecord(b->pixels_done, ps), "event record")) return ZJ_E_DEVICE;
        b->pixels_pending = true;
        // scratch shared with other batches: their entropy passes must not start before these pixels are out
        if (b->scratch_pooled && b->ws) {
            zj_workspace *ws = b->ws;
            if (!ws->pix_ev_valid) {
                if (!hip_ok(c, hipEventCreateWithFlags(&ws->pix_ev, hipEventDisableTiming), "hipEventCreate")) return ZJ_E_DEVICE;
                ws->pix_ev_valid = true;
            }
            if (!hip_ok(c, hipEventRecord(ws->pix_ev, ps), "event record")) return ZJ_E_DEVICE;
            ws->pix_pending = true;
        } else if (b->scratch_pooled && !hip_ok(c, hipStreamWaitEvent(st, b->pixels_done, 0), "stream wait")) return ZJ_E_DEVICE;
    }
    b->decoded = true;
    return ZJ_OK;
}

extern "C" int zj_batch_decode(zj_batch *b, uint8_t *const *rgb, const size_t *stride)
{
    if (!b) return ZJ_E_INVALID;
    // the checks of the second half first: a call that is going to be refused launches nothing
    for (size_t i = 0; i < b->n; i++) {
        if (rgb && rgb[i] && !b->unsupported[i] && stride && stride[i] && stride[i] < (size_t)b->imgs[i].w * 3u) {
            b->ctx->err = "output pitch is smaller than 3 * width";
            return ZJ_E_INVALID;
        }
    }
    const int rc = decode_entropy(b);
    return rc != ZJ_OK ? rc : decode_pixels(b, rgb, stride, b->ctx->stream);
}

extern "C" int zj_batch_decode_entropy(zj_batch *b)
{
    return b ? decode_entropy(b) : ZJ_E_INVALID;
}

extern "C" int zj_batch_decode_pixels(zj_batch *b, uint8_t *const *rgb, const size_t *stride, void *stream)
{
    if (!b) return ZJ_E_INVALID;
    return decode_pixels(b, rgb, stride, stream ? (hipStream_t)stream : b->ctx->stream);
}

extern "C" int zj_batch_finish(zj_batch *b, uint32_t *status)
{
    if (!b) return ZJ_E_INVALID;
    zj_ctx *c = b->ctx;
    if (!hip_ok(c, hipSetDevice(c->device), "hipSetDevice")) return ZJ_E_DEVICE;
    hipStream_t st = c->stream;
    if (!hip_ok(c, hipStreamSynchronize(st), "stream sync")) return ZJ_E_DEVICE;
    if (b->pixels_pending) {
        if (!hip_ok(c, hipEventSynchronize(b->pixels_done), "event sync")) return ZJ_E_DEVICE;
        b->pixels_pending = false;
    }
    if (b->decoded && b->L.nchunks) {
        if (!hip_ok(c, hipMemcpy(b->h_flags, b->P.flags, 8, hipMemcpyDeviceToHost), "D2H flags")) return ZJ_E_DEVICE;
        if ((b->h_flags[0] || b->h_flags[1]) && b->range.on) {
            c->err = "part of a split stream: the walks did not converge, decode the stream whole";
            return ZJ_E_CHAIN;
        }
        if (b->h_flags[0] || b->h_flags[1]) {
            // Subsequence records did not converge in the fixed number of fix-up passes: iterate to a
            // fixed point (each pass completes at least one more chunk), then redo everything behind.
            b->slow_path_calls++;
            if (b->scratch_pooled) {
                // another batch may have used the shared scratch since: start over from the stuffed scans -- behind the
                // pixel pass of whichever batch of the workspace decoded last (it may still be reading the coefficients
                // this rerun overwrites; its coefficients are gone afterwards, as after any decode call on the workspace)
                if (b->ws && b->ws->pix_pending) {
                    if (!hip_ok(c, hipStreamWaitEvent(st, b->ws->pix_ev, 0), "stream wait")) return ZJ_E_DEVICE;
                    b->ws->pix_pending = false;
                }
                zj_launch_stage1(b->P, b->L, st, nullptr);
                if (!hip_ok(c, hipStreamSynchronize(st), "stream sync")) return ZJ_E_DEVICE;
            }
            uint32_t passes = 0;
            for (uint32_t it = 0; it <= b->L.nchunks; it++) {
                if (!hip_ok(c, hipMemsetAsync(b->P.flags, 0, 8, st), "memset flags")) return ZJ_E_DEVICE;
                zj_launch_fix_only(b->P, b->L, st);
                passes++;
                if (!hip_ok(c, hipMemcpyAsync(b->h_flags, b->P.flags, 8, hipMemcpyDeviceToHost, st), "D2H flags")) return ZJ_E_DEVICE;
                if (!hip_ok(c, hipStreamSynchronize(st), "stream sync")) return ZJ_E_DEVICE;
                if (!b->h_flags[0]) break;
            }
            // the batch's later decode calls run that many fix passes up front (a pass that finds nothing to do is a few
            // microseconds), so that a stream of this kind pays for the loop once
            b->L.fix_iters = (int)std::min<uint32_t>(16u, std::max<uint32_t>((uint32_t)b->L.fix_iters, b->L.fix_iters + passes));
            if (!hip_ok(c, hipMemsetAsync(b->P.status, 0, b->n * sizeof(uint32_t), st), "memset status")) return ZJ_E_DEVICE;
            zj_launch_stage1_tail(b->P, b->L, st);
            bool any_rgb = false;
            for (size_t i = 0; i < b->n; i++) any_rgb = any_rgb || b->h_rgb[i];
            ZjLaunch L2 = b->L;
            if (!any_rgb) { L2.nstrips = 0; for (int c = 0; c < ZJ_K2_CLASSES; c++) L2.k2_count[c] = 0; }
            zj_launch_stage2(b->P, L2, st, nullptr);
            if (!hip_ok(c, hipStreamSynchronize(st), "stream sync")) return ZJ_E_DEVICE;
            if (!hip_ok(c, hipMemcpy(b->h_flags, b->P.flags, 8, hipMemcpyDeviceToHost), "D2H flags")) return ZJ_E_DEVICE;
            if (b->h_flags[1]) { c->err = "subsequence records inconsistent after exhaustive fix-up"; return ZJ_E_DEVICE; }
        }
    }
    if (status && b->n) {
        if (!hip_ok(c, hipMemcpy(status, b->P.status, b->n * sizeof(uint32_t), hipMemcpyDeviceToHost), "D2H status")) return ZJ_E_DEVICE;
        for (size_t i = 0; i < b->n; i++)
            if (b->unsupported[i]) status[i] = ZJ_ST_UNSUPPORTED;
    }
    return ZJ_OK;
}

extern "C" int zj_batch_counters(zj_batch *b, uint32_t *counters)
{
    if (!b || !counters) return ZJ_E_INVALID;
    zj_ctx *c = b->ctx;
    if (!hip_ok(c, hipSetDevice(c->device), "hipSetDevice")) return ZJ_E_DEVICE;
    if (!hip_ok(c, hipStreamSynchronize(c->stream), "stream sync")) return ZJ_E_DEVICE;
    uint32_t f[2] = {0, 0};
    if (b->P.counters && !hip_ok(c, hipMemcpy(f, b->P.counters, 8, hipMemcpyDeviceToHost), "D2H counters")) return ZJ_E_DEVICE;
    counters[0] = f[0];
    counters[1] = f[1];
    counters[2] = b->slow_path_calls;
    return ZJ_OK;
}

// The write pass leaves the DC difference in coefficient 0 of every block and the IDCT/colour kernel takes the value from
// the DC array; a caller that wants the arena itself gets the values written into it first (once per decode call).
static bool ensure_dc_patched(zj_batch *b)
{
    if (b->dc_patched || !b->entropy_launched || b->L.nchunks == 0) return true;
    if (b->scratch_pooled && b->ws && b->ws->scratch_owner != b) return true; // another batch has decoded into the shared scratch since: nothing of this one is left to complete
    zj_ctx *c = b->ctx;
    if (!hip_ok(c, hipSetDevice(c->device), "hipSetDevice")) return false;
    zj_launch_patch_dc(b->P, b->L.nblk256 * 256u, c->stream);
    if (!hip_ok(c, hipGetLastError(), "DC patch launch")) return false;
    b->dc_patched = true;
    return true;
}

extern "C" const int16_t *zj_batch_qcoef(zj_batch *b, size_t i)
{
    if (!b || i >= b->n || b->unsupported[i]) return nullptr;
    if (!ensure_dc_patched(b)) return nullptr;
    return b->P.qcoef + (size_t)b->imgs[i].first_block * 64;
}

extern "C" int zj_batch_export_coef(zj_batch *b, size_t i, int16_t *qcoef, int16_t *dqcoef, int mem)
{
    if (!b || i >= b->n || b->unsupported[i]) return ZJ_E_INVALID;
    zj_ctx *c = b->ctx;
    if (!hip_ok(c, hipSetDevice(c->device), "hipSetDevice")) return ZJ_E_DEVICE;
    hipStream_t st = c->stream;
    if (!ensure_dc_patched(b)) return ZJ_E_DEVICE;
    const size_t bytes = (size_t)b->imgs[i].nblocks * 128;
    const hipMemcpyKind kind = mem == ZJ_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (qcoef && !hip_ok(c, hipMemcpyAsync(qcoef, zj_batch_qcoef(b, i), bytes, kind, st), "copy qcoef")) return ZJ_E_DEVICE;
    if (dqcoef) {
        int16_t *tmp = dqcoef;
        if (mem != ZJ_MEM_DEVICE && !hip_ok(c, hipMalloc((void **)&tmp, bytes ? bytes : 16), "hipMalloc")) return ZJ_E_NOMEM;
        zj_launch_dequant(b->P, (uint32_t)i, b->imgs[i].nblocks, tmp, st);
        bool ok = hip_ok(c, hipGetLastError(), "dequant launch");
        if (ok && mem != ZJ_MEM_DEVICE) ok = hip_ok(c, hipMemcpyAsync(dqcoef, tmp, bytes, hipMemcpyDeviceToHost, st), "copy dqcoef");
        ok = hip_ok(c, hipStreamSynchronize(st), "stream sync") && ok;
        if (mem != ZJ_MEM_DEVICE) (void)hipFree(tmp);
        if (!ok) return ZJ_E_DEVICE;
    }
    if (!hip_ok(c, hipStreamSynchronize(st), "stream sync")) return ZJ_E_DEVICE;
    // coefficients are produced for the MCUs [mcu_lo, mcu_hi) only (all of them unless the batch was made with a window of
    // rows or is a part of a split stream): what the arena holds elsewhere is not this image's, so the export is zero there
    const ZjImgDev &d = b->imgs[i];
    const size_t lo = (size_t)d.mcu_lo * d.bpm * 128u, hi = std::min(bytes, (size_t)d.mcu_hi * d.bpm * 128u);
    for (int16_t *dst : {qcoef, dqcoef}) {
        if (!dst) continue;
        uint8_t *p8 = reinterpret_cast<uint8_t *>(dst);
        if (mem == ZJ_MEM_DEVICE) {
            if (lo && !hip_ok(c, hipMemsetAsync(p8, 0, std::min(lo, bytes), st), "memset")) return ZJ_E_DEVICE;
            if (hi < bytes && !hip_ok(c, hipMemsetAsync(p8 + hi, 0, bytes - hi, st), "memset")) return ZJ_E_DEVICE;
        } else {
            if (lo) memset(p8, 0, std::min(lo, bytes));
            if (hi < bytes) memset(p8 + hi, 0, bytes - hi);
        }
    }
    if (mem == ZJ_MEM_DEVICE && !hip_ok(c, hipStreamSynchronize(st), "stream sync")) return ZJ_E_DEVICE;
    return ZJ_OK;
}

extern "C" size_t zj_batch_segments(zj_batch *b, size_t i, uint64_t *ranges, size_t cap)
{
    if (!b || i >= b->n || b->unsupported[i]) return 0;
    const ZjImgDev &d = b->imgs[i];
    const uint64_t base = b->scan_at[i];
    for (uint32_t k = 0; k < d.nseg && k < cap; k++) {
        ranges[2 * (size_t)k] = b->segs[d.first_seg + k].raw_begin - base;
        ranges[2 * (size_t)k + 1] = b->segs[d.first_seg + k].raw_end - base;
    }
    return d.nseg;
}

extern "C" int zj_batch_set_profiling(zj_batch *b, int enable)
{
    if (!b) return ZJ_E_INVALID;
    b->profiling = enable != 0;
    return ZJ_OK;
}

extern "C" int zj_batch_kernel_times(zj_batch *b, float *ms)
{
    if (!b || !ms || !b->ev_valid) return ZJ_E_INVALID;
    zj_ctx *c = b->ctx;
    if (!hip_ok(c, hipEventSynchronize(b->ev[8]), "event sync")) return ZJ_E_DEVICE;
    for (int k = 0; k < 8; k++)
        if (!hip_ok(c, hipEventElapsedTime(&ms[k], b->ev[k], b->ev[k + 1]), "event elapsed")) return ZJ_E_DEVICE;
    if (b->P.dc_in_write) std::swap(ms[5], ms[6]); // the write pass came before the DC scan: the names follow the other order
    if (!hip_ok(c, hipEventElapsedTime(&ms[8], b->ev[0], b->ev[8]), "event elapsed")) return ZJ_E_DEVICE;
    return ZJ_OK;
}

extern "C" int zj_batch_bytes(zj_batch *b, uint64_t *scan_bytes, uint64_t *coef_bytes, uint64_t *rgb_bytes, uint64_t *pixels)
{
    if (!b) return ZJ_E_INVALID;
    if (scan_bytes) *scan_bytes = b->scan_bytes;
    if (coef_bytes) *coef_bytes = b->coef_bytes;
    if (rgb_bytes) *rgb_bytes = b->rgb_bytes;
    if (pixels) *pixels = b->pixels;
    return ZJ_OK;
}

extern "C" int zj_decode_batch(zj_ctx *ctx, const zj_image *imgs, size_t n, const zj_output *outs, uint32_t *status)
{
    if (!ctx || (!imgs && n) || (!outs && n)) return ZJ_E_INVALID;
    zj_batch *b = zj_batch_create(ctx, imgs, n, 0);
    if (!b) return ctx->err_code ? ctx->err_code : ZJ_E_DEVICE;
    std::vector<uint8_t *> rgb(n, nullptr);
    std::vector<size_t> stride(n, 0);
    std::vector<void *> tmp;
    int rc = ZJ_OK;
    for (size_t i = 0; i < n && rc == ZJ_OK; i++) {
        if (b->unsupported[i] || !outs[i].rgb) continue;
        if (outs[i].mem == ZJ_MEM_DEVICE) { rgb[i] = outs[i].rgb; stride[i] = outs[i].rgb_stride; }
        else {
            void *p = nullptr;
            const size_t row = align_up((size_t)imgs[i].width * 3, 16);
            if (outs[i].rgb_stride && outs[i].rgb_stride < (size_t)imgs[i].width * 3) { ctx->err = "output pitch is smaller than 3 * width"; rc = ZJ_E_INVALID; break; }
            if (!hip_ok(ctx, hipMalloc(&p, row * imgs[i].height + 16), "hipMalloc rgb")) { rc = ZJ_E_NOMEM; break; }
            tmp.push_back(p);
            rgb[i] = (uint8_t *)p;
            stride[i] = row;
        }
    }
    if (rc == ZJ_OK) rc = zj_batch_decode(b, rgb.data(), stride.data());
    if (rc == ZJ_OK) rc = zj_batch_finish(b, status);
    for (size_t i = 0; i < n && rc == ZJ_OK; i++) {
        if (b->unsupported[i]) continue;
        if (outs[i].rgb && outs[i].mem != ZJ_MEM_DEVICE) {
            const size_t dstride = outs[i].rgb_stride ? outs[i].rgb_stride : (size_t)imgs[i].width * 3;
            if (!hip_ok(ctx, hipMemcpy2D(outs[i].rgb, dstride, rgb[i], stride[i], (size_t)imgs[i].width * 3, imgs[i].height, hipMemcpyDeviceToHost), "D2H rgb"))
                rc = ZJ_E_DEVICE;
        }
        if (rc == ZJ_OK && (outs[i].qcoef || outs[i].dqcoef)) rc = zj_batch_export_coef(b, i, outs[i].qcoef, outs[i].dqcoef, outs[i].mem);
    }
    for (void *p : tmp) (void)hipFree(p);
    zj_batch_destroy(b);
    return rc;
}

// ---------------------------------------------------------------------------------- a stream split over several GPUs
// An image without restart markers is one serial chain (src/decoder.rs:376-389): nothing in the stream says where an
// MCU begins or what the DC predictors are until everything before has been decoded.  The chain is cut the way the
// subsequences of one GPU are: part r starts its walk `lead_bytes` in front of its boundary from a guessed state, and
// by the boundary its walk has met the true one (self-synchronisation).  What it cannot know on its own -- whether
// that is so, the raster index of its first MCU and the predictors there -- comes out of one all-gather of a small
// report per part: the part before says where ITS walk, whose start is already established, sees the first block
// behind the common boundary and which place in the MCU that block has; if the two parts agree there, everything
// behind is the same walk.  MCUs belong to the part in whose range they begin.
#define ZJ_RANGE_TAIL 4096u // raw bytes a part reads beyond its end: more than any MCU can take (10 blocks of at most 1 665 bits, stuffed)

// Kept bytes of raw[a, b) (src/bitstream.rs:37-50: a 00 behind an FF is dropped); a and b do not split such a pair.
static uint64_t kept_bytes(const uint8_t *raw, uint64_t a, uint64_t b)
{
    uint64_t n = b - a;
    for (uint64_t k = a; k + 1 < b; k++) {
        const uint8_t *ff = (const uint8_t *)memchr(raw + k, 0xff, (size_t)(b - 1 - k));
        if (!ff) break;
        k = (uint64_t)(ff - raw);
        if (raw[k + 1] == 0x00) { n--; k++; }
    }
    return n;
}

// A byte offset of the scan moved forward until it does not follow an FF (so that it splits no FF 00 pair).
static uint64_t clear_of_ff(const uint8_t *raw, uint64_t at, uint64_t len)
{
    while (at > 0 && at < len && raw[at - 1] == 0xff) at++;
    return at;
}

extern "C" zj_batch *zj_batch_create_range(zj_ctx *ctx, const zj_image *img, uint32_t rank, uint32_t nranks, uint32_t lead_bytes, uint32_t flags)
{
    if (!ctx) return nullptr;
    ctx->err_code = ZJ_E_INVALID;
    if (!img || nranks == 0 || rank >= nranks) { ctx->err = "bad rank"; return nullptr; }
    if (!zj_image_supported(img) || img->restart_interval != 0) { ctx->err = "stream split: one supported image without restart interval"; return nullptr; }
    const uint8_t *raw = img->scan;
    const uint64_t len = img->scan_len;
    auto boundary = [&](uint32_t r) -> uint64_t { return r >= nranks ? len : clear_of_ff(raw, (len * r / nranks) & ~63ull, len); };
    const uint64_t bx = rank ? boundary(rank) : 0, by = boundary(rank + 1);
    const uint64_t lo = (rank && bx > lead_bytes) ? clear_of_ff(raw, bx - lead_bytes, len) : 0;
    const uint64_t hi = std::min<uint64_t>(len, by + (rank + 1 < nranks ? ZJ_RANGE_TAIL : 0u));
    if (bx > by || lo > bx) { ctx->err = "stream too short to split"; return nullptr; }
    const uint64_t seg[2] = {lo, hi};
    zj_batch *b = batch_create_impl(ctx, img, 1, flags, ZJ_ALLOC_OWN, nullptr, 0, nullptr, 0, nullptr, nullptr, seg);
    if (!b) return nullptr;
    b->range.on = true;
    b->range.rank = rank;
    b->range.nranks = nranks;
    b->range.last = rank + 1 == nranks;
    b->range.x_bits = 8u * kept_bytes(raw, lo, bx);
    b->range.y_bits = 8u * kept_bytes(raw, lo, by);
    b->range.stage = 0;
    return b;
}

// The first block that begins at or behind bit `x` of the part's stream, from the bitmap and the window counts the
// walks left: its position, the number of blocks that begin before it (without the segment's bias), and the place in
// its MCU that the walk gives it (from the state recorded at the end of the window before).
struct ZjBoundary {
    bool found;
    uint64_t pos;
    uint32_t before, k;
};

static bool read_boundary(zj_batch *b, uint64_t x, uint32_t pay, uint32_t bias, ZjBoundary *out)
{
    zj_ctx *c = b->ctx;
    const ZjSegDev &sg = b->segs[0];
    const ZjImgDev &d = b->imgs[0];
    const uint32_t nwin = std::min<uint64_t>((uint64_t)sg.nsub * sg.nc, ((uint64_t)pay + ZJ_WIN_BITS - 1) / ZJ_WIN_BITS);
    out->found = false;
    uint32_t w = (uint32_t)(x / ZJ_WIN_BITS);
    for (uint32_t tries = 0; tries < 64u && w < nwin; tries++, w++) {
        uint32_t words[ZJ_WIN_WORDS];
        if (!hip_ok(c, hipMemcpy(words, b->P.bitmap + sg.out_off + (size_t)w * ZJ_WIN_WORDS * 4u, sizeof words, hipMemcpyDeviceToHost), "D2H bitmap")) return false;
        uint32_t below = 0;
        for (uint32_t i = 0; i < ZJ_WIN_WORDS; i++) {
            for (uint32_t bit = 0; bit < 32u; bit++) {
                if (!((words[i] >> bit) & 1u)) continue;
                const uint64_t p = (uint64_t)w * ZJ_WIN_BITS + i * 32u + bit;
                if (p < x) { below++; continue; }
                uint32_t wb = 0, rec = 0;
                if (!hip_ok(c, hipMemcpy(&wb, b->P.wbase + sg.first_win + w, 4, hipMemcpyDeviceToHost), "D2H window base")) return false;
                if (w > 0) {
                    const uint32_t q = (w - 1u) / sg.nc, cc = (w - 1u) % sg.nc;
                    if (!hip_ok(c, hipMemcpy(&rec, b->P.winfo + sg.first_win + zj_rec_index(q, cc, sg.nc, sg.lanes), 4, hipMemcpyDeviceToHost), "D2H record")) return false;
                }
                const ZjState st = zj_unpack_state(rec, 0);
                out->found = true;
                out->pos = p;
                out->before = wb - bias + below; // wbase carries the bias the last scan ran with
                out->k = (st.b + (st.z != 0 ? 1u : 0u) + below) % d.bpm;
                return true;
            }
        }
    }
    return true;
}

extern "C" int zj_batch_range_front(zj_batch *b, zj_range_report *report)
{
    if (!b || !report || !b->range.on) return ZJ_E_INVALID;
    zj_ctx *c = b->ctx;
    if (!hip_ok(c, hipSetDevice(c->device), "hipSetDevice")) return ZJ_E_DEVICE;
    hipStream_t st = c->stream;
    const ZjImgDev &d = b->imgs[0];
    const uint32_t bpm = d.bpm;
    memset(report, 0, sizeof *report);
    // the walks, and the block counts per window without a bias
    b->segs[0].blk_bias = 0;
    if (!hip_ok(c, hipMemcpyAsync(const_cast<ZjSegDev *>(b->P.segs), &b->segs[0], sizeof(ZjSegDev), hipMemcpyHostToDevice, st), "H2D segment")) return ZJ_E_DEVICE;
    if (!hip_ok(c, hipMemsetAsync(b->P.blockpos + d.first_block, 0, 16 * sizeof(uint32_t), st), "memset positions")) return ZJ_E_DEVICE;
    zj_launch_range_sync(b->P, b->L, st);
    uint32_t pay = 0, flags4[4] = {0, 0, 0, 0};
    if (!hip_ok(c, hipGetLastError(), "kernel launch") || !hip_ok(c, hipStreamSynchronize(st), "stream sync") ||
        !hip_ok(c, hipMemcpy(&pay, b->P.seg_payload_bits, 4, hipMemcpyDeviceToHost), "D2H payload") ||
        !hip_ok(c, hipMemcpy(flags4, b->P.flags, 8, hipMemcpyDeviceToHost), "D2H flags")) return ZJ_E_DEVICE;
    if (flags4[0]) {
        // a fix lane ran through a whole chunk: iterate the fix pass to its fixed point, as zj_batch_finish does, and count again
        for (uint32_t it = 0; flags4[0] && it <= b->L.nchunks; it++) {
            if (!hip_ok(c, hipMemsetAsync(b->P.flags, 0, 8, st), "memset flags")) return ZJ_E_DEVICE;
            zj_launch_fix_only(b->P, b->L, st);
            if (!hip_ok(c, hipMemcpyAsync(flags4, b->P.flags, 8, hipMemcpyDeviceToHost, st), "D2H flags") || !hip_ok(c, hipStreamSynchronize(st), "stream sync")) return ZJ_E_DEVICE;
            b->L.fix_iters = std::min(16, b->L.fix_iters + 1);
        }
        if (flags4[0]) { c->err = "part of a split stream: the walks did not converge, decode the stream whole"; return ZJ_E_CHAIN; }
        zj_launch_range_dc(b->P, b->L, st); // the window counts again (the rest of it is repeated below, once the bias is known)
        if (!hip_ok(c, hipGetLastError(), "kernel launch") || !hip_ok(c, hipStreamSynchronize(st), "stream sync")) return ZJ_E_DEVICE;
    }
    ZjBoundary bx, by;
    if (!read_boundary(b, b->range.x_bits, pay, 0, &bx)) return ZJ_E_DEVICE;
    if (!bx.found) { c->err = "part of a split stream: no block begins behind its boundary"; return ZJ_E_CHAIN; }
    // Number the blocks so that the first one behind the boundary gets its place in its MCU as its number: every
    // number behind it is then congruent to the block's place, and what the walks took for blocks in the lead-in gets
    // numbers below zero (beyond the frame's block count in unsigned arithmetic), which the position pass drops.
    const uint32_t bias = bx.k - bx.before;
    b->segs[0].blk_bias = bias;
    if (!hip_ok(c, hipMemcpyAsync(const_cast<ZjSegDev *>(b->P.segs), &b->segs[0], sizeof(ZjSegDev), hipMemcpyHostToDevice, st), "H2D segment")) return ZJ_E_DEVICE;
    zj_launch_range_dc(b->P, b->L, st);
    uint32_t found = 0;
    if (!hip_ok(c, hipGetLastError(), "kernel launch") || !hip_ok(c, hipStreamSynchronize(st), "stream sync") ||
        !hip_ok(c, hipMemcpy(&found, b->P.seg_found, 4, hipMemcpyDeviceToHost), "D2H block count")) return ZJ_E_DEVICE;
    const uint32_t gx = bias + bx.before;
    uint32_t gy = found;
    report->start_delta = (int32_t)(bx.pos - b->range.x_bits);
    report->start_k = gx % bpm;
    if (!b->range.last) {
        if (!read_boundary(b, b->range.y_bits, pay, bias, &by)) return ZJ_E_DEVICE;
        if (!by.found) { c->err = "part of a split stream: no block begins behind its end"; return ZJ_E_CHAIN; }
        gy = bias + by.before;
        report->end_delta = (int32_t)(by.pos - b->range.y_bits);
        report->end_k = gy % bpm;
    } else {
        report->end_delta = 0;
        report->end_k = gy % bpm; // 0 when the stream holds whole MCUs
    }
    const uint32_t m_first = (gx + bpm - 1) / bpm, m_end = std::max(m_first, (gy + bpm - 1) / bpm);
    b->range.m_first_local = m_first;
    b->range.g_first = m_first * bpm;
    b->range.g_end = m_end * bpm;
    if ((uint64_t)b->range.g_end > (uint64_t)d.nblocks) { c->err = "part of a split stream: more blocks than the frame holds"; return ZJ_E_CHAIN; }
    report->mcus = m_end - m_first;
    // DC prefix in front of the first MCU owned and behind the last one: per component the value of its last block in
    // the MCU before
    auto prefix_at = [&](uint32_t g_mcu_end, uint32_t *pre) -> bool {
        pre[0] = pre[1] = pre[2] = 0;
        if (g_mcu_end < bpm) return true;
        int16_t v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (!hip_ok(c, hipMemcpy(v, b->P.dcabs + d.first_block + (g_mcu_end - bpm), bpm * sizeof(int16_t), hipMemcpyDeviceToHost), "D2H DC values")) return false;
        for (uint32_t k = 0; k < bpm; k++) pre[(d.compsel >> (2u * k)) & 3u] = (uint16_t)v[k];
        return true;
    };
    uint32_t end[3];
    if (!prefix_at(b->range.g_first, b->range.before) || !prefix_at(b->range.g_end, end)) return ZJ_E_DEVICE;
    for (int k = 0; k < 3; k++) report->dc_sum[k] = (int16_t)(uint16_t)(end[k] - b->range.before[k]);
    if (report->mcus == 0) report->dc_sum[0] = report->dc_sum[1] = report->dc_sum[2] = 0;
    b->range.rep = *report;
    b->range.stage = 1;
    return ZJ_OK;
}

extern "C" int zj_batch_range_resolve(zj_batch *b, const zj_range_report *all, uint32_t nranks, uint32_t *mcu_first, uint32_t *mcu_count, uint32_t *row0, uint32_t *nrows)
{
    if (!b || !all || !b->range.on || b->range.stage < 1 || nranks != b->range.nranks) return ZJ_E_INVALID;
    zj_ctx *c = b->ctx;
    if (!hip_ok(c, hipSetDevice(c->device), "hipSetDevice")) return ZJ_E_DEVICE;
    ZjImgDev &d = b->imgs[0];
    // the chain of boundaries: part r's walk is the true one if part r - 1's is and both see the same block, at the same
    // place in its MCU, first behind their common boundary; part 0 starts at the stream's first bit
    uint64_t mcu0 = 0, before_last = 0;
    uint32_t pred[3] = {0, 0, 0};
    if (all[0].start_delta != 0 || all[0].start_k != 0) { c->err = "split stream: the first part does not begin with a block"; return ZJ_E_CHAIN; }
    for (uint32_t r = 0; r < nranks; r++) {
        if (r > 0 && (all[r].start_delta != all[r - 1].end_delta || all[r].start_k != all[r - 1].end_k)) {
            c->err = "split stream: part " + std::to_string(r) + " had not met the true walk at its boundary (lead-in too short for this stream)";
            return ZJ_E_CHAIN;
        }
        if (r < b->range.rank) {
            mcu0 += all[r].mcus;
            for (int k = 0; k < 3; k++) pred[k] = (pred[k] + (uint16_t)all[r].dc_sum[k]) & 0xFFFFu;
        }
        if (r + 1 < nranks) before_last += all[r].mcus;
    }
    // The last part takes what the frame still lacks: behind the last MCU the walks may have taken the padding bits of
    // the final byte for one more block, so its own count is an upper bound.
    const uint64_t frame = (uint64_t)d.mcux * d.mcuy;
    if (before_last > frame || frame - before_last > all[nranks - 1].mcus) { c->err = "split stream: the parts do not add up to the frame's MCUs"; return ZJ_E_CHAIN; }
    const uint32_t n = b->range.last ? (uint32_t)(frame - before_last) : b->range.rep.mcus;
    b->range.g_end = b->range.g_first + n * d.bpm;
    d.mcu_lo = (uint32_t)mcu0;
    d.mcu_hi = (uint32_t)mcu0 + n;
    d.mcu_shift = (uint32_t)mcu0 - b->range.m_first_local;
    d.row0 = n ? d.mcu_lo / d.mcux : 0;
    d.nrows = n ? (d.mcu_hi + d.mcux - 1) / d.mcux - d.row0 : 0;
    for (int k = 0; k < 3; k++) b->range.off[k] = (pred[k] - b->range.before[k]) & 0xFFFFu;
    b->L.nk1w = (n + b->range.per_wg - 1) / b->range.per_wg;
    b->L.k2_count[b->range.cls] = (b->range.strips_x * d.nrows + b->P.k2_ns - 1) / b->P.k2_ns;
    if (!hip_ok(c, hipMemcpyAsync(const_cast<ZjImgDev *>(b->P.imgs), &d, sizeof(ZjImgDev), hipMemcpyHostToDevice, c->stream), "H2D image")) return ZJ_E_DEVICE;
    if (!hip_ok(c, hipStreamSynchronize(c->stream), "stream sync")) return ZJ_E_DEVICE;
    if (mcu_first) *mcu_first = d.mcu_lo;
    if (mcu_count) *mcu_count = n;
    if (row0) *row0 = d.row0;
    if (nrows) *nrows = d.nrows;
    b->range.stage = 2;
    return ZJ_OK;
}

extern "C" int zj_batch_range_decode(zj_batch *b, uint8_t *rgb, size_t stride)
{
    if (!b || !b->range.on || b->range.stage < 2) return ZJ_E_INVALID;
    zj_ctx *c = b->ctx;
    if (!hip_ok(c, hipSetDevice(c->device), "hipSetDevice")) return ZJ_E_DEVICE;
    if (b->range.stage == 2) { // the DC values are corrected once
        zj_launch_range_write(b->P, b->L, c->stream, b->range.g_first, b->range.g_end, b->range.off);
        b->range.stage = 3;
    } else {
        const uint32_t zero[3] = {0, 0, 0};
        zj_launch_range_write(b->P, b->L, c->stream, 0, 0, zero);
    }
    if (!hip_ok(c, hipGetLastError(), "kernel launch")) return ZJ_E_DEVICE;
    b->entropy_launched = true;
    b->dc_patched = false;
    uint8_t *r1[1] = {b->imgs[0].nrows ? rgb : nullptr};
    size_t s1[1] = {stride};
    return decode_pixels(b, r1, s1, c->stream);
}

// ---------------------------------------------------------------------------------- no-allocation entry point
// zj_decode_batch for callers that cannot allocate: device outputs only, plan and scratch memory inside
// zj_noalloc_scratch.  Calls no hipMalloc / hipFree / hipHostMalloc and creates no stream or event; the host
// side uses the C heap for the planner's lists.
extern "C" int zj_decode_batch_noalloc(zj_ctx *ctx, const zj_image *imgs, size_t n, uint8_t *const *rgb, const size_t *stride, uint32_t *status)
{
    if (!ctx || (!imgs && n)) return ZJ_E_INVALID;
    if (!hip_ok(ctx, hipSetDevice(ctx->device), "hipSetDevice")) return ZJ_E_DEVICE;
    uint8_t *base = noalloc_ptr(ctx, ctx->device);
    if (!base) return ZJ_E_DEVICE;
    zj_batch *b = batch_create_impl(ctx, imgs, n, 0, ZJ_ALLOC_NONE, nullptr, 0, base, ZJ_NOALLOC_MAX_BYTES);
    if (!b) {
        if (ctx->err_code == ZJ_E_NOMEM) ctx->err = "input exceeds the no-allocation entry point's fixed scratch of " + std::to_string(ZJ_NOALLOC_MAX_BYTES) + " bytes: " + ctx->err;
        return ctx->err_code ? ctx->err_code : ZJ_E_DEVICE;
    }
    int rc = zj_batch_decode(b, rgb, stride);
    if (rc == ZJ_OK) rc = zj_batch_finish(b, status);
    zj_batch_destroy(b); // frees host lists only: the batch owns no device memory
    return rc;
}
