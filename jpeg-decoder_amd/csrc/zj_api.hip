// zj_api.hip -- the C ABI of include/zj.h on top of the kernels: context, batch planning (all
// host-side layout decisions, table construction and uploads), the decode call that launches the
// kernels, result/status retrieval.  The decode call itself allocates nothing.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <map>
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
};

struct zj_batch {
    zj_ctx *ctx;
    size_t n;
    std::vector<ZjImgDev> imgs;
    std::vector<ZjSegDev> segs;
    std::vector<uint32_t> unsupported; // per image: 1 = rejected at plan time
    ZjParams P;
    ZjLaunch L;
    std::vector<void *> allocs;
    // pinned staging for the per-call pointer tables
    uint8_t **h_rgb;
    size_t *h_stride;
    uint8_t **d_rgb;
    size_t *d_stride;
    uint32_t *h_flags; // pinned, 4 words
    uint64_t scan_bytes, coef_bytes, rgb_bytes, pixels;
    bool profiling;
    hipEvent_t ev[ZJ_NUM_KERNELS + 1];
    bool ev_valid;
    bool decoded;
};

static bool hip_ok(zj_ctx *c, hipError_t e, const char *what)
{
    if (e == hipSuccess) return true;
    if (c) c->err = std::string(what) + ": " + hipGetErrorString(e);
    return false;
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
    c->own_stream = (stream == nullptr);
    if (stream) c->stream = (hipStream_t)stream;
    else if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return nullptr; }
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

template <typename T>
static T *dev_alloc(zj_batch *b, size_t count, bool zero = false)
{
    void *p = nullptr;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = 16;
    if (!hip_ok(b->ctx, hipMalloc(&p, bytes), "hipMalloc")) return nullptr;
    b->allocs.push_back(p);
    if (zero && !hip_ok(b->ctx, hipMemsetAsync(p, 0, bytes, b->ctx->stream), "hipMemset")) return nullptr;
    return (T *)p;
}

template <typename T>
static T *dev_upload(zj_batch *b, const std::vector<T> &v)
{
    T *p = dev_alloc<T>(b, v.size());
    if (!p) return nullptr;
    if (!v.empty() && !hip_ok(b->ctx, hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy H2D"))
        return nullptr;
    return p;
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

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
        if (nout > 0xF0000000ull || tile_img.size() + d.ntiles > 0xF0000000ull) { ctx->err = "batch too large for 32-bit indices"; return false; }
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
    hipStream_t st = ctx->stream;
    bool ok = hip_ok(ctx, hipMalloc(&d_imgs, mi.size() * sizeof(ZjMarkImg)), "hipMalloc") &&
              hip_ok(ctx, hipMalloc(&d_tile_img, tile_img.size() * 4), "hipMalloc") &&
              hip_ok(ctx, hipMalloc(&d_tile_count, tile_img.size() * 4), "hipMalloc") &&
              hip_ok(ctx, hipMalloc(&d_img_count, mi.size() * 4), "hipMalloc") &&
              hip_ok(ctx, hipMalloc(&d_offs, (size_t)(nout + 1) * 4), "hipMalloc");
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
    (void)hipFree(d_imgs); (void)hipFree(d_tile_img); (void)hipFree(d_tile_count); (void)hipFree(d_img_count); (void)hipFree(d_offs);
    return ok;
}

extern "C" void zj_batch_destroy(zj_batch *b)
{
    if (!b) return;
    (void)hipSetDevice(b->ctx->device);
    (void)hipStreamSynchronize(b->ctx->stream);
    for (void *p : b->allocs) (void)hipFree(p);
    if (b->h_rgb) (void)hipHostFree(b->h_rgb);
    if (b->h_stride) (void)hipHostFree(b->h_stride);
    if (b->h_flags) (void)hipHostFree(b->h_flags);
    if (b->ev_valid)
        for (auto &e : b->ev) (void)hipEventDestroy(e);
    delete b;
}

extern "C" zj_batch *zj_batch_create(zj_ctx *ctx, const zj_image *images, size_t n, uint32_t flags)
{
    if (!ctx || (!images && n)) return nullptr;
    if (!hip_ok(ctx, hipSetDevice(ctx->device), "hipSetDevice")) return nullptr;
    zj_batch *b = new zj_batch();
    b->ctx = ctx;
    b->n = n;
    b->h_rgb = nullptr; b->h_stride = nullptr; b->h_flags = nullptr;
    b->ev_valid = false; b->profiling = false; b->decoded = false;
    b->scan_bytes = b->coef_bytes = b->rgb_bytes = b->pixels = 0;
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
        subbits = 1024;
        while (subbits < 65536u && (uint64_t)subbits * lanes < bits) subbits *= 2u;
    }
    if (subbits < ZJ_WIN_BITS || subbits > (1u << 20)) { ctx->err = "subsequence bits must be in 512..1048576"; delete b; return nullptr; }

    b->imgs.resize(n);
    b->unsupported.assign(n, 0);
    std::vector<uint32_t> chunk_seg, blk256_img, k1w_img, ptile_seg, tile_seg, dtile_seg;
    uint64_t nwin_total = 0;
    std::vector<uint32_t> k2_grp[ZJ_K2_CLASSES], k2_img[ZJ_K2_CLASSES]; // stage-2 workgroups per sampling layout
    std::vector<uint16_t> luts;
    std::vector<ZjCanon> canons;
    std::vector<uint16_t> qts;
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
        mcus_total += zj_image_mcus(&images[i]);
        raw_off[i] = raw_total;
        raw_total += align_up(images[i].scan_len, 16) + 16;
    }
    uint8_t *raw = dev_alloc<uint8_t>(b, raw_total + 1040, false);
    if (!raw) { zj_batch_destroy(b); return nullptr; }
    {
        // the entropy-coded bytes as stored, each image at a 16-byte aligned offset
        bool up = hip_ok(ctx, hipMemsetAsync(raw, 0, raw_total + 1040, ctx->stream), "hipMemset");
        for (size_t i = 0; up && i < n; i++) {
            if (!zj_image_supported(&images[i]) || images[i].scan_len == 0) continue;
            up = hip_ok(ctx, hipMemcpyAsync(raw + raw_off[i], images[i].scan, images[i].scan_len, hipMemcpyHostToDevice, ctx->stream),
                        "hipMemcpy scan");
        }
        if (up) up = hip_ok(ctx, hipStreamSynchronize(ctx->stream), "sync after upload");
        if (!up) { zj_batch_destroy(b); return nullptr; }
    }
    std::vector<uint32_t> mark_first(n, ZJ_NONE), mark_count, mark_outfirst, mark_offs; // device-found restart offsets per image
    if (!(flags & ZJ_FLAG_HOST_MARKER_SCAN) && !find_markers(b, images, n, raw, raw_off, mark_first, mark_count, mark_outfirst, mark_offs)) {
        zj_batch_destroy(b);
        return nullptr;
    }

    // MCUs per wavefront of the write pass: 32 (64 for 4:4:4) times a power of two up to 8, as long as the batch
    // still gives every CU several sets of wavefronts (measured: x8 is 7 % faster than x1 on a large batch).
    uint32_t k1w_mult = 1;
    {
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
        while (k1w_mult < 8u && mcus_total / (64u * k1w_mult) >= (uint64_t)cus * 12u * 4u) k1w_mult *= 2u;
    }
    // Strips per workgroup of the IDCT/colour kernel: 4, doubled up to 16 while the batch keeps every CU
    // supplied with workgroups (a strip is about 16 MCUs; measured: 16 is 4 % faster than 4 on a large batch).
    uint32_t k2_ns = ZJ_K2_NS;
    {
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
        while (k2_ns < 16u && mcus_total / 16u / (k2_ns * 2u) >= (uint64_t)cus * 16u) k2_ns *= 2u;
    }
    b->P.k2_ns = k2_ns;
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
                if (k + 1 == nseg) end = (k + 1 < nint) ? offs[(size_t)k + 1] - 2 : im.scan_len;
            }
            s.raw_begin = raw_off[i] + beg;
            s.raw_end = raw_off[i] + end;
            const size_t rl = end - beg;
            s.out_off = ust_total;
            s.out_cap = (uint32_t)(align_up(rl, 16) + 96);
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
            s.first_tile = (uint32_t)tile_seg.size();
            s.ntiles = (uint32_t)((s.raw_end - (s.raw_begin & ~15ull) + ZJ_UNSTUFF_TILE - 1u) / ZJ_UNSTUFF_TILE);
            if (s.ntiles == 0) s.ntiles = 1; // an empty segment still publishes its payload length
            for (uint32_t t = 0; t < s.ntiles; t++) tile_seg.push_back((uint32_t)b->segs.size());
            s.first_ptile = (uint32_t)ptile_seg.size();
            for (uint64_t w = 0; w < (uint64_t)s.nsub * s.nc; w += 256u) ptile_seg.push_back((uint32_t)b->segs.size());
            const uint64_t m0 = ri ? k * ri : 0;
            const uint64_t mc = ri ? std::min<uint64_t>(ri, nmcu - m0) : nmcu;
            s.first_block = d.first_block + (uint32_t)(m0 * d.bpm);
            s.nblocks = (uint32_t)(mc * d.bpm);
            s.first_dtile = (uint32_t)dtile_seg.size();
            s.ndtiles = (uint32_t)std::max<uint64_t>(1, (mc + 255u) / 256u);
            for (uint32_t t = 0; t < s.ndtiles; t++) dtile_seg.push_back((uint32_t)b->segs.size());
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
            d.first_k1w = (uint32_t)k1w_img.size();
            for (uint64_t m = 0; m < nmcu; m += per_wg) k1w_img.push_back((uint32_t)i);
        }
        const uint32_t T = ncomp == 1 ? 96u : 4u * zj_k2_wave_mcus(d.hs, d.vs); // MCUs per strip of stage 2
        d.strips_x = (d.mcux + T - 1) / T;
        const uint32_t ns = (d.strips_x * d.mcuy + k2_ns - 1) / k2_ns; // workgroups of stage 2
        const int cls = ncomp == 1 ? 4 : (d.hs == 2 && d.vs == 2) ? 0 : (d.hs == 1 && d.vs == 1) ? 1 : (d.hs == 2) ? 2 : 3;
        for (uint32_t s = 0; s < ns; s++) { k2_grp[cls].push_back(s); k2_img[cls].push_back((uint32_t)i); }
        nstrip_total += ns;
        b->scan_bytes += im.scan_len;
        b->coef_bytes += (uint64_t)d.nblocks * 128;
        b->rgb_bytes += (uint64_t)d.w * d.h * 3;
        b->pixels += (uint64_t)d.w * d.h;
        if (nsub_total > 0xF0000000ull || nblk_total > 0xF0000000ull || nwin_total > 0xF0000000ull) { ctx->err = "batch too large for 32-bit indices"; delete b; return nullptr; }
    }

    while (chunk_seg.size() % 4u) chunk_seg.push_back(ZJ_NONE); // whole workgroups
    ZjParams &P = b->P;
    bool ok = true;
    P.imgs = dev_upload(b, b->imgs);
    P.segs = dev_upload(b, b->segs);
    P.chunk_seg = dev_upload(b, chunk_seg);
    P.blk256_img = dev_upload(b, blk256_img);
    P.k1w_img = dev_upload(b, k1w_img);
    P.ptile_seg = dev_upload(b, ptile_seg);
    P.tile_seg = dev_upload(b, tile_seg);
    P.dtile_seg = dev_upload(b, dtile_seg);
    P.dtile_sum = dev_alloc<uint32_t>(b, dtile_seg.size() * 4, true);
    P.tile_status = dev_alloc<uint64_t>(b, tile_seg.size() + (b->segs.size() + 1) / 2 + 1, true);
    P.seg_ticket = P.tile_status ? reinterpret_cast<uint32_t *>(P.tile_status + tile_seg.size()) : nullptr;
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
    P.luts = dev_upload(b, luts);
    P.wluts = dev_upload(b, wluts);
    P.canons = dev_upload(b, canons);
    P.qts = dev_upload(b, qts);
    P.raw = raw;
    P.ust = dev_alloc<uint8_t>(b, ust_total + 1040, true);
    P.seg_payload_bits = dev_alloc<uint32_t>(b, b->segs.size(), true);
    P.seg_found = dev_alloc<uint32_t>(b, b->segs.size(), true);
    P.winfo = dev_alloc<uint32_t>(b, nwin_total, true);
    P.wbase = dev_alloc<uint32_t>(b, nwin_total, true);
    P.bitmap = dev_alloc<uint8_t>(b, ust_total + 1040, true);
    P.blockpos = dev_alloc<uint32_t>(b, nblk_total, true);
    P.dcabs = dev_alloc<int16_t>(b, nblk_total, true);
    P.dcbits = dev_alloc<uint8_t>(b, nblk_total, true);
    P.qcoef = dev_alloc<int16_t>(b, nblk_total * 64, false);
    P.status = dev_alloc<uint32_t>(b, n, true);
    P.flags = dev_alloc<uint32_t>(b, 4, true);
    b->d_rgb = dev_alloc<uint8_t *>(b, n, true);
    b->d_stride = dev_alloc<size_t>(b, n, true);
    P.rgb = b->d_rgb;
    P.rgb_stride = b->d_stride;
    P.nchunks = (uint32_t)chunk_seg.size();
    P.nseg = (uint32_t)b->segs.size();
    ok = P.imgs && P.segs && P.chunk_seg && P.blk256_img && P.k1w_img && P.strip_img && P.strip_img_img && P.luts && P.wluts && P.canons && P.qts && raw && P.ust &&
         P.seg_payload_bits && P.seg_found && P.winfo && P.wbase && P.bitmap && P.ptile_seg && P.tile_seg && P.tile_status && P.dtile_seg && P.dtile_sum && P.blockpos && P.dcabs && P.dcbits && P.qcoef && P.status &&
         P.flags && b->d_rgb && b->d_stride;
    if (ok) ok = hip_ok(ctx, hipHostMalloc((void **)&b->h_rgb, (n ? n : 1) * sizeof(uint8_t *), hipHostMallocDefault), "hipHostMalloc") &&
                 hip_ok(ctx, hipHostMalloc((void **)&b->h_stride, (n ? n : 1) * sizeof(size_t), hipHostMallocDefault), "hipHostMalloc") &&
                 hip_ok(ctx, hipHostMalloc((void **)&b->h_flags, 16, hipHostMallocDefault), "hipHostMalloc");
    if (!ok) { zj_batch_destroy(b); return nullptr; }
    b->L.nseg = (uint32_t)b->segs.size();
    b->L.nchunks = P.nchunks;
    b->L.nblk256 = (uint32_t)blk256_img.size();
    b->L.nk1w = (uint32_t)k1w_img.size();
    b->L.nptiles = (uint32_t)ptile_seg.size();
    b->L.ntiles = (uint32_t)tile_seg.size();
    b->L.ndtiles = (uint32_t)dtile_seg.size();
    b->L.nstrips = (uint32_t)strip_grp_all.size();
    b->L.fix_iters = 1;
    return b;
}

extern "C" int zj_batch_decode(zj_batch *b, uint8_t *const *rgb, const size_t *stride)
{
    if (!b) return ZJ_E_INVALID;
    zj_ctx *c = b->ctx;
    if (!hip_ok(c, hipSetDevice(c->device), "hipSetDevice")) return ZJ_E_DEVICE;
    hipStream_t st = c->stream;
    bool any_rgb = false;
    for (size_t i = 0; i < b->n; i++) {
        b->h_rgb[i] = (rgb && !b->unsupported[i]) ? rgb[i] : nullptr;
        b->h_stride[i] = stride ? stride[i] : 0;
        any_rgb = any_rgb || b->h_rgb[i];
    }
    if (b->n) {
        if (!hip_ok(c, hipMemcpyAsync(b->d_rgb, b->h_rgb, b->n * sizeof(uint8_t *), hipMemcpyHostToDevice, st), "H2D rgb table")) return ZJ_E_DEVICE;
        if (!hip_ok(c, hipMemcpyAsync(b->d_stride, b->h_stride, b->n * sizeof(size_t), hipMemcpyHostToDevice, st), "H2D stride table")) return ZJ_E_DEVICE;
        if (!hip_ok(c, hipMemsetAsync(b->P.status, 0, b->n * sizeof(uint32_t), st), "memset status")) return ZJ_E_DEVICE;
    }
    if (b->L.nchunks == 0) { b->decoded = true; return ZJ_OK; }
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
    ZjLaunch L2 = b->L;
    if (!any_rgb) { L2.nstrips = 0; for (int c = 0; c < ZJ_K2_CLASSES; c++) L2.k2_count[c] = 0; }
    zj_launch_stage2(b->P, L2, st, ev ? ev + 8 : nullptr);
    if (!hip_ok(c, hipGetLastError(), "kernel launch")) return ZJ_E_DEVICE;
    b->decoded = true;
    return ZJ_OK;
}

extern "C" int zj_batch_finish(zj_batch *b, uint32_t *status)
{
    if (!b) return ZJ_E_INVALID;
    zj_ctx *c = b->ctx;
    if (!hip_ok(c, hipSetDevice(c->device), "hipSetDevice")) return ZJ_E_DEVICE;
    hipStream_t st = c->stream;
    if (!hip_ok(c, hipStreamSynchronize(st), "stream sync")) return ZJ_E_DEVICE;
    if (b->decoded && b->L.nchunks) {
        if (!hip_ok(c, hipMemcpy(b->h_flags, b->P.flags, 16, hipMemcpyDeviceToHost), "D2H flags")) return ZJ_E_DEVICE;
        if (b->h_flags[0] || b->h_flags[1]) {
            // Subsequence records did not converge in the fixed number of fix-up passes: iterate to a
            // fixed point (each pass completes at least one more chunk), then redo everything behind.
            for (uint32_t it = 0; it <= b->L.nchunks; it++) {
                if (!hip_ok(c, hipMemsetAsync(b->P.flags, 0, 16, st), "memset flags")) return ZJ_E_DEVICE;
                zj_launch_fix_only(b->P, b->L, st);
                if (!hip_ok(c, hipMemcpyAsync(b->h_flags, b->P.flags, 16, hipMemcpyDeviceToHost, st), "D2H flags")) return ZJ_E_DEVICE;
                if (!hip_ok(c, hipStreamSynchronize(st), "stream sync")) return ZJ_E_DEVICE;
                if (!b->h_flags[0]) break;
            }
            if (!hip_ok(c, hipMemsetAsync(b->P.status, 0, b->n * sizeof(uint32_t), st), "memset status")) return ZJ_E_DEVICE;
            zj_launch_stage1_tail(b->P, b->L, st);
            bool any_rgb = false;
            for (size_t i = 0; i < b->n; i++) any_rgb = any_rgb || b->h_rgb[i];
            ZjLaunch L2 = b->L;
            if (!any_rgb) { L2.nstrips = 0; for (int c = 0; c < ZJ_K2_CLASSES; c++) L2.k2_count[c] = 0; }
            zj_launch_stage2(b->P, L2, st, nullptr);
            if (!hip_ok(c, hipStreamSynchronize(st), "stream sync")) return ZJ_E_DEVICE;
            if (!hip_ok(c, hipMemcpy(b->h_flags, b->P.flags, 16, hipMemcpyDeviceToHost), "D2H flags")) return ZJ_E_DEVICE;
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

extern "C" const int16_t *zj_batch_qcoef(zj_batch *b, size_t i)
{
    if (!b || i >= b->n || b->unsupported[i]) return nullptr;
    return b->P.qcoef + (size_t)b->imgs[i].first_block * 64;
}

extern "C" int zj_batch_export_coef(zj_batch *b, size_t i, int16_t *qcoef, int16_t *dqcoef, int mem)
{
    if (!b || i >= b->n || b->unsupported[i]) return ZJ_E_INVALID;
    zj_ctx *c = b->ctx;
    if (!hip_ok(c, hipSetDevice(c->device), "hipSetDevice")) return ZJ_E_DEVICE;
    hipStream_t st = c->stream;
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
    return ZJ_OK;
}

extern "C" size_t zj_batch_segments(zj_batch *b, size_t i, uint64_t *ranges, size_t cap)
{
    if (!b || i >= b->n || b->unsupported[i]) return 0;
    const ZjImgDev &d = b->imgs[i];
    const uint64_t base = d.nseg ? b->segs[d.first_seg].raw_begin : 0; // the first segment starts at the scan's first byte
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
    if (!b) return ZJ_E_DEVICE;
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
