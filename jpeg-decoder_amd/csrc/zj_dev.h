// zj_dev.h -- kernel parameter block and launch geometry shared by zj_kernels.hip and zj_api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "zj_core.h"

// Kernel order used by the per-kernel timing interface.
enum {
    ZJ_K_UNSTUFF = 0,
    ZJ_K_SYNC,
    ZJ_K_FIX,
    ZJ_K_SCAN,
    ZJ_K_POSITIONS,
    ZJ_K_DCSCAN,
    ZJ_K_WRITE,
    ZJ_K_IDCT_RGB,
    ZJ_K_TOTAL, // whole decode call
};

struct ZjParams {
    const ZjImgDev *imgs;
    const ZjSegDev *segs;
    const uint8_t *raw;         // stuffed scans, every image 16-byte aligned
    uint8_t *ust;               // unstuffed segments
    uint32_t *seg_payload_bits; // [seg] bits of unstuffed payload
    uint32_t *seg_found;        // [seg] blocks found in the payload
    const uint32_t *chunk_seg;  // [chunk of 64 subsequences] segment or ZJ_NONE
    uint32_t *winfo;            // per segment: [chunk][window of the subsequence][lane] window records (zj_core.h)
    uint32_t *wbase;            // per segment: [window in stream order] blocks that begin earlier in the segment
    uint8_t *bitmap;            // block-start bitmap, laid out like `ust` (one bit per stream bit)
    const uint32_t *ptile_seg;  // [position-pass workgroup] segment
    const uint32_t *dtile_seg;  // [DC-scan tile of 256 MCUs] segment; behind the tiles, the segments of at most 64 MCUs
    uint32_t *dtile_sum;        // [DC-scan tile][4] per component: sum of the tile's differences, then of the tiles before it
    const uint32_t *tile_seg;   // [unstuff tile] segment; behind the tiles, the segments short enough for one small tile each
    uint64_t *tile_status;      // [unstuff tile] look-back word: flag << 32 | bytes
    uint32_t *seg_ticket;       // [segment] next tile of the segment to hand out (stored behind tile_status)
    uint32_t *blockpos;         // [block] bit position of the block in its segment's stream
    int16_t *dcabs;             // [block] DC value after prediction (before the DC scan: the difference)
    uint8_t *dcbits;            // [block] bits of the DC symbol and its value (ZJ_DC_BAD: no code)
    int16_t *qcoef;             // [block][64] zig-zag, quantised: the stage seam
    const uint16_t *luts;       // [table set][ZJ_SET_ENTRIES]
    const uint16_t *wluts;      // the same sets as walk entries
    const uint32_t *ptabs;      // the same sets as pair entries, [table set][ZJ_PSET_ENTRIES]
    const uint32_t *wtabs;      // the AC tables of the same sets as write entries, [table set][2][ZJ_WT_ENTRIES]
    const uint32_t *wt_info;    // [table set][2] form of those tables: the direct form's t16, or ZJ_WT_TWO_LEVEL
    const ZjCanon *canons;      // [table set][4]
    const uint16_t *qts;        // quantisation tables widened to 16 bits, 64 entries each, zig-zag order
    const uint32_t *blk256_img; // [block / 256] image
    const uint32_t *k1w_img;    // [write-pass workgroup] image
    const uint32_t *strip_img;     // [stage-2 workgroup] index of its strip group inside its image
    const uint32_t *strip_img_img; // [stage-2 workgroup] image
    uint32_t *status;           // [image]
    uint32_t *flags;            // [0] fix pass ran through a whole chunk, [1] record chain inconsistent; cleared by every decode call,
                                // together with what lies behind them in one block: status words, look-back words, tickets
    uint32_t *counters;         // the same two events counted over the batch's life
    uint8_t *const *rgb;        // [image] device pointers
    const size_t *rgb_stride;   // [image]
    uint32_t nchunks;
    uint32_t nseg;
    uint32_t k2_ns;             // strips per workgroup of the IDCT/colour kernel
    uint32_t dc_in_write;       // the write pass stores the DC differences it decodes and the DC scan follows it (0: a pass of its own and the scan come first)
};

#define ZJ_K2_NS 4 // strips per workgroup of the IDCT/colour kernel at least; the planner doubles it up to 16 in large batches (ZjParams::k2_ns)
// MCUs of a strip that one wavefront of the IDCT/colour kernel takes (a strip is four times that): chosen so
// that (MCUs) x (pixel rows of an MCU) = 64, one (pixel row, MCU) task per lane.  One-component frames: 24.
__host__ __device__ constexpr uint32_t zj_k2_wave_mcus(uint32_t hs, uint32_t vs) { return vs == 2u ? 4u : 8u; }
#define ZJ_K2_CLASSES 5
#define ZJ_UNSTUFF_TILE 15872u // raw bytes per workgroup of the unstuff pass: 128 threads x 31 words
#ifndef ZJ_UNSTUFF_REP
#define ZJ_UNSTUFF_REP 8u // tiles per workgroup of the unstuff pass (a workgroup takes tiles of its segment until none is left)
#endif
#define ZJ_UNSTUFF_SMALL 3840u // and of its one-wavefront form for segments that fit into one such tile: 64 threads x 15 words

struct ZjLaunch {
    uint32_t nseg, nchunks, nblk256, nstrips, nk1w, nptiles, ntiles, ndtiles;
    uint32_t nsmall;  // segments unstuffed by the one-wavefront form; their entries of tile_seg follow the `ntiles` others
    uint32_t ndsmall; // segments of at most 64 MCUs, scanned by one wavefront each; their entries of dtile_seg follow the `ndtiles` others
    uint32_t ndmulti; // segments of more than one DC-scan tile
    uint32_t dc_by_segment; // many long segments (a batch of images without restart markers): one workgroup scans a whole segment
    uint32_t npacked; // chunks shared by several short segments (the packed form of the sync kernel)
    uint32_t k2_first[ZJ_K2_CLASSES], k2_count[ZJ_K2_CLASSES]; // stage-2 workgroups per layout (2x2, 1x1, 2x1, 1x2, one component)
    int fix_iters;
    size_t clear_bytes; // from ZjParams::flags on: what a decode call clears before its first kernel
};

// Restart-marker search over the uploaded scans (batch planning): one entry per image that has a restart interval.
#define ZJ_MARK_TILE 16384u // scan bytes per workgroup: 256 threads x 64 bytes
struct ZjMarkImg {
    uint64_t raw_off;    // the image's scan inside the raw arena (16-byte aligned)
    uint32_t scan_len;
    uint32_t first_tile; // of this image in the tile list
    uint32_t ntiles;
    uint32_t out_first;  // first entry of this image in `offs`
    uint32_t cap;        // markers wanted (restart intervals of the frame); later ones are counted only
    uint32_t pad;
};
struct ZjMarkParams {
    const uint8_t *raw;
    const ZjMarkImg *imgs;
    const uint32_t *tile_img; // [tile] index into imgs
    uint32_t *tile_count;     // [tile] markers in the tile, then (after the scan) markers of the image before the tile
    uint32_t *img_count;      // [image] markers found
    uint32_t *offs;           // per image: scan offset of the byte after its k-th marker
    uint32_t nimgs, ntiles;
};

extern "C" {
void zj_launch_markers(const ZjMarkParams &M, hipStream_t st);
void zj_launch_stage1(const ZjParams &P, const ZjLaunch &L, hipStream_t st, hipEvent_t *ev);
void zj_launch_fix_only(const ZjParams &P, const ZjLaunch &L, hipStream_t st);
void zj_launch_stage1_tail(const ZjParams &P, const ZjLaunch &L, hipStream_t st);
void zj_launch_stage2(const ZjParams &P, const ZjLaunch &L, hipStream_t st, hipEvent_t *ev);
void zj_launch_range_sync(const ZjParams &P, const ZjLaunch &L, hipStream_t st);
void zj_launch_range_dc(const ZjParams &P, const ZjLaunch &L, hipStream_t st);
void zj_launch_range_write(const ZjParams &P, const ZjLaunch &L, hipStream_t st, uint32_t g0, uint32_t g1, const uint32_t *off);
void zj_launch_dequant(const ZjParams &P, uint32_t ii, uint32_t nblocks, int16_t *dst, hipStream_t st);
void zj_launch_patch_dc(const ZjParams &P, uint32_t nblocks, hipStream_t st);
}
