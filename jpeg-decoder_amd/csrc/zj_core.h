// zj_core.h -- data layout and per-lane entropy-decode primitives shared by the HIP kernels and
// by the host-side table builder.  Everything here is plain integer code that compiles for the
// device (hipcc) and for the host (g++), so the same functions can be exercised lane by lane on a
// CPU in the tests.
//
// Arithmetic contract (what "identical to the reference" means for stage 1):
//   * canonical Huffman decode of one symbol            -- src/ec.rs:19-49
//   * EXTEND / sign_code                                -- src/ec.rs:52-59
//   * block walk: DC symbol, AC run/size loop, EOB, stop at index 64 -- src/decoder.rs:186-235
// For every stream the reference decodes without panicking, any correct canonical decoder yields
// the same symbols, so the lookup tables below are free to be organised for the GPU.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ZJ_HD_NOINLINE __attribute__((host)) __attribute__((device)) inline __attribute__((noinline))
#define ZJ_HD __attribute__((host)) __attribute__((device)) inline __attribute__((always_inline))
#else
#define ZJ_HD inline
#define ZJ_HD_NOINLINE inline
#endif

// ---- lookup-table format -------------------------------------------------------------------
// A table set holds four tables in the order DC id 0, DC id 1, AC id 0, AC id 1.  Each table is
// ZJ_TBL_ENTRIES uint16 entries: a first level indexed by the next ZJ_L1_BITS bits of the stream,
// followed by second-level sub-tables for longer codes.
//   entry, bit 15 clear:  [12:8] code length (0 = no such code), [7:0] symbol
//   entry, bit 15 set  :  first level only: [14:12] nb = extra index bits (1..7), [8:0] offset of the
//                         sub-table inside the second level; nb == 0: the table does not fit the
//                         second level, use the canonical search (ZjCanon) instead.
#define ZJ_L1_BITS 9
#define ZJ_L1_SIZE 512
#define ZJ_L2_SIZE 256
#define ZJ_TBL_ENTRIES (ZJ_L1_SIZE + ZJ_L2_SIZE)
#define ZJ_SET_ENTRIES (4 * ZJ_TBL_ENTRIES)
// The walks that only need to follow the decoder state (sync, fix, index) use a second form of the same
// tables whose entries hold the state transition ready-made ("walk entries"):
//   entry, bit 15 clear:  [5:0] bits to consume (code + value bits, 1..32), [12:6] zig-zag advance
//                         (DC symbol: 1; AC symbol: run + 1; EOB: 64, which ends the block; a bit
//                         pattern that is no code: consume 1 bit, advance as an EOB / a DC symbol)
//   entry, bit 15 set  :  first level only: [14:12] nb, [8:0] sub-table offset, as above

// The sync pass, which walks every symbol of every stream once without producing coefficients, uses a third
// form with up to two symbols per lookup ("pair entries"): ZJ_PT_ENTRIES uint32 entries per table, a first
// level indexed by the next ZJ_PL1_BITS bits followed by second-level sub-tables for longer codes.
//   entry, bit 31 clear:  byte 0  t1   bits of the first symbol, code and value (1..32)
//                         byte 1  t12  bits of both symbols (= t1 when the entry holds one symbol)
//                         byte 2  a1   zig-zag advance of the first symbol (as in a walk entry)
//                         byte 3  a12  advance of both (= a1 when the entry holds one symbol; a second symbol
//                                      that is an EOB adds 64; at most 127, so bit 31 stays clear)
//                         Whole bytes, so that the kernels' arithmetic picks the fields up as operands (SDWA byte
//                         selects) instead of extracting them first.
//   entry, bit 31 set  :  first level only: [26:24] nb = extra index bits (1..6), [8:0] offset of the sub-table
//                         inside the second level; nb == 0: use the canonical search.
// A second symbol is entered only behind an AC symbol that is no EOB, and only when its whole code lies inside
// the index bits (its value bits need not).  The walk takes it when the first symbol neither ends the block
// nor reaches the end of the stretch being walked, so the states a walk passes through are those of the
// symbol-by-symbol walk with some of them skipped -- never a state that walk would not reach.
#define ZJ_PL1_BITS 10
#define ZJ_PL1_SIZE 1024
#define ZJ_PL2_SIZE 256
#define ZJ_PT_ENTRIES (ZJ_PL1_SIZE + ZJ_PL2_SIZE)
#define ZJ_PSET_ENTRIES (4 * ZJ_PT_ENTRIES)

// The write pass, which turns every AC symbol into a coefficient, uses a fourth form of the AC tables ("write
// entries"): ZJ_TBL_ENTRIES uint32 entries per table, laid out like the 16-bit form, whose fields are whole bytes so
// that the kernel's arithmetic can pick them up as operands without extracting them first.
//   entry, bit 31 clear:  byte 0  bits of the symbol, code and value (0 for a bit pattern that is no code)
//                         byte 1  code length (0: no such code)
//                         byte 2  zig-zag advance: run + 1, or 64 for an EOB and for a pattern that is no code
//                                 (either ends the block)
//                         byte 3  31 - size of the value (16..31)
//   entry, bit 31 set  :  first level only: [26:24] nb, [8:0] sub-table offset, as in the 16-bit form
#define ZJ_WE_START ((1u << 8) | (64u << 16) | (31u << 24)) // what a lane holds before its first symbol: a valid code that ended a block
// A table whose long codes are few enough is stored in a second, "direct" form that needs one lookup for every code
// (canonical codes grow with their length, so all codes longer than the first level lie above the short ones): with
// t16 = the first code longer than ZJ_L1_BITS bits, left-aligned in 16 bits (a multiple of 128; 0x10000 when there is
// none), the entry for the next 16 bits x of the stream is at index (x >> 7) + max(0, x - t16) -- the first level for
// x < t16, and behind it one entry per 16-bit value, which begins in the first-level slots of the long prefixes.  The
// form is used when 512 + 0x10000 - t16 <= ZJ_WT_DIRECT (the Annex K tables need 1152 entries); otherwise the write
// entries stay in the two-level layout (ZJ_WT_TWO_LEVEL in place of t16).
#define ZJ_WT_ENTRIES 1280
// Of those, the direct form may use the first ZJ_WT_DIRECT; the 128 words behind them hold the first level of a DC table,
// one byte per entry: code length | size << 4, or 0 for "consult the 16-bit table" (a longer code, no code, a size over 15).
#define ZJ_WT_DIRECT 1152
#define ZJ_WT_TWO_LEVEL 0xFFFFFFFFu

// Canonical description of one table, used by the rare fallback path only.
struct ZjCanon {
    uint16_t ncode[17];   // number of codes of length l (index 1..16)
    uint16_t mincode[17]; // smallest code of length l (right-aligned)
    uint16_t valptr[17];  // index of its symbol
    uint8_t symbols[256];
    uint16_t nsym;
};

// ---- device-resident descriptors -----------------------------------------------------------
struct ZjImgDev {
    uint32_t w, h;
    uint32_t mcux, mcuy;
    uint32_t hs, vs, bpm;   // luma sampling factors, blocks per MCU (hs*vs + 2)
    uint32_t ri_blocks;     // blocks per restart segment (0 = one segment)
    uint32_t first_seg, nseg;
    uint32_t first_sub;     // first global subsequence index (multiple of 64; of 256 where the table set changes)
    uint32_t nsub;          // subsequence slots of this image (multiple of 64)
    uint32_t first_block;   // first global block index (multiple of 256)
    uint32_t nblocks;
    uint32_t lut_set;       // index into the table-set arena
    uint32_t dcsel;         // bit k: DC table id of block k inside the MCU
    uint32_t acsel;         // bit k: AC table id of block k inside the MCU
    uint32_t compsel;       // 2 bits per block k: component of block k
    uint32_t qt_off[3];     // element offsets of the three components' quantisation tables (64 entries each)
    uint32_t first_strip;   // first global strip index of stage 2
    uint32_t strips_x;      // strips per MCU row
    uint32_t flags;         // bit 0: supported, bit 1: the table set needs the canonical search for some codes,
                            // bit 2: so do its pair tables
    uint32_t first_k1w;     // first workgroup of the write pass
    uint32_t k1w_mcus;      // MCUs per wavefront of the write pass (a multiple of 64)
    uint32_t row0, nrows;   // MCU rows whose pixels are produced; the pixel buffer begins at MCU row row0
    uint32_t mcu_lo, mcu_hi; // MCUs (raster index in the image) whose coefficients the write pass produces: those of the
                            // rows above, or the MCUs that begin in this rank's part of a stream split over several GPUs
    uint32_t mcu_shift;     // such a part numbers the blocks it finds from its own first one: MCU m of the image is MCU
                            // m - mcu_shift of the position list, the DC values and the DC symbol lengths (0 otherwise)
};

struct ZjSegDev {
    uint64_t raw_begin, raw_end; // byte range in the raw (stuffed) scan arena
    uint64_t out_off;            // byte offset of the unstuffed stream in its arena (multiple of 16)
    uint32_t out_cap;            // bytes reserved there (multiple of 16, >= payload + 16, zero filled)
    uint32_t img;
    uint32_t first_sub;          // global subsequence index of the segment's first lane (multiple of `lanes`)
    uint32_t nsub;               // subsequences that can hold payload: ceil(raw bits / subsequence bits)
    uint32_t first_block;        // global block index of the segment's first block
    uint32_t nblocks;            // blocks coded in this segment
    uint32_t subbits;            // bits per subsequence of this segment (multiple of ZJ_WIN_BITS)
    uint32_t nc;                 // windows per subsequence = subbits / ZJ_WIN_BITS
    uint32_t lanes;              // lanes of a wavefront the segment's subsequences share: 64, or 1..32 in a packed chunk
    uint32_t first_win;          // offset of the segment's window records / window bases (lanes * nc per chunk or group)
    uint32_t first_ptile;        // first workgroup of the position pass (256 windows each)
    uint32_t first_tile, ntiles; // tiles of the unstuff pass (15 KiB of raw bytes each, from raw_begin & ~15)
    uint32_t first_dtile, ndtiles; // tiles of the DC scan (256 MCUs each, at least one)
    uint32_t blk_bias;           // number given to the first block found in the stream: 0, or -- in a part of a split stream,
                                 // which begins somewhere inside an MCU -- what gives the first block behind the part's
                                 // boundary its place in its MCU as its number (negative, modulo 2^32, when the walks
                                 // found blocks in the lead-in: those numbers fall outside the frame and are dropped)
    uint32_t pad;
};

// Window records.  Every segment's unstuffed stream is cut into windows of ZJ_WIN_BITS bits; a
// subsequence (the stretch one lane decodes on its own) is a run of subbits / ZJ_WIN_BITS windows.
// The record of a window says where a decoder stands when it crosses the window's end, and how many
// blocks began inside the window:
//   [5:0]   overflow: bits the last symbol reaches past the window's end (0..32)
//   [9:6]   index of the current block inside its MCU
//   [15:10] zig-zag index inside the block (0 = a DC symbol comes next)
//   [31:16] blocks whose DC symbol starts inside this window
// Records of one chunk (64 subsequences) are stored [window of the subsequence][lane], so that the
// lanes of a wavefront touch consecutive words.
//
// Next to the records the walks leave a bitmap with one bit per stream bit (word w of the bitmap
// belongs to stream word w; bit p & 31 of word p >> 5, counted from the least significant bit, is set
// when a block's DC symbol starts at bit p).  A window's 16 bitmap words are always written together
// by the lane that walked the window last, so a later, better-informed walk replaces them whole.
#define ZJ_WIN_WORDS 16
#define ZJ_WIN_BITS (ZJ_WIN_WORDS * 32)
#define ZJ_INFO_STATE_MASK 0xFFFFu

// Index of the record of window c of subsequence q (both relative to the segment) inside the
// segment's record area; nc = windows per subsequence, lanes = lanes of a wavefront the segment's
// subsequences share (64, or the size of the segment's group in a packed chunk).
ZJ_HD uint32_t zj_rec_index(uint32_t q, uint32_t c, uint32_t nc, uint32_t lanes = 64u) { return ((q / lanes) * nc + c) * lanes + (q % lanes); }

// How a segment of `raw_bits` stuffed bits is cut into subsequences for a wanted size of `target` bits.
// A segment that fills at least half a chunk (64 subsequences) at that size is cut into a whole number
// of chunks, with the subsequence size that fills them evenly, rounded up to whole windows: every lane
// of every chunk then has work (*lanes = 64).  A shorter segment shares a chunk with its neighbours: it
// gets a group of *lanes = 1, 2, 4 .. 32 lanes, as many as it has subsequences of about `small_target`
// bits, so that a wavefront is filled by several segments instead of by slivers of one (a subsequence
// much shorter than the distance a wrong guess survives is mostly walked again by its neighbours).
ZJ_HD void zj_seg_layout(uint64_t raw_bits, uint32_t target, uint32_t small_target, uint32_t *subbits, uint32_t *nsub, uint32_t *lanes)
{
    if (target < ZJ_WIN_BITS) target = ZJ_WIN_BITS;
    if (small_target < target) small_target = target;
    uint64_t per_lane;
    if (raw_bits >= 32ull * target) {
        const uint64_t nchunk = (raw_bits + 32ull * target) / (64ull * target); // nearest whole number of chunks, at least one
        per_lane = (raw_bits + 64ull * nchunk - 1) / (64ull * nchunk);
        *lanes = 64u;
    } else {
        uint64_t want = (raw_bits + small_target / 2u) / small_target;
        if (want > 32u) want = 32u;
        uint32_t g = 1;
        while (g < want) g *= 2u;
        per_lane = (raw_bits + g - 1u) / g;
        *lanes = g;
    }
    uint64_t sb = (per_lane + ZJ_WIN_BITS - 1) / ZJ_WIN_BITS * ZJ_WIN_BITS;
    if (sb == 0) sb = ZJ_WIN_BITS;
    *subbits = (uint32_t)sb;
    *nsub = (uint32_t)((raw_bits + sb - 1) / sb);
}

struct ZjState {
    uint32_t p; // bit position inside the segment's unstuffed stream
    uint32_t b; // block inside the MCU
    uint32_t z; // zig-zag index, 0 = at a block start
};

ZJ_HD uint32_t zj_pack_info(const ZjState &s, uint32_t nominal_end, uint32_t nblk)
{
    uint32_t o = s.p > nominal_end ? s.p - nominal_end : 0u;
    return (o & 63u) | (s.b << 6) | (s.z << 10) | (nblk << 16);
}
ZJ_HD ZjState zj_unpack_state(uint32_t info, uint32_t nominal_end)
{
    ZjState s;
    s.p = nominal_end + (info & 63u);
    s.b = (info >> 6) & 15u;
    s.z = (info >> 10) & 63u;
    return s;
}

// ---- bit reader over the unstuffed stream ----------------------------------------------------
ZJ_HD uint32_t zj_bswap32(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bswap32(x);
#else
    return (x >> 24) | ((x >> 8) & 0xff00u) | ((x << 8) & 0xff0000u) | (x << 24);
#endif
}

// MSB-first reader.  `base` is the 16-byte aligned start of a segment's unstuffed stream; the arena
// keeps readable, zero-filled bytes behind every segment, which reproduces the reference's zero
// padding at the end of data (src/bitstream.rs:80-84).  The stream is fetched in 16-byte pieces, one
// piece ahead of the one being consumed, so that the load of piece n+1 is in flight while the ~24
// symbols of piece n are decoded (a lane's loads are far apart in memory from its neighbours'; what
// hides their latency is this distance between issue and use, not coalescing).
struct alignas(16) ZjU4 {
    uint32_t x, y, z, w;
};

struct ZjBits {
    const ZjU4 *base;
    uint32_t pidx; // next piece to load
    ZjU4 cur, nxt; // piece being consumed (shifted down as words leave), piece after it
    uint32_t left; // words left in cur
    uint64_t buf;  // left-aligned window
    int32_t cnt;   // valid bits in buf, always > 32 between calls

    ZJ_HD uint32_t take()
    {
        if (left == 0) {
            cur = nxt;
            left = 4;
            nxt = base[pidx++];
        }
        uint32_t w = cur.x;
        cur.x = cur.y; cur.y = cur.z; cur.z = cur.w;
        left--;
        return zj_bswap32(w);
    }
    ZJ_HD void init(const uint32_t *b, uint32_t bitpos)
    {
        base = reinterpret_cast<const ZjU4 *>(b);
        const uint32_t word = bitpos >> 5, piece = word >> 2;
        cur = base[piece];
        nxt = base[piece + 1];
        pidx = piece + 2;
        left = 4;
        if (word & 1u) { cur.x = cur.y; cur.y = cur.z; cur.z = cur.w; left -= 1; }
        if (word & 2u) { cur.x = cur.z; cur.y = cur.w; left -= 2; }
        uint64_t w0 = take(), w1 = take();
        const uint32_t sh = bitpos & 31u;
        buf = ((w0 << 32) | w1) << sh;
        cnt = 64 - (int32_t)sh;
    }
    ZJ_HD uint32_t top32() const { return (uint32_t)(buf >> 32); }
    ZJ_HD void skip(uint32_t n) // n <= 32
    {
        buf <<= n;
        cnt -= (int32_t)n;
        if (cnt <= 32) {
            uint64_t w = take();
            buf |= w << (32 - cnt);
            cnt += 32;
        }
    }
};

// Reader for one block (the per-block write pass): 64 bytes are fetched up front, which covers a
// block of at least 385 bits without another load inside the symbol loop; longer blocks reload.
struct ZjBits4 {
    const ZjU4 *base;
    uint32_t widx; // stream word index of the next word to take
    ZjU4 cur, n1, n2, n3;
    uint32_t left; // words left in cur..n3
    uint64_t buf;
    int32_t cnt;

    ZJ_HD void fill(uint32_t word)
    {
        const uint32_t piece = word >> 2;
        cur = base[piece]; n1 = base[piece + 1]; n2 = base[piece + 2]; n3 = base[piece + 3];
        left = 16;
        if (word & 1u) { cur.x = cur.y; cur.y = cur.z; cur.z = cur.w; left -= 1; }
        if (word & 2u) { cur.x = cur.z; cur.y = cur.w; left -= 2; }
        widx = word;
    }
    ZJ_HD uint32_t take()
    {
        if (left == 0) fill(widx);
        else if ((left & 3u) == 0) { cur = n1; n1 = n2; n2 = n3; }
        uint32_t w = cur.x;
        cur.x = cur.y; cur.y = cur.z; cur.z = cur.w;
        left--;
        widx++;
        return zj_bswap32(w);
    }
    // the same with the four pieces that hold bit `bitpos` and the 48 bytes behind it already loaded
    ZJ_HD void init_loaded(const uint32_t *b, uint32_t bitpos, const ZjU4 &c0, const ZjU4 &c1, const ZjU4 &c2, const ZjU4 &c3)
    {
        base = reinterpret_cast<const ZjU4 *>(b);
        const uint32_t word = bitpos >> 5;
        cur = c0; n1 = c1; n2 = c2; n3 = c3;
        left = 16;
        if (word & 1u) { cur.x = cur.y; cur.y = cur.z; cur.z = cur.w; left -= 1; }
        if (word & 2u) { cur.x = cur.z; cur.y = cur.w; left -= 2; }
        widx = word;
        uint64_t w0 = take_first(), w1 = take();
        const uint32_t sh = bitpos & 31u;
        buf = ((w0 << 32) | w1) << sh;
        cnt = 64 - (int32_t)sh;
    }
    ZJ_HD void init(const uint32_t *b, uint32_t bitpos)
    {
        base = reinterpret_cast<const ZjU4 *>(b);
        fill(bitpos >> 5);
        // fill() leaves `left` one short of a multiple of four exactly when words were skipped, so the
        // piece rotation in take() must not fire before the first piece is used up
        uint64_t w0 = take_first(), w1 = take();
        const uint32_t sh = bitpos & 31u;
        buf = ((w0 << 32) | w1) << sh;
        cnt = 64 - (int32_t)sh;
    }
    ZJ_HD uint32_t take_first()
    {
        uint32_t w = cur.x;
        cur.x = cur.y; cur.y = cur.z; cur.z = cur.w;
        left--;
        widx++;
        return zj_bswap32(w);
    }
    ZJ_HD uint32_t top32() const { return (uint32_t)(buf >> 32); }
    ZJ_HD void skip(uint32_t n)
    {
        buf <<= n;
        cnt -= (int32_t)n;
        if (cnt <= 32) {
            uint64_t w = take();
            buf |= w << (32 - cnt);
            cnt += 32;
        }
    }
};

// ---- one Huffman symbol ----------------------------------------------------------------------
// Canonical search, the textbook form of src/ec.rs:30-47 (first length whose code range holds the
// window).  Returns an entry in the short format, 0 if the window starts no code.
ZJ_HD_NOINLINE uint32_t zj_canon_lookup(const ZjCanon *c, uint32_t top32)
{
    uint32_t w16 = top32 >> 16;
    for (uint32_t l = 1; l <= 16; l++) {
        uint32_t nc = c->ncode[l], code = w16 >> (16 - l);
        if (nc != 0 && code >= c->mincode[l] && code < (uint32_t)c->mincode[l] + nc) {
            uint32_t idx = c->valptr[l] + (code - c->mincode[l]);
            return (l << 8) | c->symbols[idx & 255u];
        }
    }
    return 0;
}

// Walk entry of a (length, symbol) entry; `isdc` says which kind of table it came from.
ZJ_HD uint32_t zj_walk_entry(uint32_t e, bool isdc)
{
    const uint32_t len = (e >> 8) & 31u, sym = e & 255u;
    if (len == 0) return 1u | ((isdc ? 1u : 64u) << 6);
    const uint32_t sbits = isdc ? (sym > 16u ? 16u : sym) : (sym & 15u);
    const uint32_t zadv = isdc ? 1u : (sym == 0 ? 64u : (sym >> 4) + 1u);
    return (len + sbits) | (zadv << 6);
}

// Write entry of a 16-bit AC table entry (first-level pointers keep their fields).
ZJ_HD uint32_t zj_write_entry(uint32_t e)
{
    if (e & 0x8000u) return 0x80000000u | (((e >> 12) & 7u) << 24) | (e & 0x1FFu);
    const uint32_t len = (e >> 8) & 31u, sym = e & 255u;
    if (len == 0) return (64u << 16) | (31u << 24);
    const uint32_t size = sym & 15u, zadv = sym == 0 ? 64u : (sym >> 4) + 1u;
    return (len + size) | (len << 8) | (zadv << 16) | ((31u - size) << 24);
}

// First-level entry for the next bits of the stream.  LDS = true says the table is in LDS (kernels only):
// the entry's address is then formed by one shift-and-add; left to the compiler, (x >> 23) << 1 becomes a
// shift, a mask and an add, in a loop that runs once per Huffman symbol.
template <bool LDS>
ZJ_HD uint32_t zj_tbl_first(const uint16_t *tbl, uint32_t top32)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (LDS) {
        const uint32_t base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint16_t *)tbl;
        uint32_t addr;
        asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(addr) : "v"(top32 >> (32 - ZJ_L1_BITS)), "v"(base));
        return *(const __attribute__((address_space(3))) uint16_t *)(uintptr_t)addr;
    }
#endif
    return tbl[top32 >> (32 - ZJ_L1_BITS)];
}

// wtbl: one table of walk entries (ZJ_TBL_ENTRIES entries).
// CANON = false is for table sets whose long codes all fit the second level (the batch planner knows):
// the loop then contains no call.
template <bool CANON = true, bool LDS = false>
ZJ_HD uint32_t zj_lookup_walk(const uint16_t *wtbl, const ZjCanon *canon, uint32_t top32, bool isdc)
{
    uint32_t e = zj_tbl_first<LDS>(wtbl, top32);
    if (e & 0x8000u) {
        const uint32_t nb = (e >> 12) & 7u;
        if (CANON && nb == 0) return zj_walk_entry(zj_canon_lookup(canon, top32), isdc);
        e = wtbl[ZJ_L1_SIZE + (e & 0x1FFu) + ((top32 >> (32 - ZJ_L1_BITS - nb)) & ((1u << nb) - 1u))];
    }
    return e;
}

// tbl: one table (ZJ_TBL_ENTRIES entries).  Returns [12:8] length, [7:0] symbol; length 0 = invalid.
template <bool CANON = true, bool LDS = false>
ZJ_HD uint32_t zj_lookup(const uint16_t *tbl, const ZjCanon *canon, uint32_t top32)
{
    uint32_t e = zj_tbl_first<LDS>(tbl, top32);
    if (e & 0x8000u) {
        uint32_t nb = (e >> 12) & 7u;
        if (CANON && nb == 0) return zj_canon_lookup(canon, top32);
        e = tbl[ZJ_L1_SIZE + (e & 0x1FFu) + ((top32 >> (32 - ZJ_L1_BITS - nb)) & ((1u << nb) - 1u))];
    }
    return e;
}

// src/ec.rs:52-59 with the i16 wrap and masked shift of a --release build.
ZJ_HD int32_t zj_extend(uint32_t nbits, uint32_t v)
{
    if (((v << 1) >> nbits) != 0) return (int32_t)(int16_t)v;
    int32_t maxv = (int32_t)(int16_t)((1u << (nbits & 15u)) - 1u);
    return (int32_t)(int16_t)((int32_t)(int16_t)v - maxv);
}

// DC difference of the block whose DC symbol starts at bit `p` (src/decoder.rs:194-199 without the
// predictor): one table lookup and the EXTEND of the bits behind the code.
// *nbits receives the bits the DC symbol and its value take, or ZJ_DC_BAD if the bits are no code.
#define ZJ_DC_BAD 0xFFu
ZJ_HD int32_t zj_dc_diff(const uint32_t *stream, const uint16_t *dctbl, const ZjCanon *canon, uint32_t p, uint32_t *nbits)
{
    const uint32_t w = p >> 5, sh = p & 31u;
    const uint64_t two = ((uint64_t)zj_bswap32(stream[w]) << 32) | zj_bswap32(stream[w + 1]);
    const uint32_t top = (uint32_t)((two << sh) >> 32);
    const uint32_t e = zj_lookup(dctbl, canon, top);
    const uint32_t len = (e >> 8) & 31u;
    uint32_t s = e & 255u;
    if (len == 0 || s > 16u) { *nbits = ZJ_DC_BAD; return 0; } // not a code: the write pass reports it
    *nbits = len + s;
    const uint32_t v = s ? ((top << len) >> (32 - s)) : 0u;
    return zj_extend(s, v);
}

// The value bits behind an AC code and their EXTEND in one go, for nbits 0..15 (AC sizes): `x` is the
// 32-bit window shifted left past the code.  Equal to zj_extend(nbits, x >> (32 - nbits)); the form
// avoids the compare and the special case for nbits == 0.
ZJ_HD int32_t zj_value_ac(uint32_t x, uint32_t nbits)
{
    const uint32_t v = (x >> 1) >> (31u - nbits);
    const uint32_t neg = (uint32_t)((int32_t)(~x) >> 31); // all ones when the leading value bit is 0
    return (int32_t)(int16_t)(v - (neg & ((1u << nbits) - 1u)));
}

struct ZjSel { // table selectors of the image a lane works on
    uint32_t bpm, dcsel, acsel;
};

// Table index (0 = DC id 0, 1 = DC id 1, 2 = AC id 0, 3 = AC id 1) of block k of the MCU as one word:
// bits [2k+1:2k] for the block's DC symbol, bits [2k+17:2k+16] for its AC symbols.
ZJ_HD uint32_t zj_selword(const ZjSel &s)
{
    uint32_t w = 0;
    for (uint32_t k = 0; k < 8u; k++) {
        w |= ((s.dcsel >> k) & 1u) << (2u * k);
        w |= (2u + ((s.acsel >> k) & 1u)) << (2u * k + 16u);
    }
    return w;
}

// What a walk leaves behind besides the state: block_start(p, f) is called for every symbol, with
// f = 1 when the symbol at bit p is the DC symbol of a block and f = 0 otherwise (a call that always
// happens keeps the loop free of branches on the decoder state).
struct ZjNoEmit {
    ZJ_HD void block_start(uint32_t, uint32_t) {}
};
struct ZjEmitWords { // bitmap words addressed directly (host side, and the one-lane fix pass)
    uint32_t *words;
    ZJ_HD void block_start(uint32_t p, uint32_t f)
    {
        if (f) words[p >> 5] |= 1u << (p & 31u);
    }
};

// Walk the symbols that START inside [st.p, stop) -- the serial loop of src/decoder.rs:194-235 seen
// as a state machine over (bit position, block in MCU, zig-zag index) -- on a reader that is already
// positioned at st.p.  `wlut` is a table set of walk entries (LDS = true: in LDS, see zj_tbl_first).
template <class BITS, class EMIT, bool CANON = true, bool LDS = false>
ZJ_HD void zj_decode_run(BITS &bits, const uint16_t *wlut, const ZjCanon *canon, uint32_t selw, uint32_t bpm,
                         ZjState &st, uint32_t stop, EMIT &em)
{
    uint32_t p = st.p, b = st.b, z = st.z;
    while (p < stop) {
        // everything below is select-based on purpose: lanes of a wavefront sit at different points of
        // different blocks, so branches on the decoder state would serialise
        const uint32_t top = bits.top32();
        const bool isdc = (z == 0);
        const uint32_t t = (selw >> (2u * b + (isdc ? 0u : 16u))) & 3u;
        const uint32_t e = zj_lookup_walk<CANON, LDS>(wlut + t * ZJ_TBL_ENTRIES, canon ? canon + t : nullptr, top, t < 2u);
        const uint32_t tot = e & 63u;
        em.block_start(p, isdc ? 1u : 0u);
        const uint32_t z2 = z + ((e >> 6) & 127u);
        const bool endblk = z2 >= 64u;
        z = endblk ? 0u : z2;
        b += endblk ? 1u : 0u;
        b = (b >= bpm) ? 0u : b;
        bits.skip(tot);
        p += tot;
    }
    st.p = p; st.b = b; st.z = z;
}

// Pair entry holding the single symbol of a walk entry.
ZJ_HD uint32_t zj_pair_single(uint32_t walk_entry)
{
    const uint32_t tot = walk_entry & 63u, adv = (walk_entry >> 6) & 127u;
    return tot | (tot << 8) | (adv << 16) | (adv << 24);
}

// First-level pair entry for the next bits of the stream (LDS = true: the table is in LDS, kernels only).
template <bool LDS>
ZJ_HD uint32_t zj_ptbl_first(const uint32_t *ptbl, uint32_t top32)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (LDS) {
        const uint32_t base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint32_t *)ptbl;
        uint32_t addr;
        asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(addr) : "v"(top32 >> (32 - ZJ_PL1_BITS)), "v"(base));
        return *(const __attribute__((address_space(3))) uint32_t *)(uintptr_t)addr;
    }
#endif
    return ptbl[top32 >> (32 - ZJ_PL1_BITS)];
}

template <bool CANON = true, bool LDS = false>
ZJ_HD uint32_t zj_lookup_pair(const uint32_t *ptbl, const ZjCanon *canon, uint32_t top32, bool isdc)
{
    uint32_t e = zj_ptbl_first<LDS>(ptbl, top32);
    if (e & 0x80000000u) {
        const uint32_t nb = (e >> 24) & 7u;
        if (CANON && nb == 0) return zj_pair_single(zj_walk_entry(zj_canon_lookup(canon, top32), isdc));
        e = ptbl[ZJ_PL1_SIZE + (e & 0x1FFu) + ((top32 >> (32 - ZJ_PL1_BITS - nb)) & ((1u << nb) - 1u))];
    }
    return e;
}

// zj_decode_run with pair entries: the same walk, up to two symbols a step.  `ptab` is a table set of pair
// entries.  The second symbol of an entry is taken only if the first neither ends the block (by an EOB or
// by reaching coefficient 63) nor is the last one that starts before `stop`; the states at st.p on return
// are therefore exactly those of the symbol-by-symbol walk.
template <class BITS, class EMIT, bool CANON = true, bool LDS = false>
ZJ_HD void zj_decode_run_pairs(BITS &bits, const uint32_t *ptab, const ZjCanon *canon, uint32_t selw, uint32_t bpm,
                               ZjState &st, uint32_t stop, EMIT &em)
{
    uint32_t p = st.p, b = st.b, z = st.z;
    while (p < stop) {
        const uint32_t top = bits.top32();
        const bool isdc = (z == 0);
        const uint32_t t = (selw >> (2u * b + (isdc ? 0u : 16u))) & 3u;
        const uint32_t e = zj_lookup_pair<CANON, LDS>(ptab + t * ZJ_PT_ENTRIES, canon ? canon + t : nullptr, top, t < 2u);
        const uint32_t t1 = e & 255u, a1 = (e >> 16) & 255u;
        const uint32_t z1 = z + a1;
        const bool one = z1 >= 64u || p + t1 >= stop;
        const uint32_t tot = one ? t1 : ((e >> 8) & 255u);
        const uint32_t z2 = one ? z1 : z + (e >> 24);
        em.block_start(p, isdc ? 1u : 0u);
        const bool endblk = z2 >= 64u;
        z = endblk ? 0u : z2;
        b += endblk ? 1u : 0u;
        b = (b >= bpm) ? 0u : b;
        bits.skip(tot);
        p += tot;
    }
    st.p = p; st.b = b; st.z = z;
}

// Number of block starts marked in a window's 16 bitmap words.
ZJ_HD uint32_t zj_popc32(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_popcount(x);
#else
    x = x - ((x >> 1) & 0x55555555u);
    x = (x & 0x33333333u) + ((x >> 2) & 0x33333333u);
    return (((x + (x >> 4)) & 0x0F0F0F0Fu) * 0x01010101u) >> 24;
#endif
}

// One window of a walk on directly addressed memory: clears the window's bitmap words, walks the
// symbols that start before `stop`, returns the window's record.  `win` is the window's index in
// the segment; bitmap = the segment's bitmap (word 0 = stream word 0).
ZJ_HD uint32_t zj_walk_window_direct(const uint32_t *stream, uint32_t *bitmap, const uint16_t *wlut,
                                     const ZjCanon *canon, uint32_t selw, uint32_t bpm, ZjState &st, uint32_t win,
                                     uint32_t stop)
{
    uint32_t *bw = bitmap + (uint64_t)win * ZJ_WIN_WORDS;
    for (uint32_t i = 0; i < ZJ_WIN_WORDS; i++) bw[i] = 0;
    if (st.p < stop) {
        ZjBits bits;
        bits.init(stream, st.p);
        ZjEmitWords em = {bitmap};
        zj_decode_run<ZjBits, ZjEmitWords>(bits, wlut, canon, selw, bpm, st, stop, em);
    }
    uint32_t n = 0;
    for (uint32_t i = 0; i < ZJ_WIN_WORDS; i++) n += zj_popc32(bw[i]);
    return zj_pack_info(st, (win + 1u) * ZJ_WIN_BITS, n);
}

// Decode one block that begins at bit `p0` into row[0..63] (zig-zag order, already zeroed; row[64] and
// row[65] are scratch that absorbs the store of the symbol that ends the block).  The DC value comes
// from the prefix scan; only its bits are skipped here.  The AC loop is written with selects instead
// of branches: the lanes of a wavefront are at different points of different blocks.  Returns status
// bits and the bit position behind the block.
ZJ_HD uint32_t zj_decode_block_on(ZjBits4 &bits, const uint16_t *lut, const ZjCanon *canon, uint32_t tdc, uint32_t tac,
                                  uint32_t p0, int16_t dc, int16_t *row, uint32_t *pend);

ZJ_HD uint32_t zj_decode_block(const uint32_t *stream, const uint16_t *lut, const ZjCanon *canon, uint32_t tdc,
                               uint32_t tac, uint32_t p0, int16_t dc, int16_t *row, uint32_t *pend)
{
    ZjBits4 bits;
    bits.init(stream, p0);
    return zj_decode_block_on(bits, lut, canon, tdc, tac, p0, dc, row, pend);
}

// The same on a reader that is already positioned at p0.
ZJ_HD uint32_t zj_decode_block_on(ZjBits4 &bits, const uint16_t *lut, const ZjCanon *canon, uint32_t tdc, uint32_t tac,
                                  uint32_t p0, int16_t dc, int16_t *row, uint32_t *pend)
{
    uint32_t p = p0, status = 0;
    {
        uint32_t top = bits.top32();
        uint32_t e = zj_lookup(lut + tdc * ZJ_TBL_ENTRIES, canon ? canon + tdc : nullptr, top);
        uint32_t len = (e >> 8) & 31u, s = e & 255u;
        if (len == 0 || s > 16u) { *pend = p; return 1u; /* ZJ_ST_BAD_CODE */ }
        bits.skip(len + s);
        p += len + s;
    }
    row[0] = dc;
    const uint16_t *ac = lut + (2u + tac) * ZJ_TBL_ENTRIES;
    const ZjCanon *cac = canon ? canon + 2u + tac : nullptr;
    uint32_t z = 1, e, z2;
    for (;;) {
        const uint32_t top = bits.top32();
        e = zj_lookup(ac, cac, top);
        const uint32_t len = (e >> 8) & 31u, s = e & 15u;
        z2 = z + ((e >> 4) & 15u);                 // src/decoder.rs:225
        const bool eob = (e & 255u) == 0;          // :212 (a pattern that is no code reads as 0 too)
        const uint32_t at = z2 < 64u ? z2 : 64u;   // :228 would index out of bounds at 64 and above
        row[eob ? 64u : at] = (int16_t)zj_value_ac(top << len, s);
        const uint32_t adv = len + s;
        p += adv;
        if (eob || z2 >= 63u) break;               // :212, :232 (and the overflow case)
        z = z2 + 1u;                               // :230
        bits.skip(adv);
    }
    if (((e >> 8) & 31u) == 0) status |= 1u;               // ZJ_ST_BAD_CODE
    else if ((e & 255u) != 0 && z2 >= 64u) status |= 2u;   // ZJ_ST_COEF_OVERFLOW
    *pend = p;
    return status;
}
