// zj_kernels.hip -- gfx950 kernels of the baseline-JPEG decode path.
//
// Stage 1 (entropy decode; replaces the serial MCU loop of src/decoder.rs:366-390 with
// src/ec.rs and src/bitstream.rs underneath):
//   k0_unstuff   FF00 -> FF compaction of every entropy-coded segment      (src/bitstream.rs:37-50)
//   k1a_sync     one lane per fixed-size bitstream subsequence, one wavefront per 64 of them;
//                lanes decode from a guessed state and keep decoding into their neighbours'
//                subsequences until their state meets the recorded one (self-synchronisation)
//                (k1a_sync_packed: the same for short segments that share a wavefront, a lane group each)
//   k1b_fix      stitches consecutive wavefront chunks the same way, one lane per chunk boundary
//   k1s_scan     exclusive prefix sum of blocks-per-subsequence inside each segment
//   k1p_positions  turns the block-start bitmap left by the walks into the list of block positions
//   k1d_dcscan   DC differences from the block positions, DC predictor = wrapping-i16 prefix sum per
//                component                                                  (src/decoder.rs:194-200)
//   k1w_write    one lane per block: decode the block into an LDS tile, store 64 blocks coalesced
// Stage 2+3 (src/decoder.rs:179-184 dequantize, :137-146 de-zigzag, src/dct.rs IDCT,
// src/decoder.rs:265-320 colour + store):
//   k2_idct_rgb  LDS-staged column pass / row pass, chroma replication, YCbCr->RGB, coalesced store
// Batch planning: km_markers / km_scan find the restart markers in the uploaded scans.
// Everything floating point keeps the reference's operation order; the file is compiled with
// -ffp-contract=off so that only the explicit fmaf chains of the IDCT are fused (src/dct.rs:95).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "zj_core.h"
#include "zj_dev.h"

#define ZJ_NONE 0xFFFFFFFFu
typedef uint32_t zj_u32x4 __attribute__((ext_vector_type(4)));
// The coefficient arena and the pixels are written once in whole lines and read once, far later or never: their stores and
// loads carry the non-temporal hint (write pass -2.8 %, IDCT/colour kernel -0.5 %).  Not so the stores of half lines that a
// neighbour completes a moment later -- the sync pass's bitmap words, the unstuff pass's pieces: with the hint they are
// not merged on their way to memory (sync pass 2.9 -> 4.9 ms on 2 048 images).

// ---------------------------------------------------------------------------------- helpers
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= (uint32_t)d) v += t;
    }
    return v;
}

// Inclusive scan over a workgroup of NW waves.  `wsum` is NW+1 words of LDS.  Returns the inclusive
// value; *total receives the workgroup sum.  Contains two barriers.
template <int NW>
__device__ __forceinline__ uint32_t block_incl_scan(uint32_t v, uint32_t *wsum, uint32_t *total)
{
    uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t inc = wave_incl_scan(v, lane);
    __syncthreads();
    if (lane == 63u) wsum[wave] = inc;
    __syncthreads();
    uint32_t off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
        uint32_t s = wsum[w];
        if ((uint32_t)w < wave) off += s;
        tot += s;
    }
    *total = tot;
    return inc + off;
}

// Order LDS accesses of the lanes of one wavefront between two phases (the wavefront runs in lock
// step, so no barrier is needed, only that neither compiler nor hardware reorders around this point).
__device__ __forceinline__ void zj_wave_lds_sync()
{
    // LDS operations of a wavefront are executed in issue order; the wait drains them, the "memory"
    // clobber keeps the compiler from moving LDS accesses across, and global loads stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ bool sub_active(const ZjSegDev &sg, uint32_t q, uint32_t subbits, uint32_t payload_bits)
{
    return q < sg.nsub && (uint64_t)q * subbits < (uint64_t)payload_bits;
}

__device__ __forceinline__ uint32_t sub_end(uint32_t q, uint32_t subbits, uint32_t payload_bits)
{
    uint64_t e = (uint64_t)(q + 1u) * subbits;
    return e < payload_bits ? (uint32_t)e : payload_bits;
}

template <class T>
__device__ __forceinline__ void load_lut_to_lds(T *lds, const T *src)
{
    const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
    uint4 *d4 = reinterpret_cast<uint4 *>(lds);
    for (uint32_t i = threadIdx.x; i < ZJ_SET_ENTRIES * sizeof(T) / 16; i += blockDim.x) d4[i] = s4[i];
}

// ---------------------------------------------------------------------------------- windowed reader
// A lane of the subsequence-parallel kernels reads a private, strictly sequential stretch of the
// stream, far from its neighbours' stretches.  Fetching it word by word from global memory puts a
// full memory latency into almost every iteration of the symbol loop (one lane or another always
// needs a word, and the wait counter is per wavefront).  Instead every lane stages 64-byte windows of
// its own stretch in a private column of LDS -- ZJ_WIN_WORDS words plus two words of look-ahead,
// word w of lane l at [w][l], so the column reads are bank-conflict free -- with all lanes of the
// wavefront staging at the same moment, and the symbol loop refills from LDS.
#define ZJ_WIN_SLOTS (ZJ_WIN_WORDS + 3)

// The reader keeps the two stream words around the current bit (already byte-swapped to MSB-first) in
// registers and fetches the word after them from the column on every symbol, so that moving on to the
// next word is a handful of selects instead of a branch that some lane of the wavefront takes on
// almost every symbol.
// The block-start bitmap of the window shares the column: the bitmap word of the stream word that
// holds the current bit is collected in a register and stored to that word's slot on every symbol
// (the slot's stream word already sits in the reader's registers), so once the reader has moved on, the
// slot holds the finished bitmap word.
struct ZjBitsLds {
    uint32_t *slot; // slot of the stream word that holds the current bit; the column's pitch is 64 words
    uint32_t w0, w1; // that word and the next one
    uint32_t o;      // bits of w0 already consumed, 0..31
    uint32_t bm;     // block starts seen in w0's word
    uint32_t nxt;    // the word after w1, fetched by top32()

    __device__ __forceinline__ void init(uint32_t *col, uint32_t wbase, uint32_t bitpos)
    {
        slot = col + ((bitpos >> 5) - wbase) * 64u;
        w0 = slot[0];
        w1 = slot[64];
        o = bitpos & 31u;
        bm = 0;
    }
    // the column was restaged for the window that starts at stream word `wbase`
    __device__ __forceinline__ void rebase(uint32_t *col, uint32_t wbase, uint32_t bitpos)
    {
        slot = col + ((bitpos >> 5) - wbase) * 64u;
    }
    __device__ __forceinline__ uint32_t top32()
    {
        nxt = slot[128];
        return (uint32_t)(((((uint64_t)w0 << 32) | w1) << o) >> 32);
    }
    // a block's DC symbol starts at the current bit (f = 1) or the symbol there is something else (f = 0)
    __device__ __forceinline__ void block_start(uint32_t, uint32_t f)
    {
        bm |= f << o;
        slot[0] = bm;
    }
    __device__ __forceinline__ void skip(uint32_t n) // n <= 32
    {
        o += n;
        const bool adv = o >= 32u;
        w0 = adv ? w1 : w0;
        w1 = adv ? nxt : w1;
        bm = adv ? 0u : bm;
        slot += adv ? 64 : 0;
        o &= 31u;
    }
    __device__ __forceinline__ uint32_t words_done(const uint32_t *col) const { return (uint32_t)(slot - col) >> 6; }
};

// The 32 stream words of one 128-byte line -- two windows -- as they were loaded.  Memory hands out whole lines: a walk
// that asks for 64 bytes per step gets every line twice, a step apart, when the first copy has left the cache.  So a
// walk keeps the line it is in: the even window of a line is staged from the registers alone (its two words of
// look-ahead are the odd window's first two), the odd window's step loads the next line for its look-ahead and keeps
// it.  Every line of the stream is asked for once.  (Segments start on line boundaries in the unstuffed arena.)
struct ZjLineRegs {
    uint4 v[8];
    uint32_t line; // index of the line held (window index / 2), ZJ_NONE: none
};

__device__ __forceinline__ void zj_load_line(const uint32_t *stream, uint32_t line, ZjLineRegs &r)
{
    const uint4 *src = reinterpret_cast<const uint4 *>(stream + line * (2u * ZJ_WIN_WORDS));
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = src[i];
    r.line = line;
}

// Sixteen words and two of look-ahead into this lane's column, byte-swapped to MSB-first order (one spare slot follows).
__device__ __forceinline__ void zj_store_window(const uint4 &a, const uint4 &b, const uint4 &c, const uint4 &d, uint32_t la0, uint32_t la1,
                                                uint32_t *col)
{
#define ZJ_SW(x) __builtin_bswap32(x)
    col[0 * 64] = ZJ_SW(a.x); col[1 * 64] = ZJ_SW(a.y); col[2 * 64] = ZJ_SW(a.z); col[3 * 64] = ZJ_SW(a.w);
    col[4 * 64] = ZJ_SW(b.x); col[5 * 64] = ZJ_SW(b.y); col[6 * 64] = ZJ_SW(b.z); col[7 * 64] = ZJ_SW(b.w);
    col[8 * 64] = ZJ_SW(c.x); col[9 * 64] = ZJ_SW(c.y); col[10 * 64] = ZJ_SW(c.z); col[11 * 64] = ZJ_SW(c.w);
    col[12 * 64] = ZJ_SW(d.x); col[13 * 64] = ZJ_SW(d.y); col[14 * 64] = ZJ_SW(d.z); col[15 * 64] = ZJ_SW(d.w);
    col[16 * 64] = ZJ_SW(la0); col[17 * 64] = ZJ_SW(la1);
#undef ZJ_SW
}

// One window for the lanes of a wavefront that have `run` set: stage window `win` (index inside the
// segment) of the lane's stream, walk the symbols that start before the window's end or `end`,
// whichever comes first, and store the window's bitmap words.  The reader carries over from window
// to window; `primed` says whether it has been positioned yet.  Returns the number of blocks that
// began in the window.
template <bool CANON>
__device__ __forceinline__ uint32_t zj_window_step(bool run, const uint32_t *stream, uint32_t *bitmap,
                                                   const uint32_t *lut, const ZjCanon *canon, uint32_t selw,
                                                   uint32_t bpm, ZjState &st, uint32_t win, uint32_t end,
                                                   uint32_t *col, ZjBitsLds &bits, bool &primed, ZjLineRegs &kept)
{
    uint32_t n = 0;
    if (run) {
        const uint32_t wbase = win * ZJ_WIN_WORDS;
        // Only a walk's first step (and a lane that joins in later) finds another line in `kept` than the one it is in.
        // The line behind may lie behind the payload or in the next segment's slot -- inside the arena either way (every
        // slot has 96 bytes of slack, which hold the zeros the look-ahead must find, and the arena ends with 1040 more).
        const uint32_t line = win >> 1;
        if (__any(kept.line != line)) zj_load_line(stream, line, kept);
        if ((win & 1u) == 0u) {
            zj_store_window(kept.v[0], kept.v[1], kept.v[2], kept.v[3], kept.v[4].x, kept.v[4].y, col);
        } else {
            ZjLineRegs next;
            zj_load_line(stream, line + 1u, next);
            zj_store_window(kept.v[4], kept.v[5], kept.v[6], kept.v[7], next.v[0].x, next.v[0].y, col);
            kept = next;
        }
        if (!primed) { bits.init(col, wbase, st.p); primed = true; }
        else bits.rebase(col, wbase, st.p);
        const uint32_t wend = (win + 1u) * ZJ_WIN_BITS;
        zj_decode_run_pairs<ZjBitsLds, ZjBitsLds, CANON, true>(bits, lut, canon, selw, bpm, st, wend < end ? wend : end, bits);
        // slots up to the reader's current word hold bitmap words, the ones behind it still stream words
        bits.slot[0] = bits.bm;
        const uint32_t taken = bits.words_done(col) + 1u;
        uint32_t w[ZJ_WIN_WORDS];
#pragma unroll
        for (int i = 0; i < ZJ_WIN_WORDS; i++) w[i] = ((volatile __attribute__((address_space(3))) uint32_t *)col)[i * 64]; // LDS reads, said so: a volatile access through a plain pointer becomes a flat load
#pragma unroll
        for (int i = 0; i < ZJ_WIN_WORDS; i++) {
            w[i] = (uint32_t)i < taken ? w[i] : 0u;
            n += __popc(w[i]);
        }
        uint4 *dst = reinterpret_cast<uint4 *>(bitmap + (size_t)win * ZJ_WIN_WORDS);
        dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
        dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
        dst[2] = make_uint4(w[8], w[9], w[10], w[11]);
        dst[3] = make_uint4(w[12], w[13], w[14], w[15]);
    }
    return n;
}

// ---------------------------------------------------------------------------------- k0: unstuff
// Drops every 00 that follows an FF (src/bitstream.rs:37-50) -- a stream compaction, done in one pass
// over tiles of 15.5 KiB of raw bytes (128 threads x 124 bytes).  A tile counts its kept bytes, gets the
// number of bytes kept before it in the same segment from its predecessors (decoupled look-back: a
// tile publishes its own count first and its inclusive prefix when known, and walks backwards over
// the published words of the tiles before it), compacts into LDS while it waits for them, and copies out
// in aligned 16-byte pieces.  Tiles of a segment are taken in ticket order, so every tile a workgroup waits for is already running.
// The pass is bound by the latency of its dependent steps (ticket, load, publish, look-back, store), so
// what counts is the number of tiles a CU holds at once: LDS allows ten of this size, and two wavefronts
// per tile keep the wavefront slots from running out first.
// A thread's share is 31 words, an odd number: the runs of neighbouring lanes then start 31 banks apart,
// and the 64 lanes of a wavefront that read or write word j of their runs touch every LDS bank twice.
// The unstuffed arena starts zeroed and the same bytes are written on every decode of a batch, so the
// bytes behind a segment's payload stay zero (the reference pads with zeros, src/bitstream.rs:80-84).
// Segments that fit into 3.75 KiB of raw bytes -- short restart intervals, of which an image can hold tens of
// thousands -- take the SINGLE form: one wavefront per segment with a tile of that size (60 bytes per lane),
// no ticket, no look-back and nothing published, so that a CU holds four times as many of them and each
// has fewer dependent steps.
#define ZJ_K0_PAD 16u // the compacted bytes begin here in the LDS buffer
#define ZJ_TS_AGG (1ull << 32)
#define ZJ_TS_INCL (2ull << 32)

template <uint32_t ZJ_K0_THREADS, uint32_t ZJ_K0_WORDS, bool SINGLE>
__global__ __launch_bounds__(ZJ_K0_THREADS) void zj_k0_unstuff(ZjParams P, uint32_t tile0)
{
    constexpr uint32_t ZJ_K0_TILE = ZJ_K0_THREADS * ZJ_K0_WORDS * 4u; // must equal the tile size the batch planner uses
    static_assert(ZJ_K0_TILE == (SINGLE ? ZJ_UNSTUFF_SMALL : ZJ_UNSTUFF_TILE), "tile size of the unstuff pass");
    __shared__ __attribute__((aligned(16))) uint8_t buf[ZJ_K0_TILE + ZJ_K0_PAD + 64]; // first the raw tile, then the compacted bytes
    __shared__ uint32_t wsum[8];
    __shared__ uint32_t s_tile, s_prefix;
    const uint32_t tid = threadIdx.x;
    // The workgroup's segment is fixed by its index; which of the segment's tiles it takes is decided by a
    // ticket per segment (one counter for the whole grid would serialise a quarter of a million atomics
    // on one address).  A workgroup takes tiles until the segment has none left, and asks for the next tile's
    // ticket and bytes while it still works on this one: what limits the pass is how long a tile occupies its
    // share of LDS, and that no longer includes the trips for its ticket and its bytes.
    const uint32_t s = P.tile_seg[tile0 + blockIdx.x];
    const ZjSegDev sg = P.segs[s];
    // everything below is counted in bytes from the 16-byte line at or below the segment's first byte: a segment is shorter
    // than 4 GiB (the planner sees to it), so 32 bits do, and a lane's addresses are one register each
    const uint8_t *raw = P.raw + (sg.raw_begin & ~15ull);
    const uint32_t seg_beg = (uint32_t)(sg.raw_begin & 15ull), seg_end = (uint32_t)(sg.raw_end - (sg.raw_begin & ~15ull));
    constexpr int NB = (int)(ZJ_K0_WORDS * 4u);
    constexpr uint32_t NPC = ZJ_K0_TILE / 16u, NLD = (NPC + ZJ_K0_THREADS - 1u) / ZJ_K0_THREADS;
    const bool nonempty = seg_end > 0u;
    const uint32_t lim = nonempty ? ((seg_end - 1u) & ~15u) : 0u; // offset of the last piece with bytes of the segment
    // The byte before this thread's first byte, and the tile in 16-byte pieces (coalesced) on its way into
    // LDS, where every thread then picks up its words.  The loads are unconditional (addresses clamped
    // into the segment, results masked afterwards) so that all of them are in flight before the first is
    // waited for: one trip to memory per tile, not one per piece.
    uint4 ld[NLD];
    uint32_t pfb;
    auto request = [&](uint32_t tj) {
        const uint32_t my = tj * ZJ_K0_TILE + tid * (ZJ_K0_WORDS * 4u);
        pfb = raw[(my > seg_beg && my <= seg_end) ? my - 1u : 0u];
#pragma unroll
        for (uint32_t i = 0; i < NLD; i++) {
            const uint32_t po = tj * ZJ_K0_TILE + 16u * (tid + i * ZJ_K0_THREADS);
            ld[i] = *reinterpret_cast<const uint4 *>(raw + (po < lim ? po : lim));
        }
    };
    uint32_t tj = 0;
    if (!SINGLE) {
        if (tid == 0) s_tile = atomicAdd(&P.seg_ticket[s], 1u);
        __syncthreads();
        tj = s_tile;
        if (tj >= sg.ntiles) return; // uniform: the segment's tiles are all taken
        __syncthreads(); // everyone has read the ticket before it is replaced below
    }
    request(tj);
    for (;;) {
        // the thread index as a value the compiler takes for new in every turn: what is derived from it is worked out again per
        // tile (a few instructions) instead of being kept in registers across the whole loop (dozens of them)
        uint32_t t = tid;
        asm volatile("" : "+v"(t));
        uint32_t next_ticket = 0; // lane 0: the ticket of the tile after this one, asked for now, passed on at the scan's barriers
        if (!SINGLE && t == 0) next_ticket = atomicAdd(&P.seg_ticket[s], 1u);
        const uint32_t tile = sg.first_tile + tj;
        const uint32_t my = tj * ZJ_K0_TILE + t * (ZJ_K0_WORDS * 4u);
        // bytes of this thread that belong to the segment: [lo, hi) of 0..NB (signed differences, clamped)
        const int64_t dlo = (int64_t)seg_beg - (int64_t)my, dhi = (int64_t)seg_end - (int64_t)my;
        const uint32_t lo = dlo <= 0 ? 0u : (dlo >= NB ? (uint32_t)NB : (uint32_t)dlo);
        const uint32_t hi = dhi <= 0 ? 0u : (dhi >= NB ? (uint32_t)NB : (uint32_t)dhi);
        const bool has_before = dlo < 0 && dhi >= 0;
#pragma unroll
        for (uint32_t i = 0; i < NLD; i++) {
            const uint32_t pc = t + i * ZJ_K0_THREADS;
            const uint32_t po = tj * ZJ_K0_TILE + 16u * pc;
            const bool ok = nonempty && po <= lim;
            const uint4 v = make_uint4(ok ? ld[i].x : 0u, ok ? ld[i].y : 0u, ok ? ld[i].z : 0u, ok ? ld[i].w : 0u);
            if (pc < NPC) reinterpret_cast<uint4 *>(buf)[pc] = v; // unconditional: a store under `ok` makes the compiler wait for each load in turn
        }
        uint32_t pf = has_before && pfb == 0xFFu; // is the byte before this thread's first byte an FF of the segment
        __syncthreads();
        // A wavefront none of whose bytes belong to the segment (most of a tile, when restart intervals are
        // short) only takes part in the barriers and the scan.
        const bool wave_empty = __all(lo >= hi);
        uint32_t w[ZJ_K0_WORDS];
#pragma unroll
        for (uint32_t j = 0; j < ZJ_K0_WORDS; j++) w[j] = 0; // (defined on every path: otherwise the values of the tile before count as live)
        if (!wave_empty) {
#pragma unroll
            for (uint32_t j = 0; j < ZJ_K0_WORDS; j++) w[j] = reinterpret_cast<const uint32_t *>(buf)[t * ZJ_K0_WORDS + j];
        }

        // kill[j]: bit 7 of byte k set when byte 4j+k is not kept (a stuffed 00, or outside the segment)
        uint32_t kill[ZJ_K0_WORDS];
#pragma unroll
        for (uint32_t j = 0; j < ZJ_K0_WORDS; j++) kill[j] = 0;
        uint32_t n;
        if (__all(lo == 0u && hi == (uint32_t)NB)) {
            // whole wavefront inside the segment (all but the first and last tile): word-parallel byte tests
            uint32_t fprev = pf ? 0x80000000u : 0u, ndrop = 0;
#pragma unroll
            for (uint32_t j = 0; j < ZJ_K0_WORDS; j++) {
                const uint32_t x = w[j], y = ~x;
                const uint32_t z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu); // bytes equal to 00
                const uint32_t f = ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu); // bytes equal to FF
                const uint32_t before = __builtin_amdgcn_alignbyte(f, fprev, 3);          // FF flag of the byte before
                kill[j] = z & before;
                ndrop += __popc(kill[j]);
                fprev = f;
            }
            n = (uint32_t)NB - ndrop;
        } else if (!wave_empty) {
            // first and last tile of a segment: the same word-parallel tests, with the bytes outside [lo, hi)
            // masked -- they are not kept, and an FF among them is not "the byte before"
            uint32_t fprev = pf ? 0x80000000u : 0u, ndrop = 0;
#pragma unroll
            for (uint32_t j = 0; j < ZJ_K0_WORDS; j++) {
                const int32_t lj = (int32_t)lo - (int32_t)(4u * j), hj = (int32_t)hi - (int32_t)(4u * j);
                const uint32_t l4 = lj <= 0 ? 0u : (lj >= 4 ? 4u : (uint32_t)lj), h4 = hj <= 0 ? 0u : (hj >= 4 ? 4u : (uint32_t)hj);
                const uint32_t ge = (uint32_t)(0x80808080ull << (8u * l4)), lt = 0x80808080u & ~(uint32_t)(0x80808080ull << (8u * h4));
                const uint32_t inm = ge & lt; // bit 7 of every byte of the word that belongs to the segment
                const uint32_t x = w[j], y = ~x;
                const uint32_t z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);
                const uint32_t f = ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu) & inm;
                const uint32_t before = __builtin_amdgcn_alignbyte(f, fprev, 3);
                kill[j] = (z & before) | (inm ^ 0x80808080u);
                ndrop += __popc(kill[j]);
                fprev = f;
            }
            n = (uint32_t)NB - ndrop;
        } else {
            n = 0;
        }
        if (!SINGLE && t == 0) s_tile = next_ticket;
        uint32_t total;
        const uint32_t inc = block_incl_scan<ZJ_K0_THREADS / 64>(n, wsum, &total); // its barriers also end the reads of the raw tile
        // the tile's own count goes out first, so that the tiles behind it can already add it up
        if (!SINGLE && t == 0 && tj > 0) __hip_atomic_store(&P.tile_status[tile], ZJ_TS_AGG | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!wave_empty) {
            // Compaction in LDS, from byte ZJ_K0_PAD on, before the tile's place in the segment is known (the
            // wait for the predecessors is hidden behind it).  Every byte is stored at the position of the next
            // kept byte: a byte that is not kept is overwritten by the kept byte that follows it in the same
            // thread (LDS stores of a lane are executed in order); only a thread's last byte has no successor
            // and goes to a spare slot.
            uint32_t o = ZJ_K0_PAD + inc - n;
#pragma unroll
            for (uint32_t j = 0; j < ZJ_K0_WORDS; j++) {
#pragma unroll
                for (uint32_t k = 0; k < 4u; k++) {
                    const uint32_t dead = (kill[j] >> (8u * k + 7u)) & 1u;
                    const uint32_t at = (j == ZJ_K0_WORDS - 1u && k == 3u && dead) ? ZJ_K0_TILE + ZJ_K0_PAD + 48u : o;
                    buf[at] = (uint8_t)(w[j] >> (8u * k));
                    o += 1u - dead;
                }
            }
        }
        // the next tile's bytes are asked for now: the registers of the byte tests are free again, and the look-back and the
        // copy below cover the trip
        uint32_t tn = 0;
        bool more = false;
        asm volatile("" ::: "memory"); // not earlier: the loads' registers would overlap those of the byte tests
        __builtin_amdgcn_sched_barrier(0);
        if (!SINGLE) {
            tn = s_tile;
            more = tn < sg.ntiles; // uniform
            if (more) request(tn);
            else {
                pfb = 0; // (defined on every path, like w and kill above)
#pragma unroll
                for (uint32_t i = 0; i < NLD; i++) ld[i] = make_uint4(0, 0, 0, 0);
            }
        }
        if (SINGLE) {
            if (t == 0) { s_prefix = 0; P.seg_payload_bits[s] = total * 8u; }
        } else if (t < 64u) {
            // look-back by the first wavefront: 64 predecessors per step.  The nearest tile that has published
            // an inclusive prefix ends the walk; every tile between it and this one must at least have
            // published its own count, otherwise the step is repeated.
            uint32_t excl = 0;
            if (tj > 0) {
                int64_t base = (int64_t)tile - 1;
                for (;;) {
                    const int64_t j = base - (int64_t)t;
                    uint64_t sv = ZJ_TS_INCL; // in front of the segment's first tile: nothing, and final
                    if (j >= (int64_t)sg.first_tile) sv = __hip_atomic_load(&P.tile_status[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t flag = (uint32_t)(sv >> 32);
                    const uint64_t incl = __ballot(flag == 2u), pending = __ballot(flag == 0u);
                    const uint32_t first = incl ? (uint32_t)__builtin_ctzll(incl) : 64u;
                    const uint64_t needed = first >= 63u ? ~0ull : ((2ull << first) - 1ull); // lanes 0..first
                    if (pending & needed) { __builtin_amdgcn_s_sleep(8); continue; }
                    uint32_t v = (t <= first) ? (uint32_t)sv : 0u;
#pragma unroll
                    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
                    excl += v;
                    if (first < 64u) break;
                    base -= 64;
                }
            }
            if (t == 0) {
                __hip_atomic_store(&P.tile_status[tile], ZJ_TS_INCL | (excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_prefix = excl;
                if (tile + 1u == sg.first_tile + sg.ntiles) P.seg_payload_bits[s] = (excl + total) * 8u;
            }
        }
        __syncthreads();
        // Copy out in 16-byte pieces aligned in global memory.  In piece coordinates (byte 0 = the aligned
        // address at or below the tile's first output byte) the tile owns [shift, shift + total); coordinate c
        // is LDS byte ZJ_K0_PAD + c - shift, so a piece is 16 bytes at one fixed misalignment for the whole
        // tile: two aligned 16-byte LDS reads and a byte funnel shift.
        const uint32_t prefix = s_prefix;
        const uint32_t shift = prefix & 15u;
        uint8_t *out = P.ust + sg.out_off + (prefix - shift);
        const uint32_t endpos = shift + total;
        const uint32_t mis = (uint32_t)__builtin_amdgcn_readfirstlane((int)((ZJ_K0_PAD - shift) & 15u)); // LDS misalignment of a piece
        const uint32_t wo = mis >> 2, bs = mis & 3u;
        for (uint32_t pc = t; pc * 16u < endpos; pc += ZJ_K0_THREADS) {
            const uint32_t b0 = pc * 16u;
            const uint32_t src = ZJ_K0_PAD + b0 - shift; // >= 1
            if (b0 >= shift && b0 + 16u <= endpos) {
                const uint4 lo = *reinterpret_cast<const uint4 *>(buf + (src & ~15u));
                const uint4 hi = *reinterpret_cast<const uint4 *>(buf + (src & ~15u) + 16u);
                uint32_t a0, a1, a2, a3, a4;
                switch (wo) { // uniform
                case 0: a0 = lo.x; a1 = lo.y; a2 = lo.z; a3 = lo.w; a4 = hi.x; break;
                case 1: a0 = lo.y; a1 = lo.z; a2 = lo.w; a3 = hi.x; a4 = hi.y; break;
                case 2: a0 = lo.z; a1 = lo.w; a2 = hi.x; a3 = hi.y; a4 = hi.z; break;
                default: a0 = lo.w; a1 = hi.x; a2 = hi.y; a3 = hi.z; a4 = hi.w; break;
                }
                *reinterpret_cast<uint4 *>(out + b0) = make_uint4(__builtin_amdgcn_alignbyte(a1, a0, bs), __builtin_amdgcn_alignbyte(a2, a1, bs),
                                                                  __builtin_amdgcn_alignbyte(a3, a2, bs), __builtin_amdgcn_alignbyte(a4, a3, bs));
            } else { // a piece shared with the neighbouring tile: only this tile's bytes
                for (uint32_t k = 0; k < 16u; k++)
                    if (b0 + k >= shift && b0 + k < endpos) out[b0 + k] = buf[src + k];
            }
        }
        // The segment's last tile clears the reserved bytes behind the payload: the readers run up to 96 bytes
        // past it and must find zeros there (src/bitstream.rs:80-84), whatever the arena held before.
        if (SINGLE || tile + 1u == sg.first_tile + sg.ntiles) {
            uint8_t *seg_out = P.ust + sg.out_off;
            for (uint32_t i = prefix + total + t; i < sg.out_cap; i += ZJ_K0_THREADS) seg_out[i] = 0;
        }

        if (SINGLE || !more) break;
        tj = tn;
        __syncthreads(); // the buffer is free for the next tile
    }
}

// ---------------------------------------------------------------------------------- k1a: sync
// 256 threads = 4 wavefronts = 4 chunks of 64 subsequences, all with one table set (in LDS), of one image or of neighbours.
// Round 0: every lane walks its own subsequence from the guessed state "a block starts here" and
// leaves one record and one set of bitmap words per window.  Round k: lane i walks on into
// subsequence i+k from the state it carries, replacing the records and bitmap words it passes, until
// its state at a window end equals the record that is already there -- from that point on the two
// walks coincide, so the lane is synchronised and stops.  The wavefront votes (ballot) on whether any
// lane still has to walk.  Subsequences are long compared with the distance a wrong guess survives
// (a few thousand bits on photographs), so the rounds after the first are short.
// Records written by one lane are read by another lane of the same wavefront in a later round: the
// stores are plain (write-through to L2), the loads bypass the per-CU cache, and every window step in
// between waits for all outstanding memory operations of the wavefront.
__device__ __forceinline__ uint32_t zj_load_coherent(const uint32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool CANON>
__device__ __forceinline__ void zj_k1a_body(const ZjParams &P, const ZjImgDev &im, const uint32_t *lut, uint32_t *col, uint32_t chunk,
                                            uint32_t lane)
{
    const ZjCanon *canon = P.canons + (size_t)im.lut_set * 4u;
    ZjSel sel = {im.bpm, im.dcsel, im.acsel};
    const uint32_t selw = zj_selword(sel);

    const uint32_t seg = P.chunk_seg[chunk];
    if (seg == ZJ_NONE) return; // wave-uniform: a padding chunk
    const ZjSegDev sg = P.segs[seg];
    const uint32_t NC = sg.nc, SB = sg.subbits;
    const uint32_t ch = chunk - (sg.first_sub >> 6);        // chunk inside the segment
    uint32_t *rec = P.winfo + sg.first_win + (size_t)ch * 64u * NC; // [window c][lane]
    const uint32_t jl = ch * 64u + lane;
    const uint32_t pay = P.seg_payload_bits[seg];
    const uint32_t *stream = reinterpret_cast<const uint32_t *>(P.ust + sg.out_off);
    uint32_t *bitmap = reinterpret_cast<uint32_t *>(P.bitmap + sg.out_off);
    ZjBitsLds bits;
    bool primed = false;
    ZjLineRegs kept;
    kept.line = ZJ_NONE;
    ZjState st = {jl * SB, 0, 0};
    const bool act0 = sub_active(sg, jl, SB, pay);
    {
        const uint32_t end = act0 ? sub_end(jl, SB, pay) : 0u;
        for (uint32_t c = 0; c < NC; c++) {
            const uint32_t w = jl * NC + c;
            const bool run = act0 && w * ZJ_WIN_BITS < end; // every window that begins inside the payload gets its record and bitmap words
            if (!__any(run)) break;
            const uint32_t n = zj_window_step<CANON>(run, stream, bitmap, lut, canon, selw, sel.bpm, st, w, end, col, bits, primed, kept);
            if (run) rec[c * 64u + lane] = zj_pack_info(st, (w + 1u) * ZJ_WIN_BITS, n);
        }
    }
    bool done = !act0;
    for (uint32_t k = 1; k < 64u; k++) {
        const uint32_t tgt = lane + k, q = jl + k;
        bool act = !done && tgt < 64u && sub_active(sg, q, SB, pay);
        if (!act) done = true;
        if (!__any(act)) break;
        const uint32_t end = act ? sub_end(q, SB, pay) : 0u;
        for (uint32_t c = 0; c < NC; c++) {
            const uint32_t w = q * NC + c;
            const bool run = act && w * ZJ_WIN_BITS < end;
            if (!__any(run)) break;
            const uint32_t n = zj_window_step<CANON>(run, stream, bitmap, lut, canon, selw, sel.bpm, st, w, end, col, bits, primed, kept);
            if (run) {
                const uint32_t nw = zj_pack_info(st, (w + 1u) * ZJ_WIN_BITS, n);
                const uint32_t old = zj_load_coherent(&rec[c * 64u + tgt]);
                rec[c * 64u + tgt] = nw;
                if (((old ^ nw) & ZJ_INFO_STATE_MASK) == 0u) { act = false; done = true; }
            }
        }
    }
}

// A packed chunk: short segments share the wavefront, each with a group of G = 1, 2, 4 .. 32 lanes (the
// same G for the whole chunk).  The lanes of a group do among themselves what the 64 lanes of an ordinary
// chunk do; G = 1 is the plain sequential walk of a restart interval from its known start.  Everything
// that is per segment -- stream, bitmap, record area, sizes -- is per lane here.
template <bool CANON>
__device__ __forceinline__ void zj_k1a_body_packed(const ZjParams &P, const ZjImgDev &im, const uint32_t *lut, uint32_t *col, uint32_t chunk,
                                                   uint32_t seg0, uint32_t G, uint32_t lane)
{
    const ZjCanon *canon = P.canons + (size_t)im.lut_set * 4u;
    ZjSel sel = {im.bpm, im.dcsel, im.acsel};
    const uint32_t selw = zj_selword(sel);

    const uint32_t grp = lane / G, jl = lane % G;
    const uint32_t sidx = seg0 + grp;
    bool have = sidx < P.nseg;
    uint32_t NC = 1, SB = ZJ_WIN_BITS, nsub = 0, pay = 0;
    const uint32_t *stream = nullptr;
    uint32_t *bitmap = nullptr, *rec = nullptr;
    if (have) {
        const ZjSegDev &sg = P.segs[sidx];
        have = sg.lanes == G && sg.first_sub == chunk * 64u + grp * G; // the group exists in this chunk
        if (have) {
            NC = sg.nc; SB = sg.subbits; nsub = sg.nsub;
            pay = P.seg_payload_bits[sidx];
            stream = reinterpret_cast<const uint32_t *>(P.ust + sg.out_off);
            bitmap = reinterpret_cast<uint32_t *>(P.bitmap + sg.out_off);
            rec = P.winfo + sg.first_win; // [window c][lane of the group]
        }
    }
    auto active = [&](uint32_t q) { return have && q < nsub && (uint64_t)q * SB < (uint64_t)pay; };
    auto sub_stop = [&](uint32_t q) { const uint64_t e = (uint64_t)(q + 1u) * SB; return e < pay ? (uint32_t)e : pay; };
    ZjBitsLds bits;
    bool primed = false;
    ZjLineRegs kept;
    kept.line = ZJ_NONE;
    ZjState st = {jl * SB, 0, 0};
    const bool act0 = active(jl);
    {
        const uint32_t end = act0 ? sub_stop(jl) : 0u;
        for (uint32_t c = 0;; c++) {
            const uint32_t w = jl * NC + c;
            const bool run = act0 && c < NC && w * ZJ_WIN_BITS < end;
            if (!__any(run)) break;
            const uint32_t n = zj_window_step<CANON>(run, stream, bitmap, lut, canon, selw, sel.bpm, st, w, end, col, bits, primed, kept);
            if (run) rec[c * G + jl] = zj_pack_info(st, (w + 1u) * ZJ_WIN_BITS, n);
        }
    }
    bool done = !act0;
    for (uint32_t k = 1; k < G; k++) {
        const uint32_t tgt = jl + k;
        bool act = !done && tgt < G && active(tgt);
        if (!act) done = true;
        if (!__any(act)) break;
        const uint32_t end = act ? sub_stop(tgt) : 0u;
        for (uint32_t c = 0;; c++) {
            const uint32_t w = tgt * NC + c;
            const bool run = act && c < NC && w * ZJ_WIN_BITS < end;
            if (!__any(run)) break;
            const uint32_t n = zj_window_step<CANON>(run, stream, bitmap, lut, canon, selw, sel.bpm, st, w, end, col, bits, primed, kept);
            if (run) {
                const uint32_t nw = zj_pack_info(st, (w + 1u) * ZJ_WIN_BITS, n);
                const uint32_t old = zj_load_coherent(&rec[c * G + tgt]);
                rec[c * G + tgt] = nw;
                if (((old ^ nw) & ZJ_INFO_STATE_MASK) == 0u) { act = false; done = true; }
            }
        }
    }
}

__device__ __forceinline__ void load_ptab_to_lds(uint32_t *lds, const uint32_t *src)
{
    const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
    uint4 *d4 = reinterpret_cast<uint4 *>(lds);
    for (uint32_t i = threadIdx.x; i < ZJ_PSET_ENTRIES * 4u / 16u; i += blockDim.x) d4[i] = s4[i];
}

__global__ __launch_bounds__(256) void zj_k1a_sync(ZjParams P)
{
    __shared__ __attribute__((aligned(16))) uint32_t lut[ZJ_PSET_ENTRIES]; // pair entries: up to two symbols a lookup
    __shared__ uint32_t win[4][ZJ_WIN_SLOTS * 64];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t *col = &win[wave][lane];
    const uint32_t chunk = blockIdx.x * 4u + wave;
    const uint32_t seg_first = P.chunk_seg[blockIdx.x * 4u];
    const ZjImgDev im = P.imgs[P.segs[seg_first].img]; // by value: a reference would be read again after every store
    load_ptab_to_lds(lut, P.ptabs + (size_t)im.lut_set * ZJ_PSET_ENTRIES);
    __syncthreads();
    const uint32_t seg = P.chunk_seg[chunk];
    if (seg == ZJ_NONE || P.segs[seg].lanes != 64u) return; // wave-uniform: a padding chunk, or one for the packed kernel
    // the four chunks of a workgroup share a table set, not necessarily an image: layout and selectors are the wavefront's own
    const ZjImgDev imw = P.imgs[P.segs[seg].img];
    if (imw.flags & 4u) zj_k1a_body<true>(P, imw, lut, col, chunk, lane);
    else zj_k1a_body<false>(P, imw, lut, col, chunk, lane);
}

// The same for packed chunks; a kernel of its own so that each form keeps its own register budget
// (launched only when the batch has packed chunks).
__global__ __launch_bounds__(256) void zj_k1a_sync_packed(ZjParams P)
{
    __shared__ __attribute__((aligned(16))) uint32_t lut[ZJ_PSET_ENTRIES];
    __shared__ uint32_t win[4][ZJ_WIN_SLOTS * 64];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t *col = &win[wave][lane];
    const uint32_t chunk = blockIdx.x * 4u + wave;
    bool any = false; // does this workgroup hold a packed chunk at all
    for (uint32_t c = 0; c < 4u; c++) {
        const uint32_t sc = P.chunk_seg[blockIdx.x * 4u + c];
        any = any || (sc != ZJ_NONE && P.segs[sc].lanes != 64u);
    }
    if (!any) return;
    const uint32_t seg_first = P.chunk_seg[blockIdx.x * 4u];
    const ZjImgDev im = P.imgs[P.segs[seg_first].img]; // by value: a reference would be read again after every store
    load_ptab_to_lds(lut, P.ptabs + (size_t)im.lut_set * ZJ_PSET_ENTRIES);
    __syncthreads();
    const uint32_t seg = P.chunk_seg[chunk];
    if (seg == ZJ_NONE) return;
    const uint32_t G = P.segs[seg].lanes;
    if (G == 64u) return;
    const ZjImgDev imw = P.imgs[P.segs[seg].img]; // a packed chunk holds segments of one image
    if (imw.flags & 4u) zj_k1a_body_packed<true>(P, imw, lut, col, chunk, seg, G, lane);
    else zj_k1a_body_packed<false>(P, imw, lut, col, chunk, seg, G, lane);
}

// ---------------------------------------------------------------------------------- k1b: fix
// One lane per chunk (one wavefront per group of four chunks, which share an image and so a table
// set in LDS): starts from the state recorded at the end of the previous chunk and replaces the window
// records and bitmap words of its own chunk until it meets a record that already agrees.
struct ZjEmitCol {
    uint32_t *col;  // bitmap word s of the current window at col[s * 64]
    uint32_t wbase; // stream word of slot 0
    __device__ __forceinline__ void block_start(uint32_t p, uint32_t f)
    {
        atomicOr(&col[((p >> 5) - wbase) * 64u], f << (p & 31u));
    }
};

// PAIRS: walk with pair entries (up to two symbols a lookup, 20 KB of tables per workgroup) or with walk entries (one
// symbol, 6 KB).  A few long walks -- one image, whose fix lanes are a serial tail of the sync pass -- gain from the
// pairs; a large batch has thousands of workgroups whose table loads are most of the pass, and takes the small tables.
template <bool PAIRS>
__global__ __launch_bounds__(64) void zj_k1b_fix(ZjParams P)
{
    typedef typename std::conditional<PAIRS, uint32_t, uint16_t>::type entry_t;
    __shared__ __attribute__((aligned(16))) entry_t lut[PAIRS ? ZJ_PSET_ENTRIES : ZJ_SET_ENTRIES];
    __shared__ uint32_t bmp[ZJ_WIN_WORDS * 64];
    const uint32_t lane = threadIdx.x;
    const uint32_t seg_first = P.chunk_seg[blockIdx.x * 4u];
    const ZjImgDev im = P.imgs[P.segs[seg_first].img]; // by value: a reference would be read again after every store
    if constexpr (PAIRS) load_ptab_to_lds(reinterpret_cast<uint32_t *>(lut), P.ptabs + (size_t)im.lut_set * ZJ_PSET_ENTRIES);
    else load_lut_to_lds(reinterpret_cast<uint16_t *>(lut), P.wluts + (size_t)im.lut_set * ZJ_SET_ENTRIES);
    __syncthreads();
    if (lane >= 4u) return;
    const uint32_t chunk = blockIdx.x * 4u + lane;
    const uint32_t seg = P.chunk_seg[chunk];
    if (seg == ZJ_NONE) return;
    const ZjSegDev sg = P.segs[seg];
    if (sg.lanes != 64u || chunk * 64u == sg.first_sub) return; // packed chunks hold whole segments; a segment's first chunk starts from the true state
    const uint32_t SB = sg.subbits, NC = sg.nc;
    const uint32_t pay = P.seg_payload_bits[seg];
    const uint32_t q0 = chunk * 64u - sg.first_sub;
    if (!sub_active(sg, q0, SB, pay)) return;
    const ZjCanon *canon = P.canons + (size_t)im.lut_set * 4u;
    const ZjImgDev iml = P.imgs[sg.img]; // the chunk's own image: the workgroup shares only the table set
    ZjSel sel = {iml.bpm, iml.dcsel, iml.acsel};
    const uint32_t selw = zj_selword(sel);
    const uint32_t *stream = reinterpret_cast<const uint32_t *>(P.ust + sg.out_off);
    uint32_t *bitmap = reinterpret_cast<uint32_t *>(P.bitmap + sg.out_off);
    uint32_t *rec = P.winfo + sg.first_win;
    ZjState st = zj_unpack_state(rec[zj_rec_index(q0 - 1u, NC - 1u, NC)], q0 * SB);
    ZjBits bits;
    bits.init(stream, st.p);
    ZjEmitCol em = {&bmp[lane], 0};
    for (uint32_t t = 0; t < 64u; t++) {
        const uint32_t q = q0 + t;
        if (!sub_active(sg, q, SB, pay)) return;
        const uint32_t end = sub_end(q, SB, pay);
        for (uint32_t c = 0; c < NC; c++) {
            const uint32_t w = q * NC + c;
            if (w * ZJ_WIN_BITS >= end) break;
            const uint32_t wend = (w + 1u) * ZJ_WIN_BITS;
#pragma unroll
            for (int i = 0; i < ZJ_WIN_WORDS; i++) em.col[i * 64] = 0;
            em.wbase = w * ZJ_WIN_WORDS;
            if constexpr (PAIRS) zj_decode_run_pairs<ZjBits, ZjEmitCol>(bits, reinterpret_cast<const uint32_t *>(lut), canon, selw, sel.bpm, st, wend < end ? wend : end, em);
            else zj_decode_run<ZjBits, ZjEmitCol>(bits, reinterpret_cast<const uint16_t *>(lut), canon, selw, sel.bpm, st, wend < end ? wend : end, em);
            uint32_t bw[ZJ_WIN_WORDS], n = 0;
#pragma unroll
            for (int i = 0; i < ZJ_WIN_WORDS; i++) { bw[i] = em.col[i * 64]; n += __popc(bw[i]); }
            uint4 *dst = reinterpret_cast<uint4 *>(bitmap + (size_t)w * ZJ_WIN_WORDS);
            dst[0] = make_uint4(bw[0], bw[1], bw[2], bw[3]);
            dst[1] = make_uint4(bw[4], bw[5], bw[6], bw[7]);
            dst[2] = make_uint4(bw[8], bw[9], bw[10], bw[11]);
            dst[3] = make_uint4(bw[12], bw[13], bw[14], bw[15]);
            const uint32_t nw = zj_pack_info(st, wend, n);
            const uint32_t idx = zj_rec_index(q, c, NC);
            const uint32_t old = rec[idx];
            rec[idx] = nw;
            if (((old ^ nw) & ZJ_INFO_STATE_MASK) == 0u) return;
        }
    }
    atomicOr(&P.flags[0], 1u); // ran through the whole chunk: the next one may have started from a stale state
    atomicAdd(&P.counters[0], 1u); // the same, counted over the batch's life (never cleared by a decode call)
}

// ---------------------------------------------------------------------------------- k1s: scan
// One workgroup per segment, windows in stream order: wbase[w] = number of blocks that begin before
// window w.  This form is for batches with enough segments (256 or more) to fill the GPU.
__global__ __launch_bounds__(256) void zj_k1s_scan(ZjParams P)
{
    __shared__ uint32_t wsum[8];
    const uint32_t s = blockIdx.x;
    const ZjSegDev sg = P.segs[s];
    const uint32_t NC = sg.nc;
    const uint32_t pay = P.seg_payload_bits[s];
    const uint32_t nwin = min(sg.nsub * NC, (pay + ZJ_WIN_BITS - 1u) / ZJ_WIN_BITS);
    const uint32_t *rec = P.winfo + sg.first_win;
    uint32_t *wb = P.wbase + sg.first_win;
    uint32_t carry = sg.blk_bias;
    if (sg.lanes == 64u && NC >= 32u) {
        // Whole chunks of 64 subsequences of many windows each (with few windows per subsequence the plain steps below
        // are the shorter chain): the records lie [chunk][window of
        // the subsequence][lane] while the result goes out in stream order [lane][window], so a chunk is taken in
        // tiles of 64 windows x 64 lanes through LDS, with the reads and the writes both along their own layout.
        __shared__ uint32_t tl[64 * 65];
        __shared__ uint32_t lane_base[64], part_sum[256];
        __shared__ uint32_t chunk_total;
        const uint32_t lane = threadIdx.x & 63u, part = threadIdx.x >> 6;
        for (uint32_t k = 0; k < (sg.nsub + 63u) / 64u; k++) { // the last chunk may have lanes without a subsequence: their windows lie behind nwin
            const uint32_t *rk = rec + (size_t)k * NC * 64u;
            const uint32_t w_lane = (k * 64u + lane) * NC; // first window of this thread's lane, in stream order
            uint32_t sum = 0;
            for (uint32_t c = part; c < NC; c += 4u)
                if (w_lane + c < nwin) sum += rk[c * 64u + lane] >> 16;
            tl[part * 64u + lane] = sum;
            __syncthreads();
            if (threadIdx.x < 64u) {
                const uint32_t t = tl[lane] + tl[64u + lane] + tl[128u + lane] + tl[192u + lane];
                const uint32_t inc = wave_incl_scan(t, lane);
                lane_base[lane] = carry + inc - t; // blocks that begin before this lane's subsequence
                if (lane == 63u) chunk_total = inc;
            }
            __syncthreads();
            for (uint32_t c0 = 0; c0 < NC; c0 += 64u) {
#pragma unroll 4
                for (uint32_t i = 0; i < 16u; i++) {
                    const uint32_t cl = part * 16u + i, c = c0 + cl;
                    tl[cl * 65u + lane] = (c < NC && w_lane + c < nwin) ? rk[c * 64u + lane] >> 16 : 0u;
                }
                __syncthreads();
                {
                    // the running count along a lane's 64 windows, in four stretches of 16: every thread adds up its
                    // stretch, learns what the stretches before it hold, and walks its stretch again with the count
                    uint32_t mine = 0;
#pragma unroll 4
                    for (uint32_t j = 0; j < 16u; j++) mine += tl[(part * 16u + j) * 65u + lane];
                    part_sum[part * 64u + lane] = mine;
                    __syncthreads();
                    uint32_t run = lane_base[lane];
                    for (uint32_t q = 0; q < part; q++) run += part_sum[q * 64u + lane];
#pragma unroll 4
                    for (uint32_t j = 0; j < 16u; j++) {
                        const uint32_t v = tl[(part * 16u + j) * 65u + lane];
                        tl[(part * 16u + j) * 65u + lane] = run;
                        run += v;
                    }
                    __syncthreads(); // every thread has read lane_base
                    if (part == 3u) lane_base[lane] = run;
                }
                __syncthreads();
#pragma unroll 4
                for (uint32_t i = 0; i < 16u; i++) {
                    const uint32_t l = part + 4u * i, c = c0 + lane; // here a thread's low bits run along the windows
                    const uint32_t wi = (k * 64u + l) * NC + c;
                    if (c < NC && wi < nwin) wb[wi] = tl[lane * 65u + l];
                }
                __syncthreads();
            }
            carry += chunk_total;
        }
    } else {
        // one window per thread and step, neighbouring threads read neighbouring records
        for (uint32_t t0 = 0; t0 < nwin; t0 += 256u) {
            const uint32_t wi = t0 + threadIdx.x;
            uint32_t n = 0;
            if (wi < nwin) {
                const uint32_t q = wi / NC, c = wi - q * NC;
                n = rec[zj_rec_index(q, c, NC, sg.lanes)] >> 16;
            }
            uint32_t total;
            uint32_t inc = block_incl_scan<4>(n, wsum, &total);
            if (wi < nwin) wb[wi] = carry + inc - n;
            carry += total;
        }
    }
    if (threadIdx.x == 0) P.seg_found[s] = carry;
}

// The same for a batch of few segments, possibly one long one (an image without restart markers): the pass is a chain of
// steps with the two barriers of the workgroup scan each, so the workgroup is as large as it can be and a thread takes
// eight consecutive windows per step.
__global__ __launch_bounds__(1024) void zj_k1s_scan_few(ZjParams P)
{
    __shared__ uint32_t wsum[17];
    const uint32_t s = blockIdx.x;
    const ZjSegDev sg = P.segs[s];
    const uint32_t NC = sg.nc;
    const uint32_t pay = P.seg_payload_bits[s];
    const uint32_t nwin = min(sg.nsub * NC, (pay + ZJ_WIN_BITS - 1u) / ZJ_WIN_BITS);
    const uint32_t *rec = P.winfo + sg.first_win;
    uint32_t *wb = P.wbase + sg.first_win;
    uint32_t carry = sg.blk_bias;
    constexpr uint32_t PT = 8u; // windows per thread and step
    for (uint32_t t0 = 0; t0 < nwin; t0 += 1024u * PT) {
        const uint32_t w0 = t0 + threadIdx.x * PT;
        uint32_t n[PT], sum = 0;
        uint32_t q = w0 / NC, c = w0 - q * NC;
#pragma unroll
        for (uint32_t j = 0; j < PT; j++) {
            n[j] = (w0 + j < nwin) ? rec[zj_rec_index(q, c, NC, sg.lanes)] >> 16 : 0u;
            sum += n[j];
            if (++c == NC) { c = 0; q++; }
        }
        uint32_t total;
        uint32_t run = carry + block_incl_scan<16>(sum, wsum, &total) - sum;
#pragma unroll
        for (uint32_t j = 0; j < PT; j++) {
            if (w0 + j < nwin) wb[w0 + j] = run;
            run += n[j];
        }
        carry += total;
    }
    if (threadIdx.x == 0) P.seg_found[s] = carry;
}

// ---------------------------------------------------------------------------------- k1p: positions
// Turns the bitmap words into entries of the block position list.  A workgroup takes 256 windows of a segment; a lane
// takes a quarter of a window (four words) at a time, so that the 64 lanes of a wavefront load 1 KiB side by side, and
// learns how many blocks begin in the quarters before its own from the other three lanes of its quad.
__global__ __launch_bounds__(256) void zj_k1p_positions(ZjParams P)
{
    const uint32_t s = P.ptile_seg[blockIdx.x];
    const ZjSegDev sg = P.segs[s];
    const uint32_t pay = P.seg_payload_bits[s];
    const uint32_t nwin = min(sg.nsub * sg.nc, (pay + ZJ_WIN_BITS - 1u) / ZJ_WIN_BITS);
    const uint32_t w0 = (blockIdx.x - sg.first_ptile) * 256u; // the workgroup's first window
    if (w0 >= nwin) return;
    const uint4 *src = reinterpret_cast<const uint4 *>(P.bitmap + sg.out_off) + (size_t)w0 * 4u;
    const uint32_t *wbase = P.wbase + sg.first_win + w0;
    uint32_t *pos = P.blockpos + sg.first_block;
    const uint32_t quarter = threadIdx.x & 3u;
    uint4 q[4];
    uint32_t g[4];
    // all loads first, unconditional (a quarter behind the last window reads the tile's first one and is ignored)
#pragma unroll
    for (uint32_t it = 0; it < 4u; it++) {
        const uint32_t idx = it * 256u + threadIdx.x;
        const bool ok = w0 + (idx >> 2) < nwin;
        q[it] = src[ok ? idx : 0u];
        g[it] = wbase[ok ? idx >> 2 : 0u];
    }
#pragma unroll
    for (uint32_t it = 0; it < 4u; it++) {
        const uint32_t idx = it * 256u + threadIdx.x;
        const uint32_t wi = w0 + (idx >> 2);
        const bool ok = wi < nwin;
        const uint32_t w[4] = {ok ? q[it].x : 0u, ok ? q[it].y : 0u, ok ? q[it].z : 0u, ok ? q[it].w : 0u};
        const int c = (int)(__popc(w[0]) + __popc(w[1]) + __popc(w[2]) + __popc(w[3]));
        // blocks that begin in quarters 0, 1, 2 of this lane's window (quad_perm broadcasts of one lane of the quad)
        const uint32_t c0 = (uint32_t)__builtin_amdgcn_mov_dpp(c, 0x00, 0xF, 0xF, true);
        const uint32_t c1 = (uint32_t)__builtin_amdgcn_mov_dpp(c, 0x55, 0xF, 0xF, true);
        const uint32_t c2 = (uint32_t)__builtin_amdgcn_mov_dpp(c, 0xAA, 0xF, 0xF, true);
        uint32_t gi = g[it] + (quarter > 0u ? c0 : 0u) + (quarter > 1u ? c1 : 0u) + (quarter > 2u ? c2 : 0u);
        const uint32_t bit0 = wi * ZJ_WIN_BITS + quarter * 128u;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint32_t m = w[i];
            while (m) {
                const uint32_t bit = (uint32_t)__builtin_ctz(m);
                m &= m - 1u;
                if (gi < sg.nblocks) pos[gi] = bit0 + (uint32_t)i * 32u + bit;
                gi++;
            }
        }
    }
}

// ---------------------------------------------------------------------------------- k1c: DC differences
// One lane per block: decode the DC symbol at the recorded block position (src/decoder.rs:194-199
// without the predictor).  Fully parallel; the tables are read through the cache.
#define ZJ_K1C_NB 4 // groups of 256 blocks per workgroup: a lane has that many independent chains of loads in flight
__global__ __launch_bounds__(256) void zj_k1c_dcdiff(ZjParams P, uint32_t ngroups)
{
    // The pass is a chain of dependent trips to memory per block (position, stream words, table entry) with
    // next to no arithmetic, so a lane works on several blocks at once: all positions are requested, then all
    // stream words, then the tables are consulted.  Loads are unconditional, with indices clamped to block 0
    // of the group's image, and what they return is ignored where the block does not exist.
    uint32_t g[ZJ_K1C_NB], pos[ZJ_K1C_NB], tsel[ZJ_K1C_NB], lutset[ZJ_K1C_NB], gs[ZJ_K1C_NB], fnd[ZJ_K1C_NB], first[ZJ_K1C_NB];
    uint64_t soff[ZJ_K1C_NB];
    bool ok[ZJ_K1C_NB], inimg[ZJ_K1C_NB];
#pragma unroll
    for (int k = 0; k < ZJ_K1C_NB; k++) {
        const uint32_t grp = blockIdx.x * ZJ_K1C_NB + k;
        inimg[k] = false; g[k] = 0; tsel[k] = 0; lutset[k] = 0; gs[k] = 0; first[k] = 0;
        uint32_t seg = 0;
        if (grp < ngroups) { // uniform
            const uint32_t ii = P.blk256_img[grp];
            const ZjImgDev &im = P.imgs[ii];
            g[k] = grp * 256u + threadIdx.x;
            const uint32_t gl = g[k] - im.first_block;
            inimg[k] = gl < im.nblocks;
            const uint32_t glc = inimg[k] ? gl : 0u;
            const uint32_t segi = im.ri_blocks ? glc / im.ri_blocks : 0u;
            seg = im.first_seg + segi;
            gs[k] = glc - segi * im.ri_blocks;
            tsel[k] = (im.dcsel >> (gs[k] % im.bpm)) & 1u;
            lutset[k] = im.lut_set;
            first[k] = im.first_block;
        }
        fnd[k] = P.seg_found[seg];
        soff[k] = P.segs[seg].out_off;
    }
    __builtin_amdgcn_sched_barrier(0);
    const uint32_t *stream[ZJ_K1C_NB];
#pragma unroll
    for (int k = 0; k < ZJ_K1C_NB; k++) {
        ok[k] = inimg[k] && gs[k] < fnd[k];
        stream[k] = reinterpret_cast<const uint32_t *>(P.ust + soff[k]);
        pos[k] = P.blockpos[ok[k] ? g[k] : first[k]];
    }
    __builtin_amdgcn_sched_barrier(0); // keep the phases apart: the scheduler would otherwise finish one block's chain before it starts the next
    uint32_t w0[ZJ_K1C_NB], w1[ZJ_K1C_NB];
#pragma unroll
    for (int k = 0; k < ZJ_K1C_NB; k++) {
        const uint32_t p = ok[k] ? pos[k] : 0u;
        w0[k] = stream[k][p >> 5];
        w1[k] = stream[k][(p >> 5) + 1u];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < ZJ_K1C_NB; k++) {
        int32_t d = 0;
        uint32_t nbits = ZJ_DC_BAD;
        if (ok[k]) {
            const uint32_t two[2] = {w0[k], w1[k]};
            d = zj_dc_diff(two, P.luts + (size_t)lutset[k] * ZJ_SET_ENTRIES + tsel[k] * ZJ_TBL_ENTRIES,
                           P.canons + (size_t)lutset[k] * 4u + tsel[k], pos[k] & 31u, &nbits);
        }
        if (inimg[k]) {
            P.dcabs[g[k]] = (int16_t)d; // turned into the predicted value in place by the scan
            P.dcbits[g[k]] = (uint8_t)nbits;
        }
    }
}

// ---------------------------------------------------------------------------------- k1d: DC scan
// dc[g] = sum of the DC differences of the same component up to and including block g, i16 wrapping
// (src/decoder.rs:199-200; the predictors start at 0 in every segment, :376).  A segment is cut into
// tiles of 256 MCUs (one lane per MCU), so that one long segment -- an image without restart markers
// -- is scanned by many workgroups: (1) per tile the sums of its differences per component, (2) per
// segment the exclusive scan of the tile sums, (3) per tile the scan of its own blocks on top of that.
struct ZjDcTile {
    uint32_t d[6], loc[3]; // this lane's differences, and their sums per component
};

__device__ __forceinline__ ZjDcTile zj_dc_tile_load(const int16_t *da, uint32_t m, uint32_t nmcu, uint32_t bpm, uint32_t compsel)
{
    ZjDcTile t;
#pragma unroll
    for (uint32_t cc = 0; cc < 3u; cc++) t.loc[cc] = 0;
#pragma unroll
    for (uint32_t k = 0; k < 6u; k++) t.d[k] = 0;
    if (bpm == 6u && (reinterpret_cast<uintptr_t>(da) & 3u) == 0) {
        // six blocks per MCU: a lane's twelve bytes in three word loads instead of six halfword loads
        if (m < nmcu) {
            const uint32_t *w = reinterpret_cast<const uint32_t *>(da + m * 6u);
            const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
            t.d[0] = (uint32_t)(int32_t)(int16_t)w0; t.d[1] = (uint32_t)((int32_t)w0 >> 16);
            t.d[2] = (uint32_t)(int32_t)(int16_t)w1; t.d[3] = (uint32_t)((int32_t)w1 >> 16);
            t.d[4] = (uint32_t)(int32_t)(int16_t)w2; t.d[5] = (uint32_t)((int32_t)w2 >> 16);
        }
    } else {
#pragma unroll
        for (uint32_t k = 0; k < 6u; k++)
            if (k < bpm && m < nmcu) t.d[k] = (uint32_t)(int32_t)da[m * bpm + k];
    }
#pragma unroll
    for (uint32_t k = 0; k < 6u; k++) {
        const uint32_t c = (compsel >> (2u * k)) & 3u;
#pragma unroll
        for (uint32_t cc = 0; cc < 3u; cc++)
            if (c == cc && k < bpm) t.loc[cc] += t.d[k];
    }
    return t;
}

__device__ __forceinline__ void zj_dc_tile_store(int16_t *da, uint32_t m, uint32_t nmcu, uint32_t bpm, const uint32_t (&v)[6])
{
    if (m >= nmcu) return;
    if (bpm == 6u && (reinterpret_cast<uintptr_t>(da) & 3u) == 0) {
        uint32_t *w = reinterpret_cast<uint32_t *>(da + m * 6u);
        w[0] = v[0] | (v[1] << 16); w[1] = v[2] | (v[3] << 16); w[2] = v[4] | (v[5] << 16);
    } else {
#pragma unroll
        for (uint32_t k = 0; k < 6u; k++)
            if (k < bpm) da[m * bpm + k] = (int16_t)v[k];
    }
}

__global__ __launch_bounds__(256) void zj_k1d_tilesum(ZjParams P)
{
    __shared__ uint32_t wsum[8];
    const uint32_t tile = blockIdx.x;
    const uint32_t s = P.dtile_seg[tile];
    const ZjSegDev sg = P.segs[s];
    if (sg.ndtiles == 1u) return; // nothing before its only tile
    const ZjImgDev im = P.imgs[sg.img]; // by value: a reference would be read again after every store
    const uint32_t nmcu = sg.nblocks / im.bpm;
    const uint32_t m = (tile - sg.first_dtile) * 256u + threadIdx.x;
    const ZjDcTile t = zj_dc_tile_load(P.dcabs + sg.first_block, m, nmcu, im.bpm, im.compsel);
    uint32_t tot[3];
#pragma unroll
    for (uint32_t cc = 0; cc < 3u; cc++) (void)block_incl_scan<4>(t.loc[cc], wsum, &tot[cc]);
    if (threadIdx.x < 3u) P.dtile_sum[(size_t)tile * 4u + threadIdx.x] = tot[threadIdx.x];
}

__global__ __launch_bounds__(256) void zj_k1d_tilescan(ZjParams P)
{
    __shared__ uint32_t wsum[8];
    const uint32_t s = blockIdx.x;
    const ZjSegDev sg = P.segs[s];
    if (sg.ndtiles == 1u) return;
    uint32_t *ts = P.dtile_sum + (size_t)sg.first_dtile * 4u;
    uint32_t carry[3] = {0, 0, 0};
    for (uint32_t t0 = 0; t0 < sg.ndtiles; t0 += 256u) {
        const uint32_t t = t0 + threadIdx.x;
#pragma unroll
        for (uint32_t cc = 0; cc < 3u; cc++) {
            const uint32_t v = t < sg.ndtiles ? ts[(size_t)t * 4u + cc] : 0u;
            uint32_t total;
            const uint32_t inc = block_incl_scan<4>(v, wsum, &total);
            if (t < sg.ndtiles) ts[(size_t)t * 4u + cc] = carry[cc] + inc - v; // in place: sum -> sum of the tiles before
            carry[cc] += total;
        }
    }
}

__global__ __launch_bounds__(256) void zj_k1d_dcscan(ZjParams P)
{
    __shared__ uint32_t wsum[8];
    const uint32_t tile = blockIdx.x;
    const uint32_t s = P.dtile_seg[tile];
    const ZjSegDev sg = P.segs[s];
    const ZjImgDev im = P.imgs[sg.img]; // by value: a reference would be read again after every store
    const uint32_t bpm = im.bpm, compsel = im.compsel;
    const uint32_t nmcu = sg.nblocks / bpm;
    int16_t *da = P.dcabs + sg.first_block; // holds the differences on entry
    const uint32_t m = (tile - sg.first_dtile) * 256u + threadIdx.x;
    const ZjDcTile t = zj_dc_tile_load(da, m, nmcu, bpm, compsel);
    uint32_t run[3];
#pragma unroll
    for (uint32_t cc = 0; cc < 3u; cc++) {
        const uint32_t before = sg.ndtiles == 1u ? 0u : P.dtile_sum[(size_t)tile * 4u + cc];
        uint32_t total;
        const uint32_t inc = block_incl_scan<4>(t.loc[cc], wsum, &total);
        run[cc] = before + inc - t.loc[cc];
    }
    uint32_t out[6];
#pragma unroll
    for (uint32_t k = 0; k < 6u; k++) {
        const uint32_t c = (compsel >> (2u * k)) & 3u;
        uint32_t v = 0;
#pragma unroll
        for (uint32_t cc = 0; cc < 3u; cc++)
            if (c == cc && k < bpm) { run[cc] += t.d[k]; v = run[cc]; }
        out[k] = v & 0xFFFFu;
    }
    zj_dc_tile_store(da, m, nmcu, bpm, out);
}

// The three launches above as one, for batches of many long segments (thousands of images without restart markers): a
// workgroup of 1024 lanes takes a whole segment, four consecutive MCUs per lane and step, so the differences are
// read once and the tile sums never leave the workgroup.
__global__ __launch_bounds__(1024) void zj_k1d_dcscan_seg(ZjParams P)
{
    __shared__ uint32_t wsum[3][16];
    const uint32_t s = blockIdx.x;
    const ZjSegDev sg = P.segs[s];
    const ZjImgDev im = P.imgs[sg.img]; // by value: a reference would be read again after every store
    const uint32_t bpm = im.bpm, compsel = im.compsel;
    const uint32_t nmcu = sg.nblocks / bpm;
    if (nmcu <= 64u) return; // the one-wavefront form takes these
    int16_t *da = P.dcabs + sg.first_block; // holds the differences on entry
    constexpr uint32_t K = 4u;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t carry[3] = {0, 0, 0};
    for (uint32_t m0 = 0; m0 < nmcu; m0 += 1024u * K) {
        const uint32_t m = m0 + threadIdx.x * K;
        ZjDcTile t[K];
        uint32_t loc[3] = {0, 0, 0};
#pragma unroll
        for (uint32_t k = 0; k < K; k++) {
            t[k] = zj_dc_tile_load(da, m + k, nmcu, bpm, compsel);
#pragma unroll
            for (uint32_t cc = 0; cc < 3u; cc++) loc[cc] += t[k].loc[cc];
        }
        uint32_t inc[3];
#pragma unroll
        for (uint32_t cc = 0; cc < 3u; cc++) inc[cc] = wave_incl_scan(loc[cc], lane);
        __syncthreads(); // the sums of the step before have been read
        if (lane == 63u) {
#pragma unroll
            for (uint32_t cc = 0; cc < 3u; cc++) wsum[cc][wave] = inc[cc];
        }
        __syncthreads();
        uint32_t run[3];
#pragma unroll
        for (uint32_t cc = 0; cc < 3u; cc++) {
            uint32_t off = 0, tot = 0;
#pragma unroll
            for (uint32_t w = 0; w < 16u; w++) {
                const uint32_t v = wsum[cc][w];
                if (w < wave) off += v;
                tot += v;
            }
            run[cc] = carry[cc] + off + inc[cc] - loc[cc];
            carry[cc] += tot;
        }
#pragma unroll
        for (uint32_t k = 0; k < K; k++) {
            uint32_t out[6];
#pragma unroll
            for (uint32_t j = 0; j < 6u; j++) {
                const uint32_t c = (compsel >> (2u * j)) & 3u;
                uint32_t v = 0;
#pragma unroll
                for (uint32_t cc = 0; cc < 3u; cc++)
                    if (c == cc && j < bpm) { run[cc] += t[k].d[j]; v = run[cc]; }
                out[j] = v & 0xFFFFu;
            }
            zj_dc_tile_store(da, m + k, nmcu, bpm, out);
        }
    }
}

// The same for segments of at most 64 MCUs (short restart intervals), which the planner lists apart: one
// wavefront per segment, four segments per workgroup, no shared memory and no barrier.
__global__ __launch_bounds__(256) void zj_k1d_dcscan_small(ZjParams P, uint32_t first, uint32_t count)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + (threadIdx.x >> 6)));
    if (idx >= count) return;
    const uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane((int)P.dtile_seg[first + idx]);
    const ZjSegDev sg = P.segs[s];
    const ZjImgDev im = P.imgs[sg.img]; // by value: a reference would be read again after every store
    const uint32_t bpm = im.bpm, compsel = im.compsel;
    const uint32_t nmcu = sg.nblocks / bpm;
    int16_t *da = P.dcabs + sg.first_block; // holds the differences on entry
    const ZjDcTile t = zj_dc_tile_load(da, lane, nmcu, bpm, compsel);
    uint32_t run[3];
#pragma unroll
    for (uint32_t cc = 0; cc < 3u; cc++) run[cc] = wave_incl_scan(t.loc[cc], lane) - t.loc[cc];
    uint32_t out[6];
#pragma unroll
    for (uint32_t k = 0; k < 6u; k++) {
        const uint32_t c = (compsel >> (2u * k)) & 3u;
        uint32_t v = 0;
#pragma unroll
        for (uint32_t cc = 0; cc < 3u; cc++)
            if (c == cc && k < bpm) { run[cc] += t.d[k]; v = run[cc]; }
        out[k] = v & 0xFFFFu;
    }
    zj_dc_tile_store(da, lane, nmcu, bpm, out);
}

// A part of a stream that was split over several GPUs scans its DC differences from its own first block, where the
// predictors are unknown (and what it found in its lead-in is no block at all).  When the parts before it have reported
// their sums (src/decoder.rs:199-200 carries the predictors from MCU to MCU; wrapping i16 addition is associative),
// every value of the blocks it owns is corrected by one constant per component.
__global__ __launch_bounds__(256) void zj_k1d_add_offset(ZjParams P, uint32_t ii, uint32_t g0, uint32_t g1, uint32_t off0, uint32_t off1, uint32_t off2)
{
    const ZjImgDev &im = P.imgs[ii];
    const uint32_t g = g0 + blockIdx.x * 256u + threadIdx.x;
    if (g >= g1) return;
    const uint32_t c = (im.compsel >> (2u * (g % im.bpm))) & 3u;
    const uint32_t off = c == 0u ? off0 : (c == 1u ? off1 : off2);
    int16_t *da = P.dcabs + im.first_block;
    da[g] = (int16_t)((uint32_t)(uint16_t)da[g] + off);
}

// ---------------------------------------------------------------------------------- k1w: write
// One lane per block.  A wavefront owns a run of MCUs and works through it in rounds of 64 blocks of
// one kind -- all luma blocks first, then all chroma blocks -- because luma blocks carry several
// times the symbols of chroma blocks and a round lasts as long as its longest block.  Each lane
// decodes its block into a private 132-byte row of an LDS tile (a row pitch of 33 words keeps lanes
// on distinct banks while they write the same zig-zag index); then the wavefront stores the 64 rows,
// eight lanes per block, one full 128-byte line per block.
// The block's bits are staged in a private LDS column of 14 words (from the word that holds the block's
// first bit, fetched a round ahead), read with a branch-free two-word reader; a block that outgrows its
// column waits for the others and is restaged.  The DC symbol is not decoded again: its length comes from
// the DC pass.
// MCUs per wavefront: chosen by the planner, a multiple of 64 for 4:4:4 and of 32 otherwise (whole rounds);
// long runs amortise a wavefront's start (tables into LDS, two dependent trips to memory before the first symbol).
#ifndef ZJ_K1W_SLOTS
#define ZJ_K1W_SLOTS 14u
#endif
typedef uint32_t zj_u32x4_a4 __attribute__((ext_vector_type(4), aligned(4))); // sixteen bytes at any word address
typedef uint32_t zj_u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));

__device__ __forceinline__ uint32_t zj_lds_addr(const void *p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char *)p; }

// The AC symbols of one block (src/decoder.rs:208-235), one per turn of the loop, for the lanes that are active on
// entry.  State of a lane: w0, w1 = the stream word that holds the current bit and the one behind it (MSB first);
// r = 32 minus the bits of w0 that are used up (0..31: r = 0 says all of w0 is used up and the reader has not yet moved
// on, so that the next 32 bits are always ({w0, w1} >> r), one v_alignbit_b32); si = LDS address of w0's slot in the
// lane's column (pitch 256 bytes); z = zig-zag index of the next coefficient (1..63; 64 and more: the block has ended);
// e = the last symbol's write entry (zj_write_entry).  A lane leaves the loop when its block has ended or when the
// word after w1 would lie behind slot `last`.
// Written in assembly: the compiler's version of the same loop (zj_k1w_symbols_ref) takes 31 vector and 14 scalar
// instructions per symbol and, for a long code, a second trip to LDS that depends on the first; this one takes 24 and 3
// with the table in its direct form (one lookup whatever the code length), 22 and 4 plus the second trip in the
// two-level form.  The fields of an entry are whole bytes that the arithmetic picks up through SDWA operand selects,
// the borrow of the bit count moves the reader on, and the loop mask is one s_and.
// The value of a symbol is worked out one turn late, while the next symbol's table entry is on its way from LDS:
// x = its value bits (on top of the word), e = its entry, a = where its coefficient goes.
#define ZJ_K1W_SDWA(s0, s1) " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:" s0 " src1_sel:" s1 "\n\t"
// EXTEND of the symbol before (src/ec.rs:52-59): with b = leading value bit, the value is the (size + 1)-bit number
// (~b, value bits) sign-extended, plus 1 - b
#define ZJ_K1W_VALUE                                                                                                   \
    "v_lshrrev_b32_e32 %[v], 1, %[x]\n\t"                                                                              \
    "v_bitop3_b32 %[v], %[v], %[x], %[k80] bitop3:0xf2\n\t" /* (x >> 1) | (~x & 0x80000000) */                         \
    "v_ashrrev_i32_sdwa %[v], %[e], %[v]" ZJ_K1W_SDWA("BYTE_3", "DWORD") /* >> (31 - size) */                          \
    "v_ashrrev_i32_e32 %[u], 31, %[x]\n\t"                                                                             \
    "v_add3_u32 %[v], %[v], %[u], 1\n\t"                                                                               \
    "ds_write_b16 %[a], %[v]\n\t"
// what follows the arrival of the entry e2 and of wn, the word after w1
#define ZJ_K1W_STEP                                                                                                    \
    "v_sub_co_u32_sdwa %[r], vcc, %[r], %[e2]" ZJ_K1W_SDWA("DWORD", "BYTE_0") /* r -= bits of the symbol */            \
    "v_cndmask_b32_e32 %[w0], %[w0], %[w1], vcc\n\t"                          /* a borrow moves the reader on */       \
    "v_cndmask_b32_e32 %[w1], %[w1], %[wn], vcc\n\t"                                                                   \
    "v_cndmask_b32_e64 %[t], 0, 1, vcc\n\t"                                                                            \
    "v_lshl_add_u32 %[si], %[t], 8, %[si]\n\t"                                                                         \
    "v_and_b32_e32 %[r], 31, %[r]\n\t"                                                                                 \
    "v_lshlrev_b32_sdwa %[x], %[e2], %[top]" ZJ_K1W_SDWA("BYTE_1", "DWORD") /* value bits on top */                    \
    "v_add_u32_sdwa %[z], %[z], %[e2]" ZJ_K1W_SDWA("DWORD", "BYTE_2")       /* z += run + 1 */                         \
    "v_min_u32_e32 %[t], 0x41, %[z]\n\t"                                    /* index z - 1, or the spare slot 64 */    \
    "v_lshl_add_u32 %[a], %[t], 1, %[rowm2]\n\t"                                                                       \
    "v_mov_b32_e32 %[e], %[e2]\n\t"                                                                                    \
    "v_cmp_gt_u32_e32 vcc, 64, %[z]\n\t"                /* the block goes on */                                        \
    "v_cmp_ge_u32_e64 %[s1], %[last], %[si]\n\t"        /* and the column holds the words of the next turn */          \
    "s_and_b64 exec, vcc, %[s1]\n\t"                                                                                   \
    "s_cbranch_scc1 1b\n\t"                                                                                            \
    "s_mov_b64 exec, %[sx]\n\t"
#define ZJ_K1W_OPERANDS                                                                                                \
    : [w0] "+v"(w0), [w1] "+v"(w1), [r] "+v"(r), [si] "+v"(si), [z] "+v"(z), [e] "+v"(e), [x] "+v"(x), [a] "+v"(a), [top] "=&v"(top),   \
      [t] "=&v"(t), [u] "=&v"(u), [e2] "=&v"(e2), [wn] "=&v"(wn), [v] "=&v"(v), [sx] "=&s"(sx), [s1] "=&s"(s1)

// Two-level form of the table: first level of ZJ_L1_BITS bits, sub-tables behind it (zj_write_entry).
__device__ __forceinline__ void zj_k1w_symbols(uint32_t &w0, uint32_t &w1, uint32_t &r, uint32_t &si, uint32_t &z, uint32_t &e,
                                               uint32_t lut, uint32_t rowm2, uint32_t last)
{
    uint32_t top, t, u, e2, wn, v;
    uint64_t sx, s1;
    const uint32_t k80 = 0x80000000u;
    uint32_t x = 0, a = rowm2 + 130u; // nothing is pending yet: a zero for the spare slot
    e = 31u << 24;
    asm volatile(
        "s_mov_b64 %[sx], exec\n"
        "1:\n\t"
        "v_alignbit_b32 %[top], %[w0], %[w1], %[r]\n\t"                 // the next 32 bits
        "v_lshrrev_b32_e32 %[t], 23, %[top]\n\t"
        "v_lshl_add_u32 %[t], %[t], 2, %[lut]\n\t"
        "ds_read_b32 %[e2], %[t]\n\t"                                   // first-level entry
        "ds_read_b32 %[wn], %[si] offset:512\n\t"                       // the word after w1
        ZJ_K1W_VALUE
        "s_waitcnt lgkmcnt(1)\n\t"                                      // both reads are back (LDS answers in order)
        "v_cmp_gt_i32_e32 vcc, 0, %[e2]\n\t"                            // bit 31: the code is longer than the first level
        "s_cbranch_vccz 2f\n\t"
        "s_and_saveexec_b64 %[s1], vcc\n\t"                             // the lanes with a long code look up their second entry
        "v_bfe_u32 %[u], %[e2], 24, 3\n\t"
        "v_sub_u32_e32 %[t], 23, %[u]\n\t"
        "v_lshrrev_b32_e32 %[t], %[t], %[top]\n\t"
        "v_bfe_u32 %[t], %[t], 0, %[u]\n\t"
        "v_and_b32_e32 %[u], 0x1ff, %[e2]\n\t"
        "v_add_lshl_u32 %[t], %[t], %[u], 2\n\t"
        "v_add_u32_e32 %[t], %[t], %[lut]\n\t"
        "ds_read_b32 %[e2], %[t] offset:2048\n\t"
        "s_mov_b64 exec, %[s1]\n\t"
        "s_waitcnt lgkmcnt(0)\n"
        "2:\n\t"
        ZJ_K1W_STEP
        ZJ_K1W_VALUE                                                    // the value of every lane's last symbol
        ZJ_K1W_OPERANDS
        : [lut] "v"(lut), [rowm2] "v"(rowm2), [last] "v"(last), [k80] "s"(k80)
        : "vcc", "scc", "memory");
}

// Direct form of the table (zj_build_direct_table): one lookup for every code.  t16 = the table's first code longer than
// ZJ_L1_BITS bits, left-aligned in 16 bits (0x10000: none); with d = (next 16 bits) - t16 the entry is at index
// (next ZJ_L1_BITS bits) + max(0, d).
__device__ __forceinline__ void zj_k1w_symbols_direct(uint32_t &w0, uint32_t &w1, uint32_t &r, uint32_t &si, uint32_t &z, uint32_t &e,
                                                      uint32_t lut, uint32_t t16, uint32_t rowm2, uint32_t last)
{
    uint32_t top, t, u, e2, wn, v;
    uint64_t sx, s1;
    const uint32_t k80 = 0x80000000u;
    uint32_t x = 0, a = rowm2 + 130u; // nothing is pending yet: a zero for the spare slot
    e = 31u << 24;
    asm volatile(
        "s_mov_b64 %[sx], exec\n"
        "1:\n\t"
        "v_alignbit_b32 %[top], %[w0], %[w1], %[r]\n\t"                 // the next 32 bits
        "v_sub_u32_sdwa %[u], %[top], %[t16]" ZJ_K1W_SDWA("WORD_1", "DWORD")
        "v_lshrrev_b32_e32 %[t], 23, %[top]\n\t"
        "v_max_i32_e32 %[u], 0, %[u]\n\t"
        "v_add_lshl_u32 %[t], %[t], %[u], 2\n\t"
        "v_add_u32_e32 %[t], %[t], %[lut]\n\t"
        "ds_read_b32 %[e2], %[t]\n\t"                                   // the symbol's entry
        "ds_read_b32 %[wn], %[si] offset:512\n\t"                       // the word after w1
        ZJ_K1W_VALUE
        "s_waitcnt lgkmcnt(1)\n\t"                                      // both reads are back (LDS answers in order)
        ZJ_K1W_STEP
        ZJ_K1W_VALUE                                                    // the value of every lane's last symbol
        ZJ_K1W_OPERANDS
        : [lut] "v"(lut), [t16] "v"(t16), [rowm2] "v"(rowm2), [last] "v"(last), [k80] "s"(k80)
        : "vcc", "scc", "memory");
}

// The same loop in C++, with the first-level entry supplied by `lookup(top)` (a 16-bit table entry, zj_core.h): what the
// assembly above is checked against, and the path of table sets whose long codes need the canonical search.
template <class LOOKUP>
__device__ __forceinline__ void zj_k1w_symbols_ref(uint32_t &w0, uint32_t &w1, uint32_t &r, uint32_t &si, uint32_t &z, uint32_t &e,
                                                   uint32_t rowm2, uint32_t last, LOOKUP lookup)
{
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    typedef __attribute__((address_space(3))) int16_t lds_i16;
    for (;;) {
        const uint32_t top = (uint32_t)((((uint64_t)w0 << 32) | w1) >> r);
        const uint32_t wn = *(const lds_u32 *)(uintptr_t)(si + 512u);
        e = zj_write_entry(lookup(top));
        const uint32_t adv = e & 255u, len = (e >> 8) & 255u, size = 31u - (e >> 24);
        const bool step = r < adv;
        r = (r - adv) & 31u;
        w0 = step ? w1 : w0;
        w1 = step ? wn : w1;
        si += step ? 256u : 0u;
        z += (e >> 16) & 255u;
        *(lds_i16 *)(uintptr_t)(rowm2 + 2u * (z < 65u ? z : 65u)) = (int16_t)zj_value_ac(top << len, size);
        if (!(z < 64u && si <= last)) break;
    }
}

__global__ __launch_bounds__(256) void zj_k1w_write(ZjParams P)
{
    __shared__ __attribute__((aligned(16))) uint32_t lut[ZJ_WT_ENTRIES]; // write entries of this workgroup's AC table; behind them (ZJ_WT_DIRECT) the first level of its DC table, a byte per entry
    __shared__ uint32_t tile[4][64 * 33];
    __shared__ uint32_t scol[4][ZJ_K1W_SLOTS * 64];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t ii = P.k1w_img[blockIdx.x];
    const ZjImgDev im = P.imgs[ii]; // by value: a reference would be read again after every store
    const uint32_t nl = im.hs * im.vs, bpm = im.bpm;
    const ZjCanon *canon = P.canons + (size_t)im.lut_set * 4u + 2u;
    const uint16_t *glut = P.luts + (size_t)im.lut_set * ZJ_SET_ENTRIES + 2u * ZJ_TBL_ENTRIES; // the AC tables in their 16-bit form
    const uint32_t shl = nl == 4u ? 2u : (nl == 2u ? 1u : 0u);     // log2 of the luma blocks per MCU
    const uint32_t M = im.k1w_mcus;                                // MCUs per wavefront
    const uint32_t m0 = im.mcu_lo + ((blockIdx.x - im.first_k1w) * 4u + wave) * M; // first MCU of this wavefront
    const uint32_t nmcu = im.mcu_hi;                                               // end of the MCUs that are wanted
    uint32_t *row32 = &tile[wave][lane * 33u];
    int16_t *row = reinterpret_cast<int16_t *>(row32);
    uint32_t *col = &scol[wave][lane];
    const uint32_t colb = zj_lds_addr(col), rowm2 = zj_lds_addr(row32) - 2u, lutb = zj_lds_addr(lut);
    const uint32_t last = colb + (ZJ_K1W_SLOTS - 3u) * 256u; // the reader needs two words behind its own
    uint32_t status = 0;
    // images without restart markers have one segment: its description is the same for every block
    const bool oneseg = im.ri_blocks == 0u;
    uint32_t seg = im.first_seg, found = 0, pay = 0;
    const uint32_t *stream = nullptr;
    if (oneseg) {
        found = min(P.seg_found[seg], P.segs[seg].nblocks);
        pay = P.seg_payload_bits[seg];
        stream = reinterpret_cast<const uint32_t *>(P.ust + P.segs[seg].out_off);
    }
    // A wavefront takes all luma blocks of its MCUs first (phase 0), then all chroma blocks (phase 1): the workgroup
    // holds one AC table in LDS at a time -- in its direct form a table takes up to ZJ_WT_ENTRIES entries -- and
    // changes it between the phases, which its wavefronts enter together.
    for (uint32_t phase = 0; phase < 2u; phase++) {
        const bool luma = phase == 0;
        if (!luma && bpm == nl) break; // one component: no chroma blocks (uniform)
        const uint32_t tac0 = (im.acsel >> (luma ? 0u : nl)) & 1u;          // the AC table of these blocks
        const bool mixed = !luma && ((im.acsel >> (nl + 1u)) & 1u) != tac0; // Cb and Cr with different AC tables
        const uint32_t t16 = P.wt_info[im.lut_set * 2u + tac0]; // ZJ_WT_TWO_LEVEL, or the parameter of the direct form
        const uint32_t tdc0 = (im.dcsel >> (luma ? 0u : nl)) & 1u;          // the DC table of these blocks (of Cb in the chroma phase)
        const bool new_ac = luma || tac0 != (im.acsel & 1u), new_dc = luma || tdc0 != (im.dcsel & 1u); // uniform: the chroma blocks may use the luma tables
        if (new_ac || new_dc) {
            if (phase) __syncthreads(); // every wavefront is done with the luma tables
            if (new_ac) {
                const uint4 *s4 = reinterpret_cast<const uint4 *>(P.wtabs + (size_t)(im.lut_set * 2u + tac0) * ZJ_WT_ENTRIES);
                uint4 *d4 = reinterpret_cast<uint4 *>(lut);
                for (uint32_t i = threadIdx.x; i < ZJ_WT_DIRECT * 4u / 16u; i += 256u) d4[i] = s4[i];
            }
            if (new_dc && threadIdx.x < (ZJ_WT_ENTRIES - ZJ_WT_DIRECT) * 4u / 16u)
                reinterpret_cast<uint4 *>(lut + ZJ_WT_DIRECT)[threadIdx.x] =
                    reinterpret_cast<const uint4 *>(P.wtabs + (size_t)(im.lut_set * 2u + tdc0) * ZJ_WT_ENTRIES + ZJ_WT_DIRECT)[threadIdx.x];
            __syncthreads();
        }
        // the table sets whose long codes need the canonical search, and rounds whose lanes use two tables, take the
        // loop in C++ with the 16-bit tables in global memory
        const bool use_ref = (im.flags & 2u) != 0 || mixed; // uniform
        const bool direct = t16 != ZJ_WT_TWO_LEVEL;         // uniform
        const uint32_t sh = luma ? shl : 1u;                // log2 of this phase's blocks per MCU
        const uint32_t rounds = m0 < nmcu ? (M << sh) / 64u : 0u; // a wavefront behind the last MCU only takes part in the barriers
        // What a lane needs to know about its block of a round, and the first 56 bytes of the block's bits.
        // Both are fetched ahead of the round that uses them -- the description two rounds ahead, the bits
        // one round ahead (their address comes out of the description) -- so that a round starts decoding
        // at once instead of waiting for two dependent trips to memory.
        // The loads of a description are unconditional (indices clamped to an existing block) and what they
        // return is only looked at through have / has_next in the round that uses it: a load under a
        // condition, or a value converted on the spot, would make the wavefront wait for memory right here.
        struct Desc {
            uint32_t p0, pnext, pay, kb;
            uint32_t g; // the block's index in the batch's arrays
            const uint32_t *stream;
            bool valid, have, has_next; // the block exists / its position is known / so is the next block's
        };
        auto describe = [&](uint32_t r) -> Desc {
            Desc d;
            d.valid = false; d.have = false; d.has_next = false; d.p0 = 0; d.pnext = 0; d.pay = 0; d.kb = 0; d.stream = nullptr; d.g = 0;
            if (r >= rounds) return d; // uniform
            const uint32_t qi = r * 64u + lane;
            const uint32_t mcu = m0 + (qi >> sh);
            d.kb = (luma ? 0u : nl) + (qi & ((1u << sh) - 1u));
            d.valid = mcu < nmcu;
            const uint32_t gl = d.valid ? (mcu - im.mcu_shift) * bpm + d.kb : 0u; // block index in the position list
            const uint32_t g = im.first_block + gl;
            uint32_t gs = gl, fnd = found;
            d.pay = pay;
            d.stream = stream;
            if (!oneseg) { // uniform
                const uint32_t segi = gl / im.ri_blocks;
                const uint32_t sgi = im.first_seg + segi;
                gs = gl - segi * im.ri_blocks;
                fnd = min(P.seg_found[sgi], P.segs[sgi].nblocks);
                d.pay = P.seg_payload_bits[sgi];
                d.stream = reinterpret_cast<const uint32_t *>(P.ust + P.segs[sgi].out_off);
            }
            d.have = d.valid && gs < fnd;
            d.has_next = d.have && gs + 1u < fnd;
            const uint32_t gi = d.have ? g : im.first_block;
            d.g = g;
            d.p0 = P.blockpos[gi];
            d.pnext = P.blockpos[d.has_next ? g + 1u : gi];
            return d;
        };
        // fourteen stream words from word `word` of `src` on
        struct Head {
            zj_u32x4_a4 a, b, c;
            zj_u32x2_a4 d;
        };
        static_assert(ZJ_K1W_SLOTS == 13u || ZJ_K1W_SLOTS == 14u, "the head loads below fetch 13 or 14 words");
        auto fetch_words = [&](const uint32_t *src, uint32_t word, Head &h) {
            const uint32_t *q = src + word;
            h.a = *reinterpret_cast<const zj_u32x4_a4 *>(q);
            h.b = *reinterpret_cast<const zj_u32x4_a4 *>(q + 4);
            h.c = *reinterpret_cast<const zj_u32x4_a4 *>(q + 8);
            if (ZJ_K1W_SLOTS == 14u) h.d = *reinterpret_cast<const zj_u32x2_a4 *>(q + 12);
            else { h.d.x = q[12]; h.d.y = 0; }
        };
        auto fetch_head = [&](const Desc &d, Head &h) {
            // unconditional as well: a lane without a block reads the head of the arena and ignores it
            fetch_words(d.have ? d.stream : reinterpret_cast<const uint32_t *>(P.ust), d.have ? d.p0 >> 5 : 0u, h);
        };
        auto stage = [&](const Head &h) {
    #define ZJ_SW(x) __builtin_bswap32(x)
            col[0 * 64] = ZJ_SW(h.a.x); col[1 * 64] = ZJ_SW(h.a.y); col[2 * 64] = ZJ_SW(h.a.z); col[3 * 64] = ZJ_SW(h.a.w);
            col[4 * 64] = ZJ_SW(h.b.x); col[5 * 64] = ZJ_SW(h.b.y); col[6 * 64] = ZJ_SW(h.b.z); col[7 * 64] = ZJ_SW(h.b.w);
            col[8 * 64] = ZJ_SW(h.c.x); col[9 * 64] = ZJ_SW(h.c.y); col[10 * 64] = ZJ_SW(h.c.z); col[11 * 64] = ZJ_SW(h.c.w);
            col[12 * 64] = ZJ_SW(h.d.x);
            if (ZJ_K1W_SLOTS == 14u) col[13 * 64] = ZJ_SW(h.d.y);
    #undef ZJ_SW
        };
        // Order of memory operations in a round: the description of the round after next is requested at the
        // top; the bits of the next round's block (requested a round ago) are staged into the column at the
        // bottom, after this round's block is done with it and before the coefficient stores are issued, and
        // the request for the bits after those follows at once.  Nothing ever waits for an operation that
        // has not had a whole round to complete -- in particular not for the stores.
        Head hd;
        Desc cur = describe(0);
        fetch_head(cur, hd);
        Desc nxt = describe(1);
        stage(hd);
        fetch_head(nxt, hd);
        for (uint32_t r = 0; r < rounds; r++) {
    #pragma unroll
            for (int i = 0; i < 33; i++) row32[i] = 0;
            const uint32_t qbase = r * 64u;
            const Desc nn = describe(r + 2u);

            // ---- this round's block (src/decoder.rs:186-235; the DC value comes from the prefix scan)
            // The DC symbol (src/decoder.rs:194-199 without the predictor): its bits are the first of the staged column.  The
            // difference goes to the DC array, which the scan behind this pass turns into values; coefficient 0 of the
            // arena keeps the difference, and whoever reads the arena takes the value from the DC array.
            uint32_t dcb = ZJ_DC_BAD;
            int32_t dcdiff = 0;
            if (cur.have) {
                const uint64_t two = ((uint64_t)col[0] << 32) | col[64];
                const uint32_t top = (uint32_t)((two << (cur.p0 & 31u)) >> 32);
                const uint32_t tdc = (im.dcsel >> cur.kb) & 1u;
                const uint32_t e8 = reinterpret_cast<const uint8_t *>(lut + ZJ_WT_DIRECT)[top >> (32 - ZJ_L1_BITS)];
                uint32_t len = e8 & 15u, sz = e8 >> 4;
                if (tdc != tdc0 || e8 == 0u) { // Cr with a DC table of its own, or not a short code: the tables in memory
                    const uint32_t en = zj_lookup(P.luts + (size_t)im.lut_set * ZJ_SET_ENTRIES + tdc * ZJ_TBL_ENTRIES, P.canons + (size_t)im.lut_set * 4u + tdc, top);
                    len = (en >> 8) & 31u;
                    sz = en & 255u;
                }
                if (len != 0u && sz <= 16u) {
                    dcb = len + sz;
                    dcdiff = zj_extend(sz, sz ? ((top << len) >> (32u - sz)) : 0u);
                }
            }
            if (P.dc_in_write && cur.valid) P.dcabs[cur.g] = (int16_t)dcdiff;
            const uint32_t pnext = cur.has_next ? cur.pnext : 0xFFFFFFFFu;
            uint32_t p = cur.p0 + dcb, z = 1, e = ZJ_WE_START; // p >= 1: a DC symbol has at least one bit
            bool live = cur.valid && dcb != ZJ_DC_BAD;            // still has symbols to decode
            if (cur.valid && !live) status |= cur.have ? 1u : 4u; // no DC code / no position: bad code / truncated
            if (live) row[0] = (int16_t)dcdiff;
            uint32_t wa = cur.p0 >> 5; // stream word held in slot 0 of the column
            const uint32_t tac = (im.acsel >> cur.kb) & 1u;
            for (;;) {
                if (live) {
                    // position the reader on bit p: the column holds the 14 words from word wa on
                    const uint32_t pm = p - 1u;
                    uint32_t si = colb + ((pm >> 5) - wa) * 256u;
                    uint32_t w0 = *(const __attribute__((address_space(3))) uint32_t *)(uintptr_t)si;
                    uint32_t w1 = *(const __attribute__((address_space(3))) uint32_t *)(uintptr_t)(si + 256u);
                    uint32_t rr = ~pm & 31u;
    #ifndef ZJ_K1W_NOASM
                    if (!use_ref && direct) zj_k1w_symbols_direct(w0, w1, rr, si, z, e, lutb, t16, rowm2, last);
                    else if (!use_ref) zj_k1w_symbols(w0, w1, rr, si, z, e, lutb, rowm2, last);
                    else
    #endif
                    {
                        const uint16_t *ac = glut + tac * ZJ_TBL_ENTRIES;
                        const ZjCanon *cac = canon + tac;
                        zj_k1w_symbols_ref(w0, w1, rr, si, z, e, rowm2, last, [&](uint32_t top) { return zj_lookup<true, false>(ac, cac, top); });
                    }
                    p = (wa + ((si - colb) >> 8)) * 32u + 32u - rr;
                    live = z < 64u;
                }
                if (!__any(live)) break;
                if (live) { // out of staged bits: restage from the word that holds bit p - 1
                    Head h2;
                    wa = (p - 1u) >> 5;
                    fetch_words(cur.stream, wa, h2);
                    stage(h2);
                }
            }
            if (cur.valid && dcb != ZJ_DC_BAD) {
                uint32_t st = 0;
                if (((e >> 8) & 255u) == 0) st = 1u;                            // ZJ_ST_BAD_CODE
                else if (((e >> 16) & 255u) < 64u && z >= 65u) st = 2u;         // ZJ_ST_COEF_OVERFLOW: a run past coefficient 63
                if (p > cur.pay) st |= 4u;                                      // ZJ_ST_TRUNCATED
                // a cleanly decoded block ends where the next one begins; if it does not, the position
                // list does not describe the sequential decode
                if (st == 0u && pnext != 0xFFFFFFFFu && pnext != p) { atomicOr(&P.flags[1], 1u); atomicAdd(&P.counters[1], 1u); }
                status |= st;
            }
            cur = nxt; nxt = nn;
            // Pin the rotation here, where the loads of this round's description are long complete and no store is
            // in flight: left to itself the compiler copies the registers at the loop's back edge, behind the
            // stores below, and has to wait there for every outstanding memory operation, stores included.
            asm volatile("" : "+v"(nxt.p0), "+v"(nxt.pnext), "+v"(nxt.g));
            // the column is free: the next round's bits go in, and the head registers take the bits after those
            stage(hd);
            fetch_head(nxt, hd);
            zj_wave_lds_sync();
            // store: piece q = it * 64 + lane is sixteen bytes q % 8 of row q / 8.  Eight rows further on the block lies
            // 8 >> sh MCUs further on, so the eight pieces of a lane are a fixed distance apart.
            {
                const uint32_t rl0 = lane >> 3, part = lane & 7u;
                const uint32_t qj0 = qbase + rl0;
                const uint32_t mj0 = m0 + (qj0 >> sh), mstep = 8u >> sh;
                const uint32_t left = mj0 < nmcu ? nmcu - mj0 : 0u; // MCUs from this lane's first row to the end
                const uint32_t gj0 = im.first_block + mj0 * bpm + (luma ? 0u : nl) + (qj0 & ((1u << sh) - 1u));
                int16_t *dst = P.qcoef + (size_t)gj0 * 64u + part * 8u;
                const size_t dstep = (size_t)mstep * bpm * 64u;
                const uint32_t *src = &tile[wave][rl0 * 33u + part * 4u];
#pragma unroll
                for (uint32_t it = 0; it < 8u; it++) {
                    if (it * mstep < left)
                        __builtin_nontemporal_store(zj_u32x4{src[it * 264u], src[it * 264u + 1u], src[it * 264u + 2u], src[it * 264u + 3u]},
                                                    reinterpret_cast<zj_u32x4 *>(dst + it * dstep));
                }
            }
            zj_wave_lds_sync();
        }
    }
    if (status) atomicOr(&P.status[ii], status);
}

// ---------------------------------------------------------------------------------- k2: IDCT + colour
// COS_TABLE of src/dct.rs:19-84 by bit pattern: c[m] = cos(m*pi/16) rounded to f32, m = 1..7.
#define ZJ_C1 0x3f7b14beu
#define ZJ_C2 0x3f6c835eu
#define ZJ_C3 0x3f54db31u
#define ZJ_C4 0x3f3504f3u
#define ZJ_C5 0x3f0e39dau
#define ZJ_C6 0x3ec3ef15u
#define ZJ_C7 0x3e47c5c2u

__device__ __forceinline__ constexpr uint32_t zj_cos_bits(int n, int k)
{
    if (k == 0) return ZJ_C4;
    int m = ((2 * n + 1) * k) & 31;
    bool neg = false;
    if (m > 16) m = 32 - m;
    if (m > 8) { m = 16 - m; neg = true; }
    uint32_t mag = m == 1 ? ZJ_C1 : m == 2 ? ZJ_C2 : m == 3 ? ZJ_C3 : m == 4 ? ZJ_C4 : m == 5 ? ZJ_C5 : m == 6 ? ZJ_C6 : ZJ_C7;
    return neg ? (mag | 0x80000000u) : mag;
}

// src/dct.rs:87-99: out[n] = 0.5 * fma(in7, C[n][7], ... fma(in0, C[n][0], 0.0)), ascending k.
// Outputs n and n + 1 are computed together, two to an instruction (v_pk_fma_f32): each half is the same
// IEEE fused multiply-add, in the same order, as the scalar chain.  Neighbouring outputs share a register
// pair, which is how the colour stage wants its pixels (pairing n with 7 - n cost a re-pairing move per value).
typedef float zj_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void zj_idct_1d(const float (&in)[8], float (&out)[8])
{
#pragma unroll
    for (int n = 0; n < 8; n += 2) {
        zj_f32x2 s = {0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const zj_f32x2 c = {__uint_as_float(zj_cos_bits(n, k)), __uint_as_float(zj_cos_bits(n + 1, k))};
            s = __builtin_elementwise_fma(zj_f32x2{in[k], in[k]}, c, s);
        }
        s = s * zj_f32x2{0.5f, 0.5f};
        out[n] = s.x;
        out[n + 1] = s.y;
    }
}

// Rust `as u8` (src/decoder.rs:275-277): truncate toward zero, saturate, NaN -> 0, packed into byte
// `pos` of `old`.  v_cvt_pk_u8_f32 saturates to 0..255, maps NaN to 0 and rounds to nearest even, so
// the value is truncated first (an integral value is unaffected by the rounding mode).
__device__ __forceinline__ uint32_t zj_pack_u8(float v, uint32_t pos, uint32_t old)
{
    return __builtin_amdgcn_cvt_pk_u8_f32(__builtin_truncf(v), pos, old);
}

// Twenty-four values to twenty-four bytes (six words, byte j of the run from v[j]) with the same cast, without
// the v_trunc_f32 in front of every conversion: v_cvt_pk_u8_f32 rounds by the wavefront's f32 rounding mode
// (checked on the hardware, tools/micro/cvt_pk_u8_rtz.hip), so the mode is switched to round-toward-zero for
// the conversions and back.  One asm statement, so that nothing else can be scheduled between the two
// switches: every other f32 operation of the kernel must run in round-to-nearest-even.
__device__ __forceinline__ void zj_pack24_u8(const float (&v)[24], uint32_t *w)
{
    uint32_t w0, w1, w2, w3, w4, w5;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
                 "s_nop 1\n\t"
                 "v_cvt_pk_u8_f32 %0, %6, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %1, %10, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %2, %14, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %3, %18, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %4, %22, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %5, %26, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %0, %7, 1, %0\n\t"
                 "v_cvt_pk_u8_f32 %1, %11, 1, %1\n\t"
                 "v_cvt_pk_u8_f32 %2, %15, 1, %2\n\t"
                 "v_cvt_pk_u8_f32 %3, %19, 1, %3\n\t"
                 "v_cvt_pk_u8_f32 %4, %23, 1, %4\n\t"
                 "v_cvt_pk_u8_f32 %5, %27, 1, %5\n\t"
                 "v_cvt_pk_u8_f32 %0, %8, 2, %0\n\t"
                 "v_cvt_pk_u8_f32 %1, %12, 2, %1\n\t"
                 "v_cvt_pk_u8_f32 %2, %16, 2, %2\n\t"
                 "v_cvt_pk_u8_f32 %3, %20, 2, %3\n\t"
                 "v_cvt_pk_u8_f32 %4, %24, 2, %4\n\t"
                 "v_cvt_pk_u8_f32 %5, %28, 2, %5\n\t"
                 "v_cvt_pk_u8_f32 %0, %9, 3, %0\n\t"
                 "v_cvt_pk_u8_f32 %1, %13, 3, %1\n\t"
                 "v_cvt_pk_u8_f32 %2, %17, 3, %2\n\t"
                 "v_cvt_pk_u8_f32 %3, %21, 3, %3\n\t"
                 "v_cvt_pk_u8_f32 %4, %25, 3, %4\n\t"
                 "v_cvt_pk_u8_f32 %5, %29, 3, %5\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
                 : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3), "=&v"(w4), "=&v"(w5)
                 : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]),
                   "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]), "v"(v[16]), "v"(v[17]), "v"(v[18]),
                   "v"(v[19]), "v"(v[20]), "v"(v[21]), "v"(v[22]), "v"(v[23]));
    w[0] = w0; w[1] = w1; w[2] = w2; w[3] = w3; w[4] = w4; w[5] = w5;
}

// The same for twelve values (three words): half the live registers around the conversions, one more pair of mode
// switches per block row.
__device__ __forceinline__ void zj_pack12_u8(const float (&v)[12], uint32_t *w)
{
    uint32_t w0, w1, w2;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
                 "s_nop 1\n\t"
                 "v_cvt_pk_u8_f32 %0, %3, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %1, %7, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %2, %11, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %0, %4, 1, %0\n\t"
                 "v_cvt_pk_u8_f32 %1, %8, 1, %1\n\t"
                 "v_cvt_pk_u8_f32 %2, %12, 1, %2\n\t"
                 "v_cvt_pk_u8_f32 %0, %5, 2, %0\n\t"
                 "v_cvt_pk_u8_f32 %1, %9, 2, %1\n\t"
                 "v_cvt_pk_u8_f32 %2, %13, 2, %2\n\t"
                 "v_cvt_pk_u8_f32 %0, %6, 3, %0\n\t"
                 "v_cvt_pk_u8_f32 %1, %10, 3, %1\n\t"
                 "v_cvt_pk_u8_f32 %2, %14, 3, %2\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
                 : "=&v"(w0), "=&v"(w1), "=&v"(w2)
                 : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]),
                   "v"(v[10]), "v"(v[11]));
    w[0] = w0; w[1] = w1; w[2] = w2;
}

// Byte offset (4 * natural index) of the zig-zag positions 8*part .. 8*part+7, one byte each: the
// inverse of the table at src/decoder.rs:114-124, i.e. the zig-zag scan order of T.81 Figure A.6.
#define ZJ_NAT8(a, b, c, d, e, f, g, h)                                                                            \
    ((uint64_t)(4 * a) | ((uint64_t)(4 * b) << 8) | ((uint64_t)(4 * c) << 16) | ((uint64_t)(4 * d) << 24) |        \
     ((uint64_t)(4 * e) << 32) | ((uint64_t)(4 * f) << 40) | ((uint64_t)(4 * g) << 48) | ((uint64_t)(4 * h) << 56))
__constant__ uint64_t zj_zz2nat_bytes[8] = {
    ZJ_NAT8(0, 1, 8, 16, 9, 2, 3, 10),     ZJ_NAT8(17, 24, 32, 25, 18, 11, 4, 5),
    ZJ_NAT8(12, 19, 26, 33, 40, 48, 41, 34), ZJ_NAT8(27, 20, 13, 6, 7, 14, 21, 28),
    ZJ_NAT8(35, 42, 49, 56, 57, 50, 43, 36), ZJ_NAT8(29, 22, 15, 23, 30, 37, 44, 51),
    ZJ_NAT8(58, 59, 52, 45, 38, 31, 39, 46), ZJ_NAT8(53, 60, 61, 54, 47, 55, 62, 63),
};

// Stores through a pointer that was itself read from memory (the table of output buffers): the
// compiler cannot see that it points to global memory and would use flat instructions, which also
// count as LDS operations and so tie every LDS wait to the pixel stores.
__device__ __forceinline__ void zj_store_global16(uint8_t *p, const uint4 &v)
{
    __builtin_nontemporal_store(zj_u32x4{v.x, v.y, v.z, v.w}, (__attribute__((address_space(1))) zj_u32x4 *)p);
}
__device__ __forceinline__ void zj_store_global1(uint8_t *p, uint8_t v) { *(__attribute__((address_space(1))) uint8_t *)p = v; }

typedef unsigned short zj_u16x2 __attribute__((ext_vector_type(2)));

#define ZJ_K2_BLOCKS 96   // blocks per strip of a one-component frame (one workgroup pass); colour layouts: zj_k2_wave_mcus
#define ZJ_K2_WBLOCKS 24  // of which each wavefront takes a quarter and works on it without workgroup barriers
#define ZJ_K2_BSTRIDE 72  // floats per block in LDS: 64 + 8 keeps the column pass conflict-free
// P.k2_ns strips per workgroup; the next strip's coefficients are fetched while this one computes

template <int HS, int VS>
__device__ __forceinline__ void zj_k2_body(const ZjParams &P, const ZjImgDev &im, uint32_t ii, uint32_t strip0,
                                           uint32_t strip1, float *blk_all, const uint16_t *qlds)
{
    constexpr int BPM = HS * VS + 2;
    constexpr int TW = (int)zj_k2_wave_mcus(HS, VS); // MCUs per wavefront: TW * MH = 64 tasks in phase 3
    constexpr int T = 4 * TW;                        // MCUs per strip
    constexpr int WB = TW * BPM;                     // blocks per wavefront: 24 (4:2:0, 4:4:4), 32 (4:2:2), 16 (4:4:0)
    constexpr int NLD = WB / 8;                      // 16-byte pieces of coefficients per lane
    constexpr int MW = 8 * HS, MH = 8 * VS;
    constexpr int ROWB = TW * MW * 3;     // bytes of one pixel row of the wavefront's tile
    constexpr int ROWP = ROWB + 16;       // row pitch of the staged rows: spreads the rows over the banks
    constexpr int NTASK = TW * MH;        // (pixel row, MCU) pairs of the tile
    constexpr int NIT = (NTASK + 63) / 64;
    constexpr int NW = MW * 3 / 4;        // packed RGB words per lane and task
    constexpr int PIECES = ROWB / 16;
    constexpr bool CHROMA_FIRST = HS * VS > 1; // subsampled chroma: its rows are transformed once, ahead of the pixel rows
    // the few descriptor fields of the loop, read once (through the reference they would be read again,
    // and waited for, after every store)
    const uint32_t im_first_block = im.first_block, im_h = im.h, im_mcux = im.mcux, im_strips_x = im.strips_x, im_w = im.w, im_row0 = im.row0;
    // the wavefront's index as a scalar: what depends on it (the tile's place in the image, its coefficients' address, its
    // share of LDS) is then worked out by the scalar unit instead of once per lane
    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float *blk = blk_all + wave * (WB * ZJ_K2_BSTRIDE);
    uint8_t *rgb = P.rgb[ii];
    if (rgb == nullptr) return;
    const size_t stride = P.rgb_stride[ii] ? P.rgb_stride[ii] : (size_t)im_w * 3u;
    const bool vec_ok = (((uintptr_t)rgb | stride) & 15u) == 0;

    const uint32_t part = lane & 7u; // a lane keeps the same eighth of a block in all its tasks
    uint32_t off[8];
    {
        const uint64_t natb = zj_zz2nat_bytes[part];
#pragma unroll
        for (int e = 0; e < 8; e++) off[e] = (uint32_t)(natb >> (8 * e)) & 0xFFu;
    }
    // quantisation values of this lane's eighth, per component, two 16-bit values per word
    const uint4 *qpart = reinterpret_cast<const uint4 *>(qlds) + part;

    // first block of this wavefront's share of a strip, in the batch's arrays
    auto strip_src = [&](uint32_t strip, uint32_t &nbw) -> size_t {
        const uint32_t my = im_row0 + strip / im_strips_x, sx = strip % im_strips_x;
        const uint32_t x0 = sx * T + wave * TW; // first MCU of this wavefront
        const uint32_t nm = x0 < im_mcux ? min((uint32_t)TW, im_mcux - x0) : 0u;
        nbw = nm * BPM;
        return (size_t)im_first_block + ((size_t)my * im_mcux + x0) * BPM;
    };
    // A lane's pieces of the coefficients, and the DC values of the blocks they belong to: coefficient 0 of the arena is
    // the DC difference as the write pass left it, the value after prediction comes from the DC array.
    // A part of a split stream numbers its blocks from its own first one (the shift is 0 otherwise), and the MCUs of its first
    // and last row that belong to its neighbours have no place in its DC array at all: their index is checked, not trusted.
    const uint32_t dc_shift = im.mcu_shift * BPM, im_nblocks = im.nblocks;
    const uint16_t *dc_img = reinterpret_cast<const uint16_t *>(P.dcabs) + im_first_block;
    auto load3 = [&](size_t blk0, uint32_t nbw, uint4 (&v)[NLD], uint32_t &dcv) {
        const int16_t *src = P.qcoef + blk0 * 64u;
        const uint32_t rel0 = (uint32_t)(blk0 - im_first_block) - dc_shift; // wraps for a block before the part's first
#pragma unroll
        for (uint32_t i = 0; i < (uint32_t)NLD; i++) {
            const uint32_t q = lane + i * 64u;
            v[i] = make_uint4(0, 0, 0, 0);
            if ((q >> 3) < nbw) {
                const zj_u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const zj_u32x4 *>(src + (size_t)q * 8u));
                v[i] = make_uint4(t.x, t.y, t.z, t.w);
            }
        }
        // lane L: the DC value of the wavefront's block L
        dcv = 0;
        const uint32_t rel = rel0 + lane;
        if (lane < nbw && rel < im_nblocks) dcv = dc_img[rel];
    };
    // what lane L does with it: its block's DC term, dequantised like every other coefficient, goes over the one that the
    // scatter of phase 1 made of the difference (natural position 0 of block L)
    const uint32_t dc_kb = lane % BPM;
    const uint32_t dc_q = qlds[(dc_kb < (uint32_t)(HS * VS) ? 0u : (dc_kb == (uint32_t)(HS * VS) ? 1u : 2u)) * 64u];

    uint4 v[NLD];
    uint32_t dcv;
    uint32_t nbw;
    {
        const size_t blk0 = strip_src(strip0, nbw);
        load3(blk0, nbw, v, dcv);
    }
    for (uint32_t strip = strip0; strip < strip1; strip++) {
        const uint32_t my = im_row0 + strip / im_strips_x, sx = strip % im_strips_x;
        const uint32_t x0 = sx * T + wave * TW;
        // phase 1: dequantise (i16 wrapping, both operands in zig-zag order, src/decoder.rs:179-184) two
        // coefficients per instruction, de-zigzag into LDS as f32 (exact), src/decoder.rs:296-300
#pragma unroll
        for (uint32_t i = 0; i < (uint32_t)NLD; i++) {
            const uint32_t b = (lane + i * 64u) >> 3;
            const uint32_t kb = b % BPM;
            const uint32_t comp = kb < (uint32_t)(HS * VS) ? 0u : (kb == (uint32_t)(HS * VS) ? 1u : 2u);
            const uint4 qv = qpart[comp * 8u];
            const uint32_t cw[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
            const uint32_t qw[4] = {qv.x, qv.y, qv.z, qv.w};
            char *bp = reinterpret_cast<char *>(blk + b * ZJ_K2_BSTRIDE);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const zj_u16x2 pr = __builtin_bit_cast(zj_u16x2, cw[j]) * __builtin_bit_cast(zj_u16x2, qw[j]);
                *reinterpret_cast<float *>(bp + off[2 * j]) = (float)(int16_t)pr.x;
                *reinterpret_cast<float *>(bp + off[2 * j + 1]) = (float)(int16_t)pr.y;
            }
        }
        if (lane < (uint32_t)WB) blk[lane * ZJ_K2_BSTRIDE] = (float)(int16_t)(uint16_t)(dcv * dc_q); // (LDS stores of a wavefront land in program order)
        // fetch the next strip's coefficients now; they are consumed in the next iteration
        if (strip + 1 < strip1) {
            const size_t blk0 = strip_src(strip + 1, nbw);
            load3(blk0, nbw, v, dcv);
        }
        zj_wave_lds_sync();
        // phase 2: 1-D IDCT over the columns of every block, in place (src/dct.rs:104-111)
#pragma unroll
        for (uint32_t i = 0; i < (uint32_t)NLD; i++) {
            const uint32_t q = lane + i * 64u;
            const uint32_t b = q >> 3, col = q & 7u;
            float *bp = blk + b * ZJ_K2_BSTRIDE + col;
            float in[8], out[8];
#pragma unroll
            for (int k = 0; k < 8; k++) in[k] = bp[k * 8];
            zj_idct_1d(in, out);
#pragma unroll
            for (int n = 0; n < 8; n++) bp[n * 8] = out[n];
        }
        zj_wave_lds_sync();
        // phase 3: 1-D IDCT over the rows (src/dct.rs:113-120), +128, replicate chroma, colour
        // (src/decoder.rs:306-312).
        uint32_t px[NIT][NW];
        if constexpr (CHROMA_FIRST) {
            // Subsampled chroma: a chroma row serves several pixel rows (and both halves of a wide MCU), so the
            // rows of the chroma blocks are transformed once, in place, one lane per (chroma block, row) -- the
            // same chain on the same operands as every pixel-row task would otherwise repeat for itself.
#pragma unroll
            for (int it = 0; it < (2 * TW * 8 + 63) / 64; it++) {
                const uint32_t q = lane + it * 64u;
                const uint32_t cblk = q >> 3, row = q & 7u;
                float *rp = blk + ((cblk >> 1) * BPM + HS * VS + (cblk & 1u)) * ZJ_K2_BSTRIDE + row * 8u;
                if (q < (uint32_t)(2 * TW * 8)) {
                    float in[8], out[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) in[k] = rp[k];
                    zj_idct_1d(in, out);
#pragma unroll
                    for (int k = 0; k < 8; k++) rp[k] = out[k];
                }
            }
            zj_wave_lds_sync();
            // One lane per (pixel row r, MCU m) of the tile, one block column u of the MCU at a time (the same for
            // every lane, so that the chroma columns of the block are the same registers in every lane).
            static_assert(NTASK == 64 && NIT == 1, "one (pixel row, MCU) task per lane");
            const uint32_t m = lane / MH, r = lane % MH;
            constexpr int NCS = 8 / HS; // chroma samples under the eight pixels of a block row
#pragma unroll
            for (int u = 0; u < HS; u++) {
                float cb[NCS], cr[NCS], in[8], yv[8];
                const float *cbp = blk + (m * BPM + HS * VS) * ZJ_K2_BSTRIDE + (r / VS) * 8u + u * NCS;
                const float *yp = blk + (m * BPM + (r / 8u) * HS + u) * ZJ_K2_BSTRIDE + (r % 8u) * 8u;
#pragma unroll
                for (int k = 0; k < NCS; k++) { cb[k] = cbp[k]; cr[k] = cbp[ZJ_K2_BSTRIDE + k]; }
#pragma unroll
                for (int k = 0; k < 8; k++) in[k] = yp[k];
                zj_idct_1d(in, yv);
                // per chroma sample: the products of src/decoder.rs:271-273, computed once and shared by the
                // pixels that replicate the sample.  (x + 128) - 128 is kept as two operations.
                float pr[NCS], pg[NCS], pb[NCS];
#pragma unroll
                for (int c = 0; c < NCS; c++) {
                    const float cbd = (cb[c] + 128.0f) - 128.0f;
                    const float crd = (cr[c] + 128.0f) - 128.0f;
                    pr[c] = 1.402f * crd;
                    pg[c] = -0.71414f * crd + -0.34414f * cbd;
                    pb[c] = 1.772f * cbd;
                }
#ifdef ZJ_K2_PACK12
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    float rgbv[12]; // four pixels of this block row, r g b each
#pragma unroll
                    for (int x = 4 * h; x < 4 * h + 4; x += 2) {
                        const int cx0 = x / HS, cx1 = (x + 1) / HS, o = 3 * (x - 4 * h);
                        const zj_f32x2 y2 = zj_f32x2{yv[x], yv[x + 1]} + zj_f32x2{128.0f, 128.0f};
                        const zj_f32x2 r2 = zj_f32x2{pr[cx0], pr[cx1]} + y2;
                        const zj_f32x2 g2 = zj_f32x2{pg[cx0], pg[cx1]} + y2;
                        const zj_f32x2 b2 = zj_f32x2{pb[cx0], pb[cx1]} + y2;
                        rgbv[o + 0] = r2.x; rgbv[o + 1] = g2.x; rgbv[o + 2] = b2.x;
                        rgbv[o + 3] = r2.y; rgbv[o + 4] = g2.y; rgbv[o + 5] = b2.y;
                    }
                    zj_pack12_u8(rgbv, &px[0][u * 6 + h * 3]); // bytes u*24 + h*12 .. +11 of the lane's run
                }
#else
                float rgbv[24]; // the eight pixels of this block row, r g b each
#pragma unroll
                for (int x = 0; x < 8; x += 2) {
                    const int cx0 = x / HS, cx1 = (x + 1) / HS;
                    const zj_f32x2 y2 = zj_f32x2{yv[x], yv[x + 1]} + zj_f32x2{128.0f, 128.0f};
                    const zj_f32x2 r2 = zj_f32x2{pr[cx0], pr[cx1]} + y2;
                    const zj_f32x2 g2 = zj_f32x2{pg[cx0], pg[cx1]} + y2;
                    const zj_f32x2 b2 = zj_f32x2{pb[cx0], pb[cx1]} + y2;
                    rgbv[3 * x + 0] = r2.x; rgbv[3 * x + 1] = g2.x; rgbv[3 * x + 2] = b2.x;
                    rgbv[3 * x + 3] = r2.y; rgbv[3 * x + 4] = g2.y; rgbv[3 * x + 5] = b2.y;
                }
                zj_pack24_u8(rgbv, &px[0][u * 6]); // bytes u*24 .. u*24+23 of the lane's run
#endif
            }
        } else {
            // One lane per (pixel row r, MCU m) of the tile.
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const uint32_t q = lane + it * 64u;
            const uint32_t m = q / MH, r = q % MH;
#pragma unroll
            for (int w4 = 0; w4 < NW; w4++) px[it][w4] = 0;
            if (q < (uint32_t)NTASK) {
                float cb[8], cr[8], in[8];
                const float *cbp = blk + (m * BPM + HS * VS) * ZJ_K2_BSTRIDE + (r / VS) * 8u;
#pragma unroll
                for (int k = 0; k < 8; k++) in[k] = cbp[k];
                zj_idct_1d(in, cb);
#pragma unroll
                for (int k = 0; k < 8; k++) in[k] = cbp[ZJ_K2_BSTRIDE + k];
                zj_idct_1d(in, cr);
                // per chroma sample: the products of src/decoder.rs:271-273, computed once and shared by the
                // pixels that replicate the sample.  (x + 128) - 128 is kept as two operations.
                float pr[8], pg[8], pb[8];
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    const float cbd = (cb[c] + 128.0f) - 128.0f;
                    const float crd = (cr[c] + 128.0f) - 128.0f;
                    pr[c] = 1.402f * crd;
                    pg[c] = -0.71414f * crd + -0.34414f * cbd;
                    pb[c] = 1.772f * cbd;
                }
#pragma unroll
                for (int u = 0; u < HS; u++) {
                    float yv[8];
                    const float *yp = blk + (m * BPM + (r / 8u) * HS + u) * ZJ_K2_BSTRIDE + (r % 8u) * 8u;
#pragma unroll
                    for (int k = 0; k < 8; k++) in[k] = yp[k];
                    zj_idct_1d(in, yv);
                    float rgbv[24]; // the eight pixels of this block row, r g b each
#pragma unroll
                    for (int x = 0; x < 8; x += 2) {
                        const int cx0 = (u * 8 + x) / HS, cx1 = (u * 8 + x + 1) / HS;
                        const zj_f32x2 y2 = zj_f32x2{yv[x], yv[x + 1]} + zj_f32x2{128.0f, 128.0f};
                        const zj_f32x2 r2 = zj_f32x2{pr[cx0], pr[cx1]} + y2;
                        const zj_f32x2 g2 = zj_f32x2{pg[cx0], pg[cx1]} + y2;
                        const zj_f32x2 b2 = zj_f32x2{pb[cx0], pb[cx1]} + y2;
                        rgbv[3 * x + 0] = r2.x; rgbv[3 * x + 1] = g2.x; rgbv[3 * x + 2] = b2.x;
                        rgbv[3 * x + 3] = r2.y; rgbv[3 * x + 4] = g2.y; rgbv[3 * x + 5] = b2.y;
                    }
                    zj_pack24_u8(rgbv, &px[it][u * 6]); // bytes u*24 .. u*24+23 of the lane's run
                }
            }
        }
        }
        zj_wave_lds_sync();
        // phase 4: RGB rows of the tile into LDS (reusing the block area), then a linear copy per pixel
        // row in 16-byte pieces (src/decoder.rs:314-315 writes the same bytes pixel by pixel)
        char *stage = reinterpret_cast<char *>(blk);
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const uint32_t q = lane + it * 64u;
            const uint32_t m = q / MH, r = q % MH;
            if (q < (uint32_t)NTASK) {
                char *d = stage + r * ROWP + m * (MW * 3);
                if constexpr (HS == 2) {
#pragma unroll
                    for (int w4 = 0; w4 < NW; w4 += 4)
                        *reinterpret_cast<uint4 *>(d + 4 * w4) = make_uint4(px[it][w4], px[it][w4 + 1], px[it][w4 + 2], px[it][w4 + 3]);
                } else {
#pragma unroll
                    for (int w4 = 0; w4 < NW; w4 += 2) *reinterpret_cast<uint2 *>(d + 4 * w4) = make_uint2(px[it][w4], px[it][w4 + 1]);
                }
            }
        }
        zj_wave_lds_sync();
        // The next strip's coefficients, requested in phase 1, have long arrived: take them over here, before
        // the pixel stores are issued.  Otherwise the first use at the top of the next iteration is where the
        // wait goes, and there it also covers the stores just issued.
#pragma unroll
        for (int i = 0; i < NLD; i++) asm volatile("" : "+v"(v[i].x), "+v"(v[i].y), "+v"(v[i].z), "+v"(v[i].w));
        asm volatile("" : "+v"(dcv));
        const uint32_t xb = x0 * MW * 3u; // byte offset of the tile inside the pixel row
        const uint32_t row_valid = min((uint32_t)ROWB, (im_w - min(im_w, x0 * MW)) * 3u); // bytes inside the image
        const uint32_t rows_valid = min((uint32_t)MH, im_h - min(im_h, my * MH));
        for (uint32_t q = lane; q < (uint32_t)(PIECES * MH); q += 64u) {
            const uint32_t r = q / PIECES, pc = q % PIECES;
            const uint32_t o = pc * 16u;
            if (r >= rows_valid || o >= row_valid) continue;
            uint8_t *dst = rgb + (size_t)((my - im_row0) * MH + r) * stride + xb + o;
            const uint4 pv = *reinterpret_cast<const uint4 *>(stage + r * ROWP + o);
            if (vec_ok && o + 16u <= row_valid) {
                zj_store_global16(dst, pv);
            } else {
                const uint32_t s4[4] = {pv.x, pv.y, pv.z, pv.w};
                const uint32_t nbytes = min(16u, row_valid - o);
                for (uint32_t i = 0; i < nbytes; i++) zj_store_global1(dst + i, (uint8_t)(s4[i >> 2] >> (8u * (i & 3u))));
            }
        }
        zj_wave_lds_sync(); // the block area is reused by the next strip
    }
}

// One instantiation per sampling layout, so that each gets its own register allocation.  `first`
// is where this layout's workgroups begin in P.strip_img.
#ifndef ZJ_K2_WAVES
#define ZJ_K2_WAVES 1 // minimum wavefronts per SIMD the register allocation must leave room for
#endif
template <int HS, int VS>
__global__ __launch_bounds__(256, ZJ_K2_WAVES) void zj_k2_idct_rgb(ZjParams P, uint32_t first)
{
    __shared__ __attribute__((aligned(16))) float blk[4u * zj_k2_wave_mcus(HS, VS) * (HS * VS + 2) * ZJ_K2_BSTRIDE];
    __shared__ __attribute__((aligned(16))) uint16_t qlds[3 * 64];
    const uint32_t grp = P.strip_img[first + blockIdx.x]; // image in the high bits is looked up below
    const uint32_t ii = P.strip_img_img[first + blockIdx.x];
    const ZjImgDev &im = P.imgs[ii]; // a reference on purpose: a copy costs this kernel registers and with them a wavefront per SIMD
    if (threadIdx.x < 192u) qlds[threadIdx.x] = P.qts[im.qt_off[threadIdx.x >> 6] + (threadIdx.x & 63u)];
    __syncthreads();
    const uint32_t nstrips = im.strips_x * im.nrows;
    const uint32_t s0 = grp * P.k2_ns;
    const uint32_t s1 = min(s0 + P.k2_ns, nstrips);
    zj_k2_body<HS, VS>(P, im, ii, s0, s1, blk, qlds);
}

// One-component frames: every MCU is a single block and the pixels are those of a colour image
// with Cb = Cr = 128, where src/decoder.rs:271-273 gives r = 1.402f * 0 + y = y and likewise g and b:
// the three bytes of a pixel are the same truncated, saturated luma value.
__global__ __launch_bounds__(256) void zj_k2_idct_gray(ZjParams P, uint32_t first)
{
    constexpr int T = ZJ_K2_BLOCKS, TW = T / 4;   // blocks (= MCUs) per strip and per wavefront
    constexpr int ROWB = TW * 24, ROWP = ROWB + 16, PIECES = ROWB / 16;
    __shared__ __attribute__((aligned(16))) float blk_all[ZJ_K2_BLOCKS * ZJ_K2_BSTRIDE];
    __shared__ __attribute__((aligned(16))) uint16_t qlds[64];
    const uint32_t grp = P.strip_img[first + blockIdx.x];
    const uint32_t ii = P.strip_img_img[first + blockIdx.x];
    const ZjImgDev &im = P.imgs[ii]; // a reference on purpose: a copy costs this kernel registers and with them a wavefront per SIMD
    if (threadIdx.x < 64u) qlds[threadIdx.x] = P.qts[im.qt_off[0] + threadIdx.x];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    float *blk = blk_all + wave * (ZJ_K2_WBLOCKS * ZJ_K2_BSTRIDE);
    uint8_t *rgb = P.rgb[ii];
    if (rgb == nullptr) return;
    const size_t stride = P.rgb_stride[ii] ? P.rgb_stride[ii] : (size_t)im.w * 3u;
    const bool vec_ok = (((uintptr_t)rgb | stride) & 15u) == 0;
    const uint32_t part = lane & 7u;
    uint32_t off[8];
    {
        const uint64_t natb = zj_zz2nat_bytes[part];
#pragma unroll
        for (int e = 0; e < 8; e++) off[e] = (uint32_t)(natb >> (8 * e)) & 0xFFu;
    }
    const uint4 qv = reinterpret_cast<const uint4 *>(qlds)[part];
    const uint32_t qw[4] = {qv.x, qv.y, qv.z, qv.w};
    const uint32_t nstrips = im.strips_x * im.nrows;
    const uint32_t s0 = grp * P.k2_ns, s1 = min(s0 + P.k2_ns, nstrips);
    for (uint32_t strip = s0; strip < s1; strip++) {
        const uint32_t my = im.row0 + strip / im.strips_x, sx = strip % im.strips_x;
        const uint32_t x0 = sx * T + wave * TW; // first block of this wavefront in the block row
        const uint32_t nbw = x0 < im.mcux ? min((uint32_t)TW, im.mcux - x0) : 0u;
        const int16_t *src = P.qcoef + ((size_t)im.first_block + (size_t)my * im.mcux + x0) * 64u;
        // dequantise and de-zigzag into LDS (src/decoder.rs:179-184, :296-300)
#pragma unroll
        for (uint32_t i = 0; i < 3u; i++) {
            const uint32_t q = lane + i * 64u, b = q >> 3;
            uint4 v = make_uint4(0, 0, 0, 0);
            uint32_t dc = 0; // the block's DC value: coefficient 0 of the arena holds the difference
            if (b < nbw) {
                v = *reinterpret_cast<const uint4 *>(src + (size_t)q * 8u);
                const uint32_t rel = my * im.mcux + x0 + b - im.mcu_shift; // wraps for a block before the first of a part of a split stream
                if (rel < im.nblocks) dc = reinterpret_cast<const uint16_t *>(P.dcabs)[im.first_block + rel];
            }
            const uint32_t cw[4] = {part == 0u ? ((v.x & 0xFFFF0000u) | dc) : v.x, v.y, v.z, v.w};
            char *bp = reinterpret_cast<char *>(blk + b * ZJ_K2_BSTRIDE);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const zj_u16x2 pr = __builtin_bit_cast(zj_u16x2, cw[j]) * __builtin_bit_cast(zj_u16x2, qw[j]);
                *reinterpret_cast<float *>(bp + off[2 * j]) = (float)(int16_t)pr.x;
                *reinterpret_cast<float *>(bp + off[2 * j + 1]) = (float)(int16_t)pr.y;
            }
        }
        zj_wave_lds_sync();
        // columns in place (src/dct.rs:104-111)
#pragma unroll
        for (uint32_t i = 0; i < 3u; i++) {
            const uint32_t q = lane + i * 64u;
            float *bp = blk + (q >> 3) * ZJ_K2_BSTRIDE + (q & 7u);
            float in[8], out[8];
#pragma unroll
            for (int k = 0; k < 8; k++) in[k] = bp[k * 8];
            zj_idct_1d(in, out);
#pragma unroll
            for (int n = 0; n < 8; n++) bp[n * 8] = out[n];
        }
        zj_wave_lds_sync();
        // rows (src/dct.rs:113-120), +128, three equal bytes per pixel: one lane per (block, row)
        uint32_t px[3][6];
#pragma unroll
        for (uint32_t i = 0; i < 3u; i++) {
            const uint32_t q = lane + i * 64u;
            const float *bp = blk + (q >> 3) * ZJ_K2_BSTRIDE + (q & 7u) * 8u;
            float in[8], y[8];
#pragma unroll
            for (int k = 0; k < 8; k++) in[k] = bp[k];
            zj_idct_1d(in, y);
#pragma unroll
            for (int w4 = 0; w4 < 6; w4++) px[i][w4] = 0;
#pragma unroll
            for (int x = 0; x < 8; x++) {
                // src/decoder.rs:271-273 with cb = cr = 128: the products are +-0 and every sum is the luma value
                const float yv = y[x] + 128.0f;
                const float r = yv, g = yv, b2 = yv;
                px[i][(3 * x + 0) >> 2] = zj_pack_u8(r, (3 * x + 0) & 3, px[i][(3 * x + 0) >> 2]);
                px[i][(3 * x + 1) >> 2] = zj_pack_u8(g, (3 * x + 1) & 3, px[i][(3 * x + 1) >> 2]);
                px[i][(3 * x + 2) >> 2] = zj_pack_u8(b2, (3 * x + 2) & 3, px[i][(3 * x + 2) >> 2]);
            }
        }
        zj_wave_lds_sync();
        char *stage = reinterpret_cast<char *>(blk);
#pragma unroll
        for (uint32_t i = 0; i < 3u; i++) {
            const uint32_t q = lane + i * 64u;
            char *d = stage + (q & 7u) * ROWP + (q >> 3) * 24u;
#pragma unroll
            for (int w4 = 0; w4 < 6; w4 += 2) *reinterpret_cast<uint2 *>(d + 4 * w4) = make_uint2(px[i][w4], px[i][w4 + 1]);
        }
        zj_wave_lds_sync();
        const uint32_t xb = x0 * 24u;
        const uint32_t row_valid = min((uint32_t)ROWB, (im.w - min(im.w, x0 * 8u)) * 3u);
        const uint32_t rows_valid = min(8u, im.h - min(im.h, my * 8u));
        for (uint32_t q = lane; q < (uint32_t)(PIECES * 8); q += 64u) {
            const uint32_t r = q / PIECES, o = (q % PIECES) * 16u;
            if (r >= rows_valid || o >= row_valid) continue;
            uint8_t *dst = rgb + (size_t)((my - im.row0) * 8u + r) * stride + xb + o;
            const uint4 pv = *reinterpret_cast<const uint4 *>(stage + r * ROWP + o);
            if (vec_ok && o + 16u <= row_valid) {
                *reinterpret_cast<uint4 *>(dst) = pv;
            } else {
                const uint32_t s4[4] = {pv.x, pv.y, pv.z, pv.w};
                const uint32_t nbytes = min(16u, row_valid - o);
                for (uint32_t k = 0; k < nbytes; k++) dst[k] = (uint8_t)(s4[k >> 2] >> (8u * (k & 3u)));
            }
        }
        zj_wave_lds_sync();
    }
}

// ---------------------------------------------------------------------------------- restart markers (planning)
// Where the restart intervals of an image begin: every position p of the scan with byte p = FF and byte
// p+1 = D0..D7 (src/bitstream.rs:41-47 stops at such a pair; the reference has no restart support, the
// intervals are the defined extension).  An ordered stream compaction in three launches: markers per tile
// of 16 KiB, exclusive scan of the tile counts per image, then the same search again writing the offsets.
// A thread owns 64 bytes and builds a 64-bit marker mask with word-parallel byte tests.
__device__ __forceinline__ uint64_t zj_marker_mask(const ZjMarkParams &M, const ZjMarkImg &d, uint32_t p0)
{
    if (p0 + 1u >= d.scan_len) return 0;
    const uint8_t *src = M.raw + d.raw_off + p0;
    uint32_t w[17];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint4 v = reinterpret_cast<const uint4 *>(src)[q];
        w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
    }
    w[16] = reinterpret_cast<const uint32_t *>(src)[16]; // the arena has slack behind the last scan
    uint32_t ff[16], dx[17];
#pragma unroll
    for (int j = 0; j < 17; j++) {
        const uint32_t y = ~w[j], x = (w[j] ^ 0xD0D0D0D0u) & 0xF8F8F8F8u;
        if (j < 16) ff[j] = ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu); // bytes equal to FF
        dx[j] = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);             // bytes D0..D7
    }
    uint64_t mask = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const uint32_t m = ff[j] & __builtin_amdgcn_alignbyte(dx[j + 1], dx[j], 1); // FF followed by D0..D7
        const uint32_t nib = ((((m >> 7) & 0x01010101u) * 0x01020408u) >> 24) & 15u;
        mask |= (uint64_t)nib << (4 * j);
    }
    // only pairs that lie inside the scan: p + 1 < scan_len
    const uint32_t nvalid = d.scan_len - 1u - p0;
    if (nvalid < 64u) mask &= (1ull << nvalid) - 1ull;
    return mask;
}

template <bool WRITE>
__global__ __launch_bounds__(256) void zj_km_markers(ZjMarkParams M)
{
    __shared__ uint32_t wsum[8];
    const ZjMarkImg d = M.imgs[M.tile_img[blockIdx.x]];
    const uint32_t p0 = (blockIdx.x - d.first_tile) * ZJ_MARK_TILE + threadIdx.x * 64u;
    uint64_t mask = zj_marker_mask(M, d, p0);
    const uint32_t n = (uint32_t)__popcll(mask);
    uint32_t total;
    const uint32_t inc = block_incl_scan<4>(n, wsum, &total);
    if (!WRITE) {
        if (threadIdx.x == 0) M.tile_count[blockIdx.x] = total;
    } else {
        uint32_t k = M.tile_count[blockIdx.x] + inc - n;
        while (mask && k < d.cap) {
            const uint32_t bit = (uint32_t)__builtin_ctzll(mask);
            M.offs[d.out_first + k] = p0 + bit + 2u;
            mask &= mask - 1ull;
            k++;
        }
    }
}

__global__ __launch_bounds__(256) void zj_km_scan(ZjMarkParams M)
{
    __shared__ uint32_t wsum[8];
    const ZjMarkImg d = M.imgs[blockIdx.x];
    uint32_t run = 0;
    for (uint32_t t0 = 0; t0 < d.ntiles; t0 += 256u) { // uniform trip count
        const uint32_t t = t0 + threadIdx.x;
        const uint32_t v = t < d.ntiles ? M.tile_count[d.first_tile + t] : 0u;
        uint32_t total;
        const uint32_t inc = block_incl_scan<4>(v, wsum, &total);
        if (t < d.ntiles) M.tile_count[d.first_tile + t] = run + inc - v;
        run += total;
        __syncthreads(); // wsum is reused by the next round
    }
    if (threadIdx.x == 0) M.img_count[blockIdx.x] = run;
}

// ---------------------------------------------------------------------------------- export (tests)
// dqcoef[b][nat] = qcoef[b][zz] * q[zz]: the reference's `blocks` entry (src/decoder.rs:237-240).
// Coefficient 0 of every block of the arena from the DC array (the write pass leaves the difference there): for callers
// that read the arena itself.
__global__ __launch_bounds__(256) void zj_kx_patch_dc(ZjParams P, uint32_t nblocks)
{
    const uint32_t g = blockIdx.x * 256u + threadIdx.x; // index in the DC array
    if (g >= nblocks) return;
    const ZjImgDev &im = P.imgs[P.blk256_img[g >> 8]];
    const uint32_t q = g + im.mcu_shift * im.bpm; // index in the arena: the same, except in a part of a split stream
    if (q - im.first_block < im.nblocks) P.qcoef[(size_t)q * 64u] = P.dcabs[g];
}

__global__ __launch_bounds__(256) void zj_kx_dequant(ZjParams P, uint32_t ii, int16_t *dst)
{
    const ZjImgDev im = P.imgs[ii]; // by value: a reference would be read again after every store
    const size_t idx = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (idx >= (size_t)im.nblocks * 64u) return;
    const uint32_t b = (uint32_t)(idx >> 6), zz = (uint32_t)(idx & 63u);
    const uint32_t kb = b % im.bpm;
    const uint32_t comp = (im.compsel >> (2u * kb)) & 3u;
    const int32_t c = P.qcoef[(size_t)im.first_block * 64u + idx];
    const int32_t qv = P.qts[im.qt_off[comp] + zz];
    const uint32_t nat = ((uint32_t)(zj_zz2nat_bytes[zz >> 3] >> (8u * (zz & 7u))) & 0xFFu) >> 2;
    dst[(size_t)b * 64u + nat] = (int16_t)(c * qv);
}

// ---------------------------------------------------------------------------------- launchers
static void zj_launch_fix(const ZjParams &P, const ZjLaunch &L, hipStream_t st)
{
    if (L.nchunks / 4 <= 1024u) hipLaunchKernelGGL(zj_k1b_fix<true>, dim3(L.nchunks / 4), dim3(64), 0, st, P);
    else hipLaunchKernelGGL(zj_k1b_fix<false>, dim3(L.nchunks / 4), dim3(64), 0, st, P);
}

static void zj_launch_scan(const ZjParams &P, const ZjLaunch &L, hipStream_t st)
{
    if (L.nseg >= 256u) hipLaunchKernelGGL(zj_k1s_scan, dim3(L.nseg), dim3(256), 0, st, P);
    else hipLaunchKernelGGL(zj_k1s_scan_few, dim3(L.nseg), dim3(1024), 0, st, P);
}

static void zj_launch_dcscan(const ZjParams &P, const ZjLaunch &L, hipStream_t st)
{
    if (L.dc_by_segment) {
        hipLaunchKernelGGL(zj_k1d_dcscan_seg, dim3(L.nseg), dim3(1024), 0, st, P);
    } else {
        if (L.ndmulti) { // segments of more than one tile: their tile sums first
            hipLaunchKernelGGL(zj_k1d_tilesum, dim3(L.ndtiles), dim3(256), 0, st, P);
            hipLaunchKernelGGL(zj_k1d_tilescan, dim3(L.nseg), dim3(256), 0, st, P);
        }
        if (L.ndtiles) hipLaunchKernelGGL(zj_k1d_dcscan, dim3(L.ndtiles), dim3(256), 0, st, P);
    }
    if (L.ndsmall) hipLaunchKernelGGL(zj_k1d_dcscan_small, dim3((L.ndsmall + 3u) / 4u), dim3(256), 0, st, P, L.ndtiles, L.ndsmall);
}

extern "C" void zj_launch_stage1(const ZjParams &P, const ZjLaunch &L, hipStream_t st, hipEvent_t *ev)
{
    int e = 0;
    if (ev) hipEventRecord(ev[e++], st);
    hipMemsetAsync(P.flags, 0, L.clear_bytes, st); // one fill: flags, status words, look-back words and the tickets behind them
    if (L.ntiles) hipLaunchKernelGGL((zj_k0_unstuff<128, 31, false>), dim3(L.ntiles), dim3(128), 0, st, P, 0u);
    if (L.nsmall) hipLaunchKernelGGL((zj_k0_unstuff<64, 15, true>), dim3(L.nsmall), dim3(64), 0, st, P, L.ntiles);
    if (ev) hipEventRecord(ev[e++], st);
    hipLaunchKernelGGL(zj_k1a_sync, dim3(L.nchunks / 4), dim3(256), 0, st, P);
    if (L.npacked) hipLaunchKernelGGL(zj_k1a_sync_packed, dim3(L.nchunks / 4), dim3(256), 0, st, P);
    if (ev) hipEventRecord(ev[e++], st);
    for (int i = 0; i < L.fix_iters; i++) {
        if (i) hipMemsetAsync(P.flags, 0, 4, st); // only the last pass says whether the records have converged
        zj_launch_fix(P, L, st);
    }
    if (ev) hipEventRecord(ev[e++], st);
    zj_launch_scan(P, L, st);
    if (ev) hipEventRecord(ev[e++], st);
    if (L.nptiles) hipLaunchKernelGGL(zj_k1p_positions, dim3(L.nptiles), dim3(256), 0, st, P);
    if (ev) hipEventRecord(ev[e++], st);
    if (P.dc_in_write) {
        // the write pass decodes the DC symbols itself and leaves the differences for the scan behind it
        hipLaunchKernelGGL(zj_k1w_write, dim3(L.nk1w), dim3(256), 0, st, P);
        if (ev) hipEventRecord(ev[e++], st);
        zj_launch_dcscan(P, L, st);
        if (ev) hipEventRecord(ev[e++], st);
    } else {
        // a window of MCU rows: the predictors at its first MCU need the differences of all blocks before it
        hipLaunchKernelGGL(zj_k1c_dcdiff, dim3((L.nblk256 + ZJ_K1C_NB - 1) / ZJ_K1C_NB), dim3(256), 0, st, P, L.nblk256);
        zj_launch_dcscan(P, L, st);
        if (ev) hipEventRecord(ev[e++], st);
        hipLaunchKernelGGL(zj_k1w_write, dim3(L.nk1w), dim3(256), 0, st, P);
        if (ev) hipEventRecord(ev[e++], st);
    }
}

// The halves of stage 1 for a part of a split stream (zj_batch_range_*): up to the block counts; from the block counts
// to the DC values; the correction of the DC values and the coefficients.
extern "C" void zj_launch_range_sync(const ZjParams &P, const ZjLaunch &L, hipStream_t st)
{
    hipMemsetAsync(P.flags, 0, L.clear_bytes, st);
    if (L.ntiles) hipLaunchKernelGGL((zj_k0_unstuff<128, 31, false>), dim3(L.ntiles), dim3(128), 0, st, P, 0u);
    if (L.nsmall) hipLaunchKernelGGL((zj_k0_unstuff<64, 15, true>), dim3(L.nsmall), dim3(64), 0, st, P, L.ntiles);
    hipLaunchKernelGGL(zj_k1a_sync, dim3(L.nchunks / 4), dim3(256), 0, st, P);
    if (L.npacked) hipLaunchKernelGGL(zj_k1a_sync_packed, dim3(L.nchunks / 4), dim3(256), 0, st, P);
    for (int i = 0; i < L.fix_iters; i++) {
        if (i) hipMemsetAsync(P.flags, 0, 4, st); // only the last pass says whether the records have converged
        zj_launch_fix(P, L, st);
    }
    zj_launch_scan(P, L, st);
}

extern "C" void zj_launch_range_dc(const ZjParams &P, const ZjLaunch &L, hipStream_t st)
{
    zj_launch_scan(P, L, st); // again, from the segment's bias
    if (L.nptiles) hipLaunchKernelGGL(zj_k1p_positions, dim3(L.nptiles), dim3(256), 0, st, P);
    hipLaunchKernelGGL(zj_k1c_dcdiff, dim3((L.nblk256 + ZJ_K1C_NB - 1) / ZJ_K1C_NB), dim3(256), 0, st, P, L.nblk256);
    zj_launch_dcscan(P, L, st);
}

extern "C" void zj_launch_range_write(const ZjParams &P, const ZjLaunch &L, hipStream_t st, uint32_t g0, uint32_t g1, const uint32_t *off)
{
    if (g1 > g0) hipLaunchKernelGGL(zj_k1d_add_offset, dim3((g1 - g0 + 255u) / 256u), dim3(256), 0, st, P, 0u, g0, g1, off[0], off[1], off[2]);
    if (L.nk1w) hipLaunchKernelGGL(zj_k1w_write, dim3(L.nk1w), dim3(256), 0, st, P);
}

extern "C" void zj_launch_markers(const ZjMarkParams &M, hipStream_t st)
{
    if (!M.ntiles) return;
    hipLaunchKernelGGL(zj_km_markers<false>, dim3(M.ntiles), dim3(256), 0, st, M);
    hipLaunchKernelGGL(zj_km_scan, dim3(M.nimgs), dim3(256), 0, st, M);
    hipLaunchKernelGGL(zj_km_markers<true>, dim3(M.ntiles), dim3(256), 0, st, M);
}

extern "C" void zj_launch_fix_only(const ZjParams &P, const ZjLaunch &L, hipStream_t st)
{
    zj_launch_fix(P, L, st);
}

extern "C" void zj_launch_stage1_tail(const ZjParams &P, const ZjLaunch &L, hipStream_t st)
{
    zj_launch_scan(P, L, st);
    if (L.nptiles) hipLaunchKernelGGL(zj_k1p_positions, dim3(L.nptiles), dim3(256), 0, st, P);
    if (P.dc_in_write) {
        hipLaunchKernelGGL(zj_k1w_write, dim3(L.nk1w), dim3(256), 0, st, P);
        zj_launch_dcscan(P, L, st);
    } else {
        hipLaunchKernelGGL(zj_k1c_dcdiff, dim3((L.nblk256 + ZJ_K1C_NB - 1) / ZJ_K1C_NB), dim3(256), 0, st, P, L.nblk256);
        zj_launch_dcscan(P, L, st);
        hipLaunchKernelGGL(zj_k1w_write, dim3(L.nk1w), dim3(256), 0, st, P);
    }
}

extern "C" void zj_launch_stage2(const ZjParams &P, const ZjLaunch &L, hipStream_t st, hipEvent_t *ev)
{
    // classes: 0 = 4:2:0 (2x2), 1 = 4:4:4 (1x1), 2 = 4:2:2 (2x1), 3 = 4:4:0 (1x2)
    if (L.k2_count[0]) hipLaunchKernelGGL((zj_k2_idct_rgb<2, 2>), dim3(L.k2_count[0]), dim3(256), 0, st, P, L.k2_first[0]);
    if (L.k2_count[1]) hipLaunchKernelGGL((zj_k2_idct_rgb<1, 1>), dim3(L.k2_count[1]), dim3(256), 0, st, P, L.k2_first[1]);
    if (L.k2_count[2]) hipLaunchKernelGGL((zj_k2_idct_rgb<2, 1>), dim3(L.k2_count[2]), dim3(256), 0, st, P, L.k2_first[2]);
    if (L.k2_count[3]) hipLaunchKernelGGL((zj_k2_idct_rgb<1, 2>), dim3(L.k2_count[3]), dim3(256), 0, st, P, L.k2_first[3]);
    if (L.k2_count[4]) hipLaunchKernelGGL(zj_k2_idct_gray, dim3(L.k2_count[4]), dim3(256), 0, st, P, L.k2_first[4]);
    if (ev) hipEventRecord(ev[0], st);
}

extern "C" void zj_launch_patch_dc(const ZjParams &P, uint32_t nblocks, hipStream_t st)
{
    if (nblocks) hipLaunchKernelGGL(zj_kx_patch_dc, dim3((nblocks + 255u) / 256u), dim3(256), 0, st, P, nblocks);
}

extern "C" void zj_launch_dequant(const ZjParams &P, uint32_t ii, uint32_t nblocks, int16_t *dst, hipStream_t st)
{
    uint32_t n = (uint32_t)(((size_t)nblocks * 64u + 255u) / 256u);
    if (n) hipLaunchKernelGGL(zj_kx_dequant, dim3(n), dim3(256), 0, st, P, ii, dst);
}
