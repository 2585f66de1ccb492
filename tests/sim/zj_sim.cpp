// zj_sim.cpp -- lane-by-lane CPU walk of the stage-1 algorithm (TEST INFRASTRUCTURE ONLY).
// It runs the very functions the kernels run (zj_core.h: zj_lookup, zj_decode_span,
// zj_decode_block; zj_host.cpp: zj_build_table, zj_restart_offsets) in the order the kernels run
// them -- per-chunk rounds, boundary fix-up, block-count scan, index pass with verification, DC
// scan, per-block write -- so the synchronisation logic can be checked against the oracle on a
// machine without a GPU.  It is not part of the shipped library and nothing ships that calls it.
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/zj.h"
#include "../../jpeg-decoder_amd/csrc/zj_core.h"
#include "../../jpeg-decoder_amd/csrc/zj_hostutil.h"

struct SimStats {
    uint32_t rounds_max, rounds_sum, nchunks, fix_steps_max, fix_steps_sum, need_more, verify_fail;
};

extern "C" int zjsim_decode(const zj_image *im, uint32_t subbits, int16_t *qcoef, uint32_t *status_out, SimStats *stats)
{
    if (!zj_image_supported(im)) return ZJ_E_UNSUPPORTED;
    memset(stats, 0, sizeof *stats);
    const uint32_t hs = im->hsamp[0], vs = im->vsamp[0], bpm = im->ncomp == 1 ? 1u : hs * vs + 2;
    const uint32_t nmcu = zj_image_mcus(im);
    std::vector<uint16_t> lut(ZJ_SET_ENTRIES);
    ZjCanon canon[4];
    const zj_huff *t4[4] = {im->ht[0][1], im->ht[1][1], im->ht[0][0], im->ht[1][0]};
    for (int t = 0; t < 4; t++) {
        if (t4[t]) { if (!zj_build_table(t4[t], &lut[t * ZJ_TBL_ENTRIES], &canon[t])) return ZJ_E_FORMAT; }
        else memset(&canon[t], 0, sizeof(ZjCanon));
    }
    std::vector<uint16_t> wlut(ZJ_SET_ENTRIES);
    for (int t = 0; t < 4; t++) zj_build_walk_table(&lut[t * ZJ_TBL_ENTRIES], t < 2, &wlut[t * ZJ_TBL_ENTRIES]);
    ZjSel sel = {bpm, 0, 0};
    uint32_t compsel = 0;
    for (uint32_t k = 0; k < bpm; k++) {
        uint32_t c = k < hs * vs ? 0u : (k == hs * vs ? 1u : 2u);
        compsel |= c << (2 * k);
        sel.dcsel |= (uint32_t)im->td[c] << k;
        sel.acsel |= (uint32_t)im->ta[c] << k;
    }
    const uint32_t ri = im->restart_interval;
    const uint32_t nseg = ri ? (nmcu + ri - 1) / ri : 1;
    std::vector<size_t> offs(nseg + 1, 0);
    size_t nint = 1;
    if (ri) nint = std::min(zj_restart_offsets(im, offs.data(), offs.size()), offs.size());
    uint32_t status = 0;
    for (uint32_t k = 0; k < nseg; k++) {
        size_t beg = im->scan_len, end = im->scan_len;
        if (k < nint) { beg = offs[k]; end = (k + 1 < nint) ? offs[k + 1] - 2 : im->scan_len; }
        const uint32_t m0 = ri ? k * ri : 0, mc = ri ? std::min(ri, nmcu - m0) : nmcu;
        const uint32_t nblocks = mc * bpm;
        int16_t *qc = qcoef + (size_t)m0 * bpm * 64;
        // k0: unstuff
        std::vector<uint8_t> ust;
        for (size_t p = beg; p < end; p++) {
            if (im->scan[p] == 0 && p > beg && im->scan[p - 1] == 0xff) continue;
            ust.push_back(im->scan[p]);
        }
        const uint32_t pay = (uint32_t)ust.size() * 8;
        ust.resize(ust.size() + 1024, 0);
        while (ust.size() % 4) ust.push_back(0);
        const uint32_t *stream = (const uint32_t *)ust.data();
        uint32_t SB, nsub, LN; // LN: lanes the segment's subsequences share (a whole wavefront, or a group of a packed chunk)
        zj_seg_layout((uint64_t)(end - beg) * 8u, subbits, subbits, &SB, &nsub, &LN);
        const uint32_t nchunk = (nsub + LN - 1) / LN;
        const uint32_t NC = SB / ZJ_WIN_BITS;
        const uint32_t selw = zj_selword(sel);
        std::vector<uint32_t> rec((size_t)nchunk * LN * NC + 1, 0), wbase((size_t)nchunk * LN * NC + 1, 0);
        std::vector<uint32_t> bitmap(ust.size() / 4 + 64, 0);
        auto active = [&](uint32_t q) { return q < nsub && (uint64_t)q * SB < pay; };
        auto send = [&](uint32_t q) { uint64_t e = (uint64_t)(q + 1) * SB; return e < pay ? (uint32_t)e : pay; };
        // one window of a walk: symbols that start before the end of window w (or `e`), on a reader that
        // is re-positioned every time (the kernels carry the reader over; the bits are the same)
        auto window = [&](ZjState &st, uint32_t w, uint32_t e) {
            uint32_t wend = (w + 1) * ZJ_WIN_BITS;
            return zj_walk_window_direct(stream, bitmap.data(), wlut.data(), canon, selw, bpm, st, w, wend < e ? wend : e);
        };
        // k1a: per chunk, rounds in lock step, one record and one set of bitmap words per window
        for (uint32_t ch = 0; ch < nchunk; ch++) {
            ZjState st[64];
            bool done[64];
            for (uint32_t l = 0; l < LN; l++) {
                uint32_t jl = ch * LN + l;
                done[l] = !active(jl);
                st[l] = {jl * SB, 0, 0};
                if (done[l]) continue;
                for (uint32_t c = 0; c < NC; c++) {
                    if (st[l].p >= send(jl)) break;
                    rec[zj_rec_index(jl, c, NC, LN)] = window(st[l], jl * NC + c, send(jl));
                }
            }
            uint32_t rounds = 0;
            for (uint32_t k2 = 1; k2 < LN; k2++) {
                bool any = false;
                bool act[64];
                for (uint32_t l = 0; l < LN; l++) {
                    act[l] = !done[l] && l + k2 < LN && active(ch * LN + l + k2);
                    if (!act[l]) done[l] = true;
                    any = any || act[l];
                }
                if (!any) break;
                rounds++;
                // the walks of one round touch distinct subsequences, so lane order is free
                for (uint32_t l = 0; l < LN; l++) {
                    if (!act[l]) continue;
                    uint32_t q = ch * LN + l + k2;
                    for (uint32_t c = 0; c < NC; c++) {
                        if (st[l].p >= send(q)) break;
                        uint32_t nw = window(st[l], q * NC + c, send(q));
                        uint32_t &r = rec[zj_rec_index(q, c, NC, LN)];
                        uint32_t old = r;
                        r = nw;
                        if (((old ^ nw) & ZJ_INFO_STATE_MASK) == 0) { done[l] = true; break; }
                    }
                }
            }
            stats->rounds_max = std::max(stats->rounds_max, rounds);
            stats->rounds_sum += rounds;
            stats->nchunks++;
        }
        // k1b: one pass of boundary fix-up (lane order = any; emulate ascending)
        for (uint32_t ch = 1; ch < nchunk; ch++) {
            uint32_t q0 = ch * LN;
            if (!active(q0)) continue;
            ZjState st = zj_unpack_state(rec[zj_rec_index(q0 - 1, NC - 1, NC, LN)], q0 * SB);
            uint32_t steps = 0;
            bool synced = false;
            for (uint32_t t = 0; t < LN && !synced; t++) {
                uint32_t q = q0 + t;
                if (!active(q)) { synced = true; break; }
                for (uint32_t c = 0; c < NC; c++) {
                    if (st.p >= send(q)) break;
                    steps++;
                    uint32_t nw = window(st, q * NC + c, send(q));
                    uint32_t &r = rec[zj_rec_index(q, c, NC, LN)];
                    uint32_t old = r;
                    r = nw;
                    if (((old ^ nw) & ZJ_INFO_STATE_MASK) == 0) { synced = true; break; }
                }
            }
            if (!synced) stats->need_more = 1;
            stats->fix_steps_max = std::max(stats->fix_steps_max, steps);
            stats->fix_steps_sum += steps;
        }
        // k1s: window bases in stream order
        const uint32_t nwin = std::min(nsub * NC, (pay + ZJ_WIN_BITS - 1) / ZJ_WIN_BITS);
        uint32_t carry = 0;
        for (uint32_t wi = 0; wi < nwin; wi++) {
            wbase[wi] = carry;
            carry += rec[zj_rec_index(wi / NC, wi % NC, NC, LN)] >> 16;
        }
        const uint32_t found = std::min(carry, nblocks);
        // k1p: bitmap -> block positions
        std::vector<uint32_t> blockpos(nblocks + 1, 0);
        std::vector<int16_t> dcabs(nblocks + 1, 0);
        for (uint32_t wi = 0; wi < nwin; wi++) {
            uint32_t g = wbase[wi];
            for (uint32_t i = 0; i < ZJ_WIN_WORDS; i++)
                for (uint32_t bit = 0; bit < 32; bit++)
                    if ((bitmap[(size_t)wi * ZJ_WIN_WORDS + i] >> bit) & 1u) {
                        if (g < nblocks) blockpos[g] = wi * ZJ_WIN_BITS + i * 32 + bit;
                        g++;
                    }
        }
        // k1d
        uint32_t run[3] = {0, 0, 0};
        for (uint32_t g = 0; g < nblocks; g++) {
            uint32_t c = (compsel >> (2 * (g % bpm))) & 3;
            if (g < found) {
                uint32_t t = (sel.dcsel >> (g % bpm)) & 1;
                uint32_t nbits;
                run[c] += (uint32_t)zj_dc_diff(stream, &lut[t * ZJ_TBL_ENTRIES], &canon[t], blockpos[g], &nbits);
            }
            dcabs[g] = (int16_t)run[c];
        }
        // k1w
        for (uint32_t g = 0; g < nblocks; g++) {
            int16_t rowbuf[66];
            memset(rowbuf, 0, sizeof rowbuf);
            int16_t *row = rowbuf;
            if (g >= found) { status |= ZJ_ST_TRUNCATED; memset(qc + (size_t)g * 64, 0, 128); continue; }
            uint32_t kb = g % bpm, pend;
            uint32_t s = zj_decode_block(stream, lut.data(), canon, (sel.dcsel >> kb) & 1, (sel.acsel >> kb) & 1, blockpos[g],
                                         dcabs[g], row, &pend);
            if (pend > pay) s |= ZJ_ST_TRUNCATED;
            if (s == 0 && g + 1 < found && blockpos[g + 1] != pend) stats->verify_fail = 1;
            status |= s;
            memcpy(qc + (size_t)g * 64, rowbuf, 128);
        }
    }
    *status_out = status;
    return ZJ_OK;
}

// EXTEND of AC value bits, both forms, for the exhaustive equality test.
extern "C" int zjsim_value_ac(uint32_t x, uint32_t nbits, int32_t *fast, int32_t *ref)
{
    *fast = zj_value_ac(x, nbits);
    *ref = zj_extend(nbits, nbits ? x >> (32 - nbits) : 0);
    return *fast == *ref;
}

// Table lookup against the canonical search on random windows, for a table built with `l2_cap`
// second-level entries (0 pushes every long code to the canonical search).  Returns mismatches.
extern "C" int zjsim_lookup_check(const zj_huff *h, uint32_t l2_cap, uint32_t ntests, uint64_t seed)
{
    std::vector<uint16_t> tbl(ZJ_TBL_ENTRIES), wtbl(ZJ_TBL_ENTRIES);
    ZjCanon canon;
    if (!zj_build_table(h, tbl.data(), &canon, l2_cap)) return -1;
    int bad = 0;
    for (int isdc = 0; isdc < 2; isdc++) {
        zj_build_walk_table(tbl.data(), isdc != 0, wtbl.data());
        uint64_t x = seed * 2654435761u + 12345u;
        for (uint32_t i = 0; i < ntests; i++) {
            x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
            const uint32_t top = (uint32_t)((x * 2685821657736338717ull) >> 32);
            const uint32_t want = zj_canon_lookup(&canon, top);
            if (zj_lookup(tbl.data(), &canon, top) != want) bad++;
            if (zj_lookup_walk(wtbl.data(), &canon, top, isdc != 0) != zj_walk_entry(want, isdc != 0)) bad++;
        }
    }
    return bad;
}

// The planner's cut of a segment into subsequences (zj_core.h), for the layout test.
extern "C" void zjsim_seg_layout(uint64_t raw_bits, uint32_t target, uint32_t small_target, uint32_t *subbits, uint32_t *nsub, uint32_t *lanes)
{
    zj_seg_layout(raw_bits, target, small_target, subbits, nsub, lanes);
}

// The walk with pair entries against the symbol-by-symbol walk on a random bit stream: from random states,
// over stretches that end at random bits, both must stop in the same state and mark the same block starts.
// Returns mismatches, -1 if the table is no prefix code.
extern "C" int zjsim_pair_walk_check(const zj_huff *dc, const zj_huff *ac, uint32_t l2_cap, uint32_t ntests, uint64_t seed, uint32_t *pairs_taken)
{
    std::vector<uint16_t> tbl(ZJ_SET_ENTRIES), wtbl(ZJ_SET_ENTRIES);
    std::vector<uint32_t> ptab(ZJ_PSET_ENTRIES);
    ZjCanon canon[4];
    const zj_huff *t4[4] = {dc, dc, ac, ac};
    for (int t = 0; t < 4; t++) {
        if (!zj_build_table(t4[t], &tbl[t * ZJ_TBL_ENTRIES], &canon[t], l2_cap)) return -1;
        zj_build_walk_table(&tbl[t * ZJ_TBL_ENTRIES], t < 2, &wtbl[t * ZJ_TBL_ENTRIES]);
    }
    for (int t = 0; t < 4; t++) {
        // the DC tables carry the first AC symbol of their blocks as second symbol (selectors below: DC id k with AC id k)
        zj_build_pair_table(&canon[t], t < 2, &ptab[t * ZJ_PT_ENTRIES], t < 2 ? &canon[2 + t] : nullptr);
        if (l2_cap == 0)
            for (uint32_t e = 0; e < (uint32_t)ZJ_PL1_SIZE; e++)
                if (ptab[t * ZJ_PT_ENTRIES + e] & 0x80000000u) ptab[t * ZJ_PT_ENTRIES + e] = 0x80000000u;
    }
    const uint32_t nwords = 4096;
    std::vector<uint32_t> stream(nwords + 64), bm1(nwords + 64), bm2(nwords + 64);
    uint64_t x = seed * 2654435761u + 99991u;
    auto rnd = [&]() { x ^= x >> 12; x ^= x << 25; x ^= x >> 27; return (uint32_t)((x * 2685821657736338717ull) >> 32); };
    ZjSel sel = {6, 0x30, 0x30};
    const uint32_t selw = zj_selword(sel);
    int bad = 0;
    uint32_t steps1 = 0, steps2 = 0;
    for (uint32_t it = 0; it < ntests; it++) {
        for (auto &w : stream) w = (it & 1u) ? rnd() : (rnd() & rnd()); // dense, and sparse in ones (short codes)
        std::fill(bm1.begin(), bm1.end(), 0u);
        std::fill(bm2.begin(), bm2.end(), 0u);
        ZjState a = {rnd() % 1000u, rnd() % 6u, rnd() % 64u}, b = a;
        const uint32_t stop = a.p + 1u + rnd() % (nwords * 32u - 2000u);
        // in stretches, as the kernels walk window by window
        while (a.p < stop) {
            const uint32_t part = std::min(stop, (a.p / 512u + 1u) * 512u);
            {
                ZjBits bits; bits.init(stream.data(), a.p);
                ZjEmitWords em = {bm1.data()};
                zj_decode_run<ZjBits, ZjEmitWords>(bits, wtbl.data(), canon, selw, sel.bpm, a, part, em);
            }
            {
                ZjBits bits; bits.init(stream.data(), b.p);
                ZjEmitWords em = {bm2.data()};
                zj_decode_run_pairs<ZjBits, ZjEmitWords>(bits, ptab.data(), canon, selw, sel.bpm, b, part, em);
            }
            if (a.p != b.p || a.b != b.b || a.z != b.z) { bad++; break; }
        }
        if (bm1 != bm2) bad++;
        (void)steps1; (void)steps2;
    }
    if (pairs_taken) {
        uint32_t n = 0;
        for (uint32_t t = 0; t < 4; t++)
            for (uint32_t e = 0; e < (uint32_t)ZJ_PL1_SIZE; e++) {
                const uint32_t v = ptab[t * ZJ_PT_ENTRIES + e];
                if (!(v & 0x80000000u) && ((v >> 8) & 255u) != (v & 255u)) n++;
            }
        *pairs_taken = n;
    }
    return bad;
}
