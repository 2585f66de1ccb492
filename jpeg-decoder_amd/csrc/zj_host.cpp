// zj_host.cpp -- host side of the decode path: the container parse the reference keeps on the CPU
// (marker walk, DQT / DHT / SOF0 / SOS, src/decoder.rs:347-642), geometry helpers, restart-interval
// bookkeeping for multi-GPU bands, lookup-table construction, PPM output.  No GPU code here.
#include <stdio.h>
#include <string.h>

#include <vector>

#include "../../include/zj.h"
#include "zj_core.h"
#include "zj_hostutil.h"

namespace {
struct Rd {
    const uint8_t *p;
    size_t len, pos;
    int u8() { return pos < len ? p[pos++] : -1; }
    int u16()
    {
        if (pos + 2 > len) { pos = len; return -1; }
        int v = (p[pos] << 8) | p[pos + 1];
        pos += 2;
        return v;
    }
};
} // namespace

extern "C" int zj_parse(const uint8_t *file, size_t len, zj_parsed *out)
{
    if (!file || !out) return ZJ_E_INVALID;
    memset(out, 0, sizeof *out);
    zj_image &im = out->image;
    Rd r{file, len, 0};
    bool seen_sof = false;
    bool qt_def[4] = {false, false, false, false};
    bool ht_def[2][2] = {{false, false}, {false, false}};
    for (;;) {
        int marker = r.u16();
        if (marker < 0 || (marker >> 8) != 0xff) return ZJ_E_FORMAT; // src/decoder.rs:62-64
        const int m = marker & 0xff;
        if (m == 0xd8) continue;           // SOI, :363
        if (m == 0xd9) return ZJ_E_FORMAT; // EOI with no scan
        if (m == 0xff) { r.pos -= 1; continue; } // fill byte before a marker
        const int seglen = r.u16();
        if (seglen < 2 || r.pos + (size_t)(seglen - 2) > len) return ZJ_E_FORMAT;
        const size_t seg_end = r.pos + (size_t)(seglen - 2);
        switch (m) {
        case 0xdb: // DQT: one segment may define several tables (src/decoder.rs:428-466)
            while (r.pos < seg_end) {
                const int info = r.u8();
                const int dst = info & 0xf;
                if (dst > 3) return ZJ_E_FORMAT;
                if (info & 0xf0) return ZJ_E_UNSUPPORTED; // 16-bit tables (assert at :450)
                if (r.pos + 64 > seg_end) return ZJ_E_FORMAT;
                memcpy(out->qt_store[dst], file + r.pos, 64); // file order = zig-zag, :453
                qt_def[dst] = true;
                r.pos += 64;
            }
            break;
        case 0xc4: // DHT (src/decoder.rs:467-562)
            while (r.pos < seg_end) {
                const int info = r.u8();
                const int id = info & 0xf;
                if (info & 0xe0) return ZJ_E_FORMAT; // :486
                if (id > 1) return ZJ_E_UNSUPPORTED;  // baseline allows two tables per class, :478
                const int is_dc = ((info >> 4) & 1) == 0;
                if (r.pos + 16 > seg_end) return ZJ_E_FORMAT;
                zj_huff &h = out->ht_store[id][is_dc]; // [id][0 = AC, 1 = DC], :553
                memset(&h, 0, sizeof h);
                memcpy(h.counts, file + r.pos, 16);
                r.pos += 16;
                int n = 0;
                for (int i = 0; i < 16; i++) n += h.counts[i];
                if (n <= 0 || n > 256 || r.pos + (size_t)n > seg_end) return ZJ_E_FORMAT; // :509
                memcpy(h.symbols, file + r.pos, (size_t)n);
                r.pos += (size_t)n;
                ht_def[id][is_dc] = true;
            }
            break;
        case 0xc0: { // SOF0 (src/decoder.rs:564-630)
            if (seglen < 8 + 3) return ZJ_E_FORMAT;
            const int prec = r.u8();
            const int h = r.u16(), w = r.u16();
            const int nc = r.u8();
            if (prec != 8) return ZJ_E_UNSUPPORTED;
            // three components, or one (admitted at :574-575; the reference's MCU loop, :170-173, cannot
            // decode it: here a one-component frame is single blocks in raster order, T.81 A.2.2)
            if (nc != 3 && nc != 1) return ZJ_E_UNSUPPORTED;
            if (seglen < 8 + 3 * (size_t)nc) return ZJ_E_FORMAT;
            if (w <= 0 || h <= 0) return ZJ_E_FORMAT;
            im.width = (uint16_t)w;
            im.height = (uint16_t)h;
            im.ncomp = (uint8_t)nc;
            for (int c = 0; c < nc; c++) {
                (void)r.u8();
                const int s = r.u8();
                im.hsamp[c] = (uint8_t)(s >> 4);
                im.vsamp[c] = (uint8_t)(s & 0xf);
                im.tq[c] = (uint8_t)r.u8();
                if (im.tq[c] > 3) return ZJ_E_FORMAT;
            }
            if (nc == 1) { // never interleaved: the sampling factors of a lone component do not matter
                im.hsamp[0] = im.vsamp[0] = 1;
                for (int c = 1; c < 3; c++) { im.hsamp[c] = im.vsamp[c] = 1; im.tq[c] = im.tq[0]; }
            }
            seen_sof = true;
            break;
        }
        case 0xdd: // DRI
            im.restart_interval = (uint16_t)r.u16();
            break;
        case 0xda: { // SOS (src/decoder.rs:366-370; the selectors the reference skips are read here)
            if (!seen_sof) return ZJ_E_FORMAT;
            const int ns = r.u8();
            if (ns != im.ncomp) return ZJ_E_UNSUPPORTED; // one scan with all components
            for (int c = 0; c < ns; c++) {
                (void)r.u8();
                const int t = r.u8();
                im.td[c] = (uint8_t)(t >> 4);
                im.ta[c] = (uint8_t)(t & 0xf);
                if (im.td[c] > 1 || im.ta[c] > 1) return ZJ_E_UNSUPPORTED;
            }
            for (int c = ns; c < 3; c++) { im.td[c] = im.td[0]; im.ta[c] = im.ta[0]; }
            r.pos = seg_end;
            // extent of the entropy-coded data: up to the first marker other than FF00 / RSTn
            size_t p = r.pos;
            uint32_t nrst = 0;
            while (p + 1 < len) {
                const uint8_t *ff = (const uint8_t *)memchr(file + p, 0xff, len - 1 - p); // only FF bytes matter
                if (!ff) { p = len; break; }
                p = (size_t)(ff - file);
                const uint8_t x = file[p + 1];
                if (x == 0x00) { p += 2; continue; }
                if (x >= 0xd0 && x <= 0xd7) { nrst++; p += 2; continue; }
                if (x == 0xff) { p += 1; continue; }
                break;
            }
            if (p + 1 >= len) p = len;
            im.scan = file + r.pos;
            im.scan_len = p - r.pos;
            out->n_restart_markers = nrst;
            for (int t = 0; t < 4; t++) im.qt[t] = qt_def[t] ? out->qt_store[t] : nullptr;
            for (int id = 0; id < 2; id++)
                for (int k = 0; k < 2; k++) im.ht[id][k] = ht_def[id][k] ? &out->ht_store[id][k] : nullptr;
            for (int c = 0; c < im.ncomp; c++) {
                if (!im.qt[im.tq[c]] || !im.ht[im.td[c]][1] || !im.ht[im.ta[c]][0]) return ZJ_E_FORMAT;
            }
            return zj_image_supported(&im) ? ZJ_OK : ZJ_E_UNSUPPORTED;
        }
        default:
            if (m >= 0xc1 && m <= 0xcf && m != 0xc8 && m != 0xcc) return ZJ_E_UNSUPPORTED; // other SOFn
            break; // APPn, COM, anything else with a length: skipped, src/decoder.rs:631-640
        }
        r.pos = seg_end;
    }
}

bool zj_image_supported(const zj_image *im)
{
    if (!im || (im->ncomp != 3 && im->ncomp != 1) || !im->width || !im->height || !im->scan) return false;
    if (im->ncomp == 1 && (im->hsamp[0] != 1 || im->vsamp[0] != 1)) return false;
    if (im->hsamp[0] < 1 || im->hsamp[0] > 2 || im->vsamp[0] < 1 || im->vsamp[0] > 2) return false;
    if (im->ncomp == 3 && (im->hsamp[1] != 1 || im->vsamp[1] != 1 || im->hsamp[2] != 1 || im->vsamp[2] != 1)) return false;
    for (int c = 0; c < im->ncomp; c++) {
        if (im->tq[c] > 3 || im->td[c] > 1 || im->ta[c] > 1) return false;
        if (!im->qt[im->tq[c]] || !im->ht[im->td[c]][1] || !im->ht[im->ta[c]][0]) return false;
    }
    if (im->scan_len >= (1ull << 28)) return false; // bit positions are 32-bit
    return true;
}

extern "C" uint32_t zj_image_mcus(const zj_image *im)
{
    const uint32_t mw = 8u * im->hsamp[0], mh = 8u * im->vsamp[0];
    return ((im->width + mw - 1) / mw) * ((im->height + mh - 1) / mh);
}

extern "C" uint32_t zj_image_blocks(const zj_image *im)
{
    return zj_image_mcus(im) * (im->ncomp == 1 ? 1u : (uint32_t)(im->hsamp[0] * im->vsamp[0] + 2));
}

extern "C" size_t zj_restart_offsets(const zj_image *im, size_t *offsets, size_t cap)
{
    size_t n = 0;
    if (n < cap) offsets[n] = 0;
    n++;
    const uint8_t *s = im->scan;
    size_t p = 0;
    while (p + 1 < im->scan_len) {
        const uint8_t *ff = (const uint8_t *)memchr(s + p, 0xff, im->scan_len - 1 - p);
        if (!ff) break;
        p = (size_t)(ff - s);
        if (s[p + 1] >= 0xd0 && s[p + 1] <= 0xd7) {
            if (n < cap) offsets[n] = p + 2;
            n++;
            p++;
        }
        p++;
    }
    return n;
}

extern "C" int zj_image_band(const zj_image *im, uint32_t row0, uint32_t nrows, zj_image *band)
{
    if (!im || !band || nrows == 0) return ZJ_E_INVALID;
    const uint32_t mw = 8u * im->hsamp[0], mh = 8u * im->vsamp[0];
    const uint32_t mcux = (im->width + mw - 1) / mw, mcuy = (im->height + mh - 1) / mh;
    if (row0 >= mcuy || row0 + nrows > mcuy) return ZJ_E_INVALID;
    const uint32_t ri = im->restart_interval;
    if (ri == 0) return ZJ_E_INVALID;
    // the band must begin and end on restart-interval boundaries
    const uint64_t m0 = (uint64_t)row0 * mcux, m1 = (uint64_t)(row0 + nrows) * mcux;
    if (m0 % ri != 0) return ZJ_E_INVALID;
    if (m1 % ri != 0 && row0 + nrows != mcuy) return ZJ_E_INVALID;
    const size_t k0 = (size_t)(m0 / ri), k1 = (size_t)((m1 + ri - 1) / ri);
    std::vector<size_t> offs(k1 + 2);
    const size_t nint = zj_restart_offsets(im, offs.data(), offs.size());
    if (k0 >= nint) return ZJ_E_INVALID;
    *band = *im;
    band->scan = im->scan + offs[k0];
    size_t end = im->scan_len;
    if (k1 < nint) end = offs[k1] - 2; // drop the RSTn that closes the band
    band->scan_len = end - offs[k0];
    const uint32_t y0 = row0 * mh;
    uint32_t hh = nrows * mh;
    if (y0 + hh > im->height) hh = im->height - y0;
    band->height = (uint16_t)hh;
    return ZJ_OK;
}

extern "C" int zj_write_ppm(const char *path, uint32_t width, uint32_t height, const uint8_t *rgb, size_t stride)
{
    if (!path || !rgb) return ZJ_E_INVALID;
    FILE *fp = fopen(path, "wb");
    if (!fp) return ZJ_E_IO;
    if (stride == 0) stride = (size_t)width * 3;
    bool ok = fprintf(fp, "P6\n%u %u\n255\n", width, height) > 0; // src/decoder.rs:580
    for (uint32_t y = 0; ok && y < height; y++) ok = fwrite(rgb + (size_t)y * stride, 1, (size_t)width * 3, fp) == (size_t)width * 3;
    ok = (fclose(fp) == 0) && ok;
    return ok ? ZJ_OK : ZJ_E_IO;
}

// ---------------------------------------------------------------------------------- lookup tables
// Canonical code assignment exactly as the table walk of src/decoder.rs:523-540 (code <<= 1 per
// length, +1 per symbol), laid out for the device as described in zj_core.h.
bool zj_build_table(const zj_huff *h, uint16_t *tbl, ZjCanon *canon, uint32_t l2_cap)
{
    if (l2_cap > ZJ_L2_SIZE) l2_cap = ZJ_L2_SIZE;
    memset(tbl, 0, sizeof(uint16_t) * ZJ_TBL_ENTRIES);
    memset(canon, 0, sizeof *canon);
    if (!h) return false;
    uint32_t code = 0, k = 0;
    uint16_t codes[256];
    uint8_t lens[256];
    for (uint32_t l = 1; l <= 16; l++) {
        canon->mincode[l] = (uint16_t)code;
        canon->valptr[l] = (uint16_t)k;
        canon->ncode[l] = h->counts[l - 1];
        for (uint32_t i = 0; i < h->counts[l - 1]; i++) {
            if (k >= 256 || code >= (1u << l)) return false; // over-subscribed: not a prefix code
            codes[k] = (uint16_t)code;
            lens[k] = (uint8_t)l;
            canon->symbols[k] = h->symbols[k];
            k++;
            code++;
        }
        code <<= 1;
    }
    canon->nsym = (uint16_t)k;
    if (k == 0) return false;
    // first level
    for (uint32_t i = 0; i < k; i++) {
        const uint32_t l = lens[i];
        if (l > ZJ_L1_BITS) continue;
        const uint32_t first = (uint32_t)codes[i] << (ZJ_L1_BITS - l);
        for (uint32_t j = 0; j < (1u << (ZJ_L1_BITS - l)); j++) tbl[first + j] = (uint16_t)((l << 8) | h->symbols[i]);
    }
    // second level: one sub-table per ZJ_L1_BITS-bit prefix that starts longer codes
    uint32_t used = 0;
    bool overflow = false;
    for (uint32_t i = 0; i < k && !overflow;) {
        if (lens[i] <= ZJ_L1_BITS) { i++; continue; }
        const uint32_t prefix = codes[i] >> (lens[i] - ZJ_L1_BITS);
        uint32_t j = i, maxlen = 0;
        while (j < k && lens[j] > ZJ_L1_BITS && (uint32_t)(codes[j] >> (lens[j] - ZJ_L1_BITS)) == prefix) {
            if (lens[j] > maxlen) maxlen = lens[j];
            j++;
        }
        const uint32_t nb = maxlen - ZJ_L1_BITS;
        if (used + (1u << nb) > l2_cap) { overflow = true; break; }
        tbl[prefix] = (uint16_t)(0x8000u | (nb << 12) | used);
        for (uint32_t t = i; t < j; t++) {
            const uint32_t extra = lens[t] - ZJ_L1_BITS;
            const uint32_t low = codes[t] & ((1u << extra) - 1u);
            const uint32_t first = low << (nb - extra);
            for (uint32_t u = 0; u < (1u << (nb - extra)); u++)
                tbl[ZJ_L1_SIZE + used + first + u] = (uint16_t)(((uint32_t)lens[t] << 8) | h->symbols[t]);
        }
        used += 1u << nb;
        i = j;
    }
    if (overflow) {
        // too many long codes for the second level: every long prefix goes to the canonical search
        for (uint32_t i = 0; i < k; i++) {
            if (lens[i] <= ZJ_L1_BITS) continue;
            tbl[codes[i] >> (lens[i] - ZJ_L1_BITS)] = 0x8000u;
        }
    }
    return true;
}

// Write entries of an AC table in the direct form (zj_core.h, ZJ_WT_DIRECT): returns false when the codes longer than
// the first level take more room than the form has, and the caller keeps the two-level layout.
bool zj_build_direct_table(const ZjCanon *canon, uint32_t *wt, uint32_t *t16)
{
    uint32_t first_long = 0x10000u;
    for (uint32_t l = ZJ_L1_BITS + 1; l <= 16 && first_long == 0x10000u; l++)
        if (canon->ncode[l]) first_long = (uint32_t)canon->mincode[l] << (16 - l);
    if (first_long != 0x10000u && ((first_long & 127u) != 0 || (0xFFFFu >> 7) + (0xFFFFu - first_long) >= ZJ_WT_DIRECT)) return false;
    const uint32_t none = zj_write_entry(0);
    for (uint32_t i = 0; i < ZJ_WT_DIRECT; i++) wt[i] = none;
    for (uint32_t l = 1; l <= 16; l++) {
        for (uint32_t i = 0; i < canon->ncode[l]; i++) {
            const uint32_t code = (uint32_t)canon->mincode[l] + i, sym = canon->symbols[(canon->valptr[l] + i) & 255u];
            const uint32_t entry = zj_write_entry((l << 8) | sym);
            const uint32_t x0 = code << (16 - l), nx = 1u << (16 - l); // the 16-bit values that begin with this code
            if (l <= ZJ_L1_BITS) {
                for (uint32_t x = x0; x < x0 + nx; x += 128u) wt[x >> 7] = entry;
            } else {
                for (uint32_t x = x0; x < x0 + nx; x++) wt[(x >> 7) + (x - first_long)] = entry;
            }
        }
    }
    *t16 = first_long;
    return true;
}

// The same table in walk-entry form (zj_core.h): long first-level entries keep their sub-table
// reference, every other entry turns into (bits to consume, zig-zag advance).
void zj_build_walk_table(const uint16_t *tbl, bool isdc, uint16_t *wtbl)
{
    for (uint32_t i = 0; i < ZJ_TBL_ENTRIES; i++) {
        const uint32_t e = tbl[i];
        if (i < ZJ_L1_SIZE && (e & 0x8000u)) wtbl[i] = (uint16_t)e;
        else wtbl[i] = (uint16_t)zj_walk_entry(e, isdc);
    }
}

// Pair entries: every first-level index is decoded symbol by symbol with the canonical description, as far
// as the index bits determine whole codes.  `partner` (DC tables only, may be null) is the AC table of the blocks
// that use this DC table: the DC symbol of a block is always followed by the block's first AC symbol, so a DC
// entry can hold that AC symbol as its second one when the DC code, its value bits and the AC code all lie inside
// the index bits (a flat block -- DC difference 0, end of block -- is one lookup then).
bool zj_build_pair_table(const ZjCanon *canon, bool isdc, uint32_t *ptbl, const ZjCanon *partner)
{
    struct Code { uint32_t code, len, sym; };
    std::vector<Code> codes, codes2;
    for (uint32_t l = 1; l <= 16; l++)
        for (uint32_t i = 0; i < canon->ncode[l]; i++)
            codes.push_back({(uint32_t)canon->mincode[l] + i, l, canon->symbols[(canon->valptr[l] + i) & 255u]});
    if (isdc && partner)
        for (uint32_t l = 1; l <= 16; l++)
            for (uint32_t i = 0; i < partner->ncode[l]; i++)
                codes2.push_back({(uint32_t)partner->mincode[l] + i, l, partner->symbols[(partner->valptr[l] + i) & 255u]});
    // the code that the top bits of the 32-bit window `x` begin with, among codes of at most `maxlen` bits
    auto match_in = [](const std::vector<Code> &cs, uint32_t x, uint32_t maxlen) -> const Code * {
        for (const Code &c : cs)
            if (c.len <= maxlen && (x >> (32u - c.len)) == c.code) return &c;
        return nullptr;
    };
    auto match = [&](uint32_t x, uint32_t maxlen) { return match_in(codes, x, maxlen); };
    auto single = [&](const Code *c) { return zj_pair_single(zj_walk_entry(c ? ((c->len << 8) | c->sym) : 0u, isdc)); };
    const uint32_t invalid = single(nullptr);
    for (uint32_t i = 0; i < ZJ_PT_ENTRIES; i++) ptbl[i] = invalid;
    for (uint32_t idx = 0; idx < ZJ_PL1_SIZE; idx++) {
        const uint32_t x = idx << (32 - ZJ_PL1_BITS);
        const Code *c1 = match(x, ZJ_PL1_BITS);
        if (!c1) continue; // no code, or the prefix of longer codes (entered below)
        uint32_t e = single(c1);
        const uint32_t t1 = e & 255u, a1 = (e >> 16) & 255u;
        const bool valid_dc = isdc && c1->sym <= 16u && !codes2.empty(); // a DC symbol above 16 is no symbol: the write pass reports it
        if (((!isdc && a1 < 64u) || valid_dc) && t1 < (uint32_t)ZJ_PL1_BITS) {
            const Code *c2 = match_in(isdc ? codes2 : codes, x << t1, ZJ_PL1_BITS - t1);
            if (c2) {
                const uint32_t w2 = zj_walk_entry((c2->len << 8) | c2->sym, false);
                const uint32_t t12 = t1 + (w2 & 63u), a12 = a1 + ((w2 >> 6) & 127u);
                if (t12 <= 32u && a12 < 128u) e = t1 | (t12 << 8) | (a1 << 16) | (a12 << 24);
            }
        }
        ptbl[idx] = e;
    }
    // second level: one sub-table per first-level prefix that begins longer codes
    uint32_t used = 0;
    bool fits = true;
    for (size_t i = 0; i < codes.size() && fits;) {
        if (codes[i].len <= (uint32_t)ZJ_PL1_BITS) { i++; continue; }
        const uint32_t prefix = codes[i].code >> (codes[i].len - ZJ_PL1_BITS);
        size_t j = i;
        uint32_t maxlen = 0;
        while (j < codes.size() && codes[j].len > (uint32_t)ZJ_PL1_BITS && (codes[j].code >> (codes[j].len - ZJ_PL1_BITS)) == prefix) {
            if (codes[j].len > maxlen) maxlen = codes[j].len;
            j++;
        }
        const uint32_t nb = maxlen - ZJ_PL1_BITS;
        if (used + (1u << nb) > (uint32_t)ZJ_PL2_SIZE) { fits = false; break; }
        ptbl[prefix] = 0x80000000u | (nb << 24) | used;
        for (size_t t = i; t < j; t++) {
            const uint32_t extra = codes[t].len - ZJ_PL1_BITS;
            const uint32_t first = (codes[t].code & ((1u << extra) - 1u)) << (nb - extra);
            for (uint32_t u = 0; u < (1u << (nb - extra)); u++) ptbl[ZJ_PL1_SIZE + used + first + u] = single(&codes[t]);
        }
        used += 1u << nb;
        i = j;
    }
    if (!fits) {
        for (const Code &c : codes)
            if (c.len > (uint32_t)ZJ_PL1_BITS) ptbl[c.code >> (c.len - ZJ_PL1_BITS)] = 0x80000000u;
    }
    return fits;
}
