/*
 * zj_oracle.c -- CPU restatement of the reference decode path (TEST INFRASTRUCTURE ONLY).
 *
 * This file is the checker for the HIP decode path.  It is never linked into, imported by,
 * or called from the shipped library; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may use it.
 *
 * PARITY UNPINNED: the reference (a Rust crate) ships no tests, golden vectors or
 * known-answer fixtures for this path, and no Rust toolchain exists in the build image, so
 * the reference itself cannot be executed.  This restatement follows the reference's source
 * text function by function (citations below, all into /root/reference/), keeps its
 * `--release` integer semantics (i16 add/mul wrap) and its exact f32 operation order, and is
 * cross-checked only against structural invariants of the reference's sample images and,
 * loosely, against libjpeg.
 *
 * Build:  gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC   (see oracle/Makefile)
 * -ffp-contract=off matters: the reference fuses multiply-add only where it calls
 * f32::mul_add (src/dct.rs:95); everything in src/decoder.rs:271-273 is separate mul and add.
 *
 * Behaviour the reference does not define, and how it is defined here (each is a documented
 * deviation, tested on its own):
 *   E1  a DHT whose codes all share one length leaves `cht` empty and the reference indexes
 *       cht[0] out of bounds (src/ec.rs:22 with src/decoder.rs:517-540).  Here: every code of
 *       such a table has length l0, i.e. the short path of src/ec.rs:23-27 is taken always.
 *   E2  sampling factors other than 1x1 (the reference reads and drops them,
 *       src/decoder.rs:612-613).  Here: T.81 A.2.3 interleaved MCU order, ceil() MCU counts,
 *       chroma upsampling by sample replication, crop to w x h, colour maths unchanged.
 *   E3  DRI / RSTn (the reference panics on DRI, src/decoder.rs:358, and silently ends the bit
 *       stream at RSTn, src/bitstream.rs:44-46).  Here: every restart interval is decoded as an
 *       independent entropy-coded segment with a fresh BitReader and dc_pred = 0.
 *   E4  component -> table selectors are read from the SOS header instead of being hard-wired
 *       Y->0, C->1 (src/decoder.rs:171-173, :368-370).  Identical on the reference's images.
 *   E5  widths/heights that are not multiples of the MCU size: ceil() blocks are decoded and the
 *       result cropped (the reference floors and skews rows, src/decoder.rs:284-285,314).
 *   E6  one-component (grayscale) frames, which the reference admits at src/decoder.rs:574 but cannot
 *       decode (:170-173): single blocks in raster order, pixels as for Cb = Cr = 128 (R = G = B).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ZJO_OK 0
#define ZJO_E_FORMAT (-1)      /* malformed container / unsupported marker layout           */
#define ZJO_E_UNSUPPORTED (-2) /* valid JPEG outside the path (progressive, 16-bit DQT ...)  */
#define ZJO_E_HUFF (-3)        /* symbol index out of table (reference: slice OOB panic)     */
#define ZJO_E_COEF (-4)        /* coefficient index >= 64 (reference: slice OOB panic)       */
#define ZJO_E_TRUNC (-5)       /* ran out of bits (reference: unwrap panic / u32 underflow)  */

/* ---------------------------------------------------------------- memory reader
 * Stand-in for BufReader<File> + read_u8/read_u16 (src/bitstream.rs:6-19). */
typedef struct {
    const uint8_t *p;
    size_t len, pos;
} rd_t;

static int rd_u8(rd_t *r) { return r->pos < r->len ? r->p[r->pos++] : -1; }
static int rd_u16(rd_t *r)
{
    if (r->pos + 2 > r->len) { r->pos = r->len; return -1; }
    int v = (r->p[r->pos] << 8) | r->p[r->pos + 1];
    r->pos += 2;
    return v;
}

/* ---------------------------------------------------------------- BitReader
 * src/bitstream.rs:21-113.  bitbuf is left-aligned, bitlen counts valid bits. */
typedef struct {
    rd_t *r;
    uint64_t bitbuf;
    uint32_t bitlen;
    int err;
} br_t;

/* src/bitstream.rs:37-50: one payload byte; FF 00 -> FF; FF xx (xx != 0) -> None with both
 * bytes consumed; end of data -> None. */
static int br_byte_refill(br_t *b)
{
    int nb = rd_u8(b->r);
    if (nb < 0) return -1;
    if (nb == 0xff) {
        int nx = rd_u8(b->r);
        if (nx < 0) return -1;
        if (nx != 0x00) return -1;
    }
    return nb;
}

/* src/bitstream.rs:76-92 with BITS = 16: zero-padded look-ahead. */
static uint16_t br_peek16(br_t *b)
{
    while (b->bitlen < 16) {
        int byte = br_byte_refill(b);
        if (byte < 0) break;
        /* (byte as u64).rotate_right(8) >> bitlen */
        b->bitbuf |= ((uint64_t)byte << 56) >> b->bitlen;
        b->bitlen += 8;
    }
    return (uint16_t)(b->bitbuf >> 48);
}

/* src/bitstream.rs:94-97.  The reference lets bitlen wrap below zero; here that is an error. */
static void br_consume(br_t *b, uint32_t bits)
{
    if (bits > b->bitlen) { b->err = ZJO_E_TRUNC; b->bitlen = 0; b->bitbuf = 0; return; }
    b->bitbuf <<= bits;
    b->bitlen -= bits;
}

/* src/bitstream.rs:99-113.  None (-> unwrap panic in the caller) becomes ZJO_E_TRUNC. */
static uint16_t br_get_n_bits(br_t *b, uint32_t bits)
{
    if (bits > 16) { b->err = ZJO_E_HUFF; return 0; }
    while (b->bitlen < bits) {
        int byte = br_byte_refill(b);
        if (byte < 0) { b->err = ZJO_E_TRUNC; return 0; }
        b->bitbuf |= ((uint64_t)byte << 56) >> b->bitlen;
        b->bitlen += 8;
    }
    if (bits == 0) return 0; /* rotate_left(0) as u16 of a left-aligned buffer with <= 23 bits */
    uint16_t code = (uint16_t)(b->bitbuf >> (64 - bits));
    b->bitbuf <<= bits;
    b->bitlen -= bits;
    return code;
}

/* ---------------------------------------------------------------- HuffmanTree
 * src/ec.rs:3-8: condensed canonical table.  cht[i] = (first code of length L_i left-aligned
 * in 16 bits, L_i, index of its symbol) for every length > l0 that occurs. */
typedef struct {
    uint8_t symbols[256];
    int nsym;
    uint16_t cht_aug[16];
    uint8_t cht_len[16];
    uint8_t cht_off[16];
    int ncht;
    uint8_t l0;
    int defined;
} hufftree_t;

/* src/decoder.rs:497-553 (the table walk; reading the bytes is done by the caller). */
static int huff_build(hufftree_t *ht, const uint8_t counts[16], const uint8_t *symbols, int nsym)
{
    if (nsym <= 0 || nsym > 256) return ZJO_E_FORMAT; /* assert at src/decoder.rs:509 */
    memset(ht, 0, sizeof *ht);
    memcpy(ht->symbols, symbols, (size_t)nsym);
    ht->nsym = nsym;
    int last_len = 0;
    for (int i = 0; i < 16; i++)
        if (counts[i] > 0) { last_len = i + 1; break; }
    ht->l0 = (uint8_t)last_len;
    uint16_t code = 0;
    int bits = 0, idx = 0;
    for (int d = 0; d < 16; d++) {
        code = (uint16_t)(code << 1);
        bits += 1;
        for (int k = 0; k < counts[d]; k++) {
            if (bits > last_len) {
                ht->cht_aug[ht->ncht] = (uint16_t)(code << (16 - bits));
                ht->cht_len[ht->ncht] = (uint8_t)bits;
                ht->cht_off[ht->ncht] = (uint8_t)idx; /* `idx as u8` */
                ht->ncht++;
                last_len = bits;
            }
            idx += 1;
            code = (uint16_t)(code + 1);
        }
    }
    ht->defined = 1;
    return ZJO_OK;
}

/* src/ec.rs:19-49, plus E1 for an empty cht. */
static int huff_read_code(const hufftree_t *ht, br_t *b, uint64_t *nsymbols, uint64_t *nshort)
{
    uint16_t w = br_peek16(b);
    (*nsymbols)++;
    if (ht->ncht == 0 || w < ht->cht_aug[0]) {
        w = (uint16_t)(w >> (16 - ht->l0));
        br_consume(b, ht->l0);
        (*nshort)++;
        if (w >= ht->nsym) { b->err = ZJO_E_HUFF; return 0; }
        return ht->symbols[w];
    }
    int j = -1;
    for (int i = 1; i < ht->ncht; i++) {
        if (ht->cht_aug[i] > w) { j = i - 1; break; }
    }
    if (j < 0) j = ht->ncht - 1;
    uint16_t aug = ht->cht_aug[j];
    int l = ht->cht_len[j];
    int off = ht->cht_off[j];
    w = (uint16_t)(w >> (16 - l));
    br_consume(b, (uint32_t)l);
    int base = aug >> (16 - l);
    int si = (int)w - base + off;
    if (si < 0 || si >= ht->nsym) { b->err = ZJO_E_HUFF; return 0; }
    return ht->symbols[si];
}

/* src/ec.rs:52-59 (JPEG EXTEND), i16 arithmetic wrapping as in --release; the shift count of
 * `1 << n_bits` is masked to the i16 width as release Rust does. */
static int16_t sign_code(uint32_t n_bits, uint16_t code)
{
    if (n_bits < 32 && ((((uint32_t)code) << 1) >> n_bits) != 0) return (int16_t)code;
    if (n_bits >= 32) return (int16_t)code;
    int16_t max_val = (int16_t)((int16_t)(1u << (n_bits & 15)) - 1);
    return (int16_t)((int16_t)code - max_val);
}

/* src/decoder.rs:114-124 is the inverse zig-zag map (natural index -> zig-zag index).  It is
 * generated here by walking the anti-diagonals, which is how scan.py:14-29 derives it. */
static uint8_t g_nat2zz[64];
static float g_cos[64];
static int g_init_done = 0;

static void oracle_init(void)
{
    if (g_init_done) return;
    int zz = 0;
    for (int s = 0; s < 15; s++) {
        if (s & 1) { /* odd diagonal: top-right -> bottom-left */
            for (int r = (s < 8 ? 0 : s - 7); r <= (s < 8 ? s : 7); r++) g_nat2zz[r * 8 + (s - r)] = (uint8_t)zz++;
        } else { /* even diagonal: bottom-left -> top-right */
            for (int r = (s < 8 ? s : 7); r >= (s < 8 ? 0 : s - 7); r--) g_nat2zz[r * 8 + (s - r)] = (uint8_t)zz++;
        }
    }
    /* src/dct.rs:19-84: COS_TABLE[8n+k] = cos((2n+1) k pi / 16) rounded to f32, with the k = 0
     * column replaced by 1/sqrt(2).  The seven magnitudes are fixed here by their f32 bit
     * patterns so that no compiler's decimal or libm rounding can move them. */
    static const uint32_t mag_bits[8] = {
        0x3f800000u, /* cos(0)        (unused: k = 0 column is 1/sqrt 2) */
        0x3f7b14beu, /* cos(1 pi/16) = 0.98078528 */
        0x3f6c835eu, /* cos(2 pi/16) = 0.92387953 */
        0x3f54db31u, /* cos(3 pi/16) = 0.83146961 */
        0x3f3504f3u, /* cos(4 pi/16) = 0.70710678 */
        0x3f0e39dau, /* cos(5 pi/16) = 0.55557023 */
        0x3ec3ef15u, /* cos(6 pi/16) = 0.38268343 */
        0x3e47c5c2u, /* cos(7 pi/16) = 0.19509032 */
    };
    for (int n = 0; n < 8; n++)
        for (int k = 0; k < 8; k++) {
            uint32_t bits;
            if (k == 0) bits = mag_bits[4];
            else {
                int m = ((2 * n + 1) * k) & 31; /* angle m pi/16, period 32 */
                int neg = 0;
                if (m > 16) m = 32 - m;          /* cos(2pi - x) = cos x  */
                if (m > 8) { m = 16 - m; neg = 1; } /* cos(pi - x) = -cos x */
                bits = (m == 8) ? 0u : mag_bits[m]; /* m == 8 cannot occur for odd (2n+1)k... kept for safety */
                if (neg) bits |= 0x80000000u;
            }
            memcpy(&g_cos[8 * n + k], &bits, 4);
        }
    g_init_done = 1;
}

/* src/dct.rs:87-99 */
static void idct_1d(const float *in, float *out)
{
    for (int n = 0; n < 8; n++) {
        float sum = 0.0f;
        for (int k = 0; k < 8; k++) sum = fmaf(in[k], g_cos[8 * n + k], sum);
        out[n] = sum * 0.5f;
    }
}

/* src/dct.rs:11-17 */
static void transpose8x8(const float *in, float *out)
{
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) out[j * 8 + i] = in[i * 8 + j];
}

/* src/dct.rs:101-125 */
static void idct(const float *in, float *out)
{
    float t[64];
    transpose8x8(in, t);
    for (int i = 0; i < 8; i++) idct_1d(&t[8 * i], &out[8 * i]);
    transpose8x8(out, t);
    for (int i = 0; i < 8; i++) idct_1d(&t[8 * i], &out[8 * i]);
}

/* Rust `f32 as u8`: truncate toward zero, saturate, NaN -> 0. */
static uint8_t f32_as_u8(float v)
{
    if (!(v == v)) return 0;
    if (v <= 0.0f) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)(int)v;
}

/* src/decoder.rs:265-280 */
static void ycbcr_to_rgb(float y, float cb, float cr, uint8_t *px)
{
    float r = 1.402f * (cr - 128.0f) + y;
    float g = -0.71414f * (cr - 128.0f) + -0.34414f * (cb - 128.0f) + y;
    float b = 1.772f * (cb - 128.0f) + y;
    px[0] = f32_as_u8(r);
    px[1] = f32_as_u8(g);
    px[2] = f32_as_u8(b);
}

/* ---------------------------------------------------------------- parsed frame */
typedef struct {
    int w, h, ncomp;
    int hs[3], vs[3], tq[3], td[3], ta[3];
    int hmax, vmax;
    int restart_interval;
    uint8_t qt[4][64]; /* file (zig-zag) order, src/decoder.rs:453 */
    int qt_defined[4];
    hufftree_t ht[4][2]; /* [table id][0 = AC, 1 = DC]  (src/decoder.rs:553) */
    size_t scan_off, scan_len;
} frame_t;

typedef struct {
    int32_t status;
    int32_t w, h, ncomp;
    int32_t hs[3], vs[3];
    int32_t mcux, mcuy, blocks_per_mcu;
    int32_t restart_interval;
    int64_t nblocks;
    int64_t scan_off, scan_len; /* entropy-coded bytes after the SOS header, up to the terminating marker */
    int64_t scan_consumed;      /* bytes of the scan the bit readers pulled (reference invariant)        */
    uint64_t nsymbols;          /* Huffman symbols decoded (DC + AC)                                      */
    uint64_t nshort;            /* of those, taken on the short path of src/ec.rs:22-27                    */
    uint64_t nff00;             /* FF 00 pairs inside the scan                                             */
    int64_t err_block;          /* scan-order index of the block whose decode failed (status HUFF/COEF/TRUNC), -1: none.
                                 * The reference panics inside that block (src/decoder.rs:194-228, src/ec.rs:27,47);
                                 * every block before it is complete in qcoef / dqcoef.                    */
} zjo_info;

/* Marker walk: src/decoder.rs:347-642.  All APPn / COM are skipped by length (the reference
 * knows only five of them), DRI is accepted (E3), and the SOS header is read (E4). */
static int parse_headers(const uint8_t *file, size_t len, frame_t *f)
{
    rd_t r = {file, len, 0};
    memset(f, 0, sizeof *f);
    int seen_sof = 0;
    for (;;) {
        int marker = rd_u16(&r);
        if (marker < 0) return ZJO_E_FORMAT;
        if ((marker >> 8) != 0xff) return ZJO_E_FORMAT; /* src/decoder.rs:62-64 */
        int m = marker & 0xff;
        if (m == 0xd8) continue;              /* SOI */
        if (m == 0xd9) return ZJO_E_FORMAT;   /* EOI before SOS */
        int seglen = rd_u16(&r);
        if (seglen < 2 || r.pos + (size_t)(seglen - 2) > len) return ZJO_E_FORMAT;
        size_t seg_end = r.pos + (size_t)(seglen - 2);
        if (m == 0xdb) { /* DQT, src/decoder.rs:428-466: several tables per segment */
            while (r.pos < seg_end) {
                int info = rd_u8(&r);
                int dst = info & 0xf;
                if (dst > 3) return ZJO_E_FORMAT;
                if ((info & 0xf0) != 0) return ZJO_E_UNSUPPORTED; /* 16-bit, assert at :450 */
                if (r.pos + 64 > seg_end) return ZJO_E_FORMAT;
                memcpy(f->qt[dst], r.p + r.pos, 64);
                f->qt_defined[dst] = 1;
                r.pos += 64;
            }
        } else if (m == 0xc4) { /* DHT, src/decoder.rs:467-562 */
            while (r.pos < seg_end) {
                int info = rd_u8(&r);
                int id = info & 0xf;
                if (id > 3 || (info & 0xe0)) return ZJO_E_FORMAT; /* asserts at :478, :486 */
                int is_dc = ((info >> 4) & 1) == 0;
                if (r.pos + 16 > seg_end) return ZJO_E_FORMAT;
                const uint8_t *counts = r.p + r.pos;
                r.pos += 16;
                int n = 0;
                for (int i = 0; i < 16; i++) n += counts[i];
                if (n <= 0 || n > 256 || r.pos + (size_t)n > seg_end) return ZJO_E_FORMAT;
                int rc = huff_build(&f->ht[id][is_dc], counts, r.p + r.pos, n);
                if (rc) return rc;
                r.pos += (size_t)n;
            }
        } else if (m == 0xc0) { /* SOF0, src/decoder.rs:564-630 */
            int prec = rd_u8(&r);
            int h = rd_u16(&r), w = rd_u16(&r);
            int nc = rd_u8(&r);
            if (prec != 8) return ZJO_E_UNSUPPORTED;
            /* The reference admits 1 or 3 components (:574) but its decode loop is 3-component only
             * (:170-173).  Defined extension E6: a one-component frame is a sequence of single blocks in
             * raster order (T.81 A.2.2), and its pixels are those of a colour image with Cb = Cr = 128. */
            if (nc != 3 && nc != 1) return ZJO_E_UNSUPPORTED;
            if (w <= 0 || h <= 0) return ZJO_E_FORMAT;
            f->w = w; f->h = h; f->ncomp = nc;
            for (int c = 0; c < nc; c++) {
                (void)rd_u8(&r); /* component id */
                int s = rd_u8(&r);
                f->hs[c] = s >> 4; f->vs[c] = s & 0xf;
                f->tq[c] = rd_u8(&r);
                if (f->hs[c] < 1 || f->hs[c] > 2 || f->vs[c] < 1 || f->vs[c] > 2 || f->tq[c] > 3) return ZJO_E_UNSUPPORTED;
                if (f->hs[c] > f->hmax) f->hmax = f->hs[c];
                if (f->vs[c] > f->vmax) f->vmax = f->vs[c];
            }
            if (nc == 1) { /* a single component is never interleaved: its sampling factors do not matter */
                f->hs[0] = f->vs[0] = 1;
                f->hmax = f->vmax = 1;
            } else if (f->hs[1] != 1 || f->vs[1] != 1 || f->hs[2] != 1 || f->vs[2] != 1) return ZJO_E_UNSUPPORTED;
            seen_sof = 1;
        } else if (m >= 0xc1 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc) {
            return ZJO_E_UNSUPPORTED; /* any other SOFn */
        } else if (m == 0xdd) { /* DRI (E3) */
            f->restart_interval = rd_u16(&r);
        } else if (m == 0xda) { /* SOS, src/decoder.rs:366-370 (+E4) */
            if (!seen_sof) return ZJO_E_FORMAT;
            int ns = rd_u8(&r);
            if (ns != f->ncomp) return ZJO_E_UNSUPPORTED; /* single interleaved scan only */
            for (int c = 0; c < ns; c++) {
                (void)rd_u8(&r);
                int t = rd_u8(&r);
                f->td[c] = t >> 4; f->ta[c] = t & 0xf;
                if (f->td[c] > 3 || f->ta[c] > 3) return ZJO_E_FORMAT;
            }
            r.pos = seg_end;
            f->scan_off = r.pos;
            /* scan extent: up to the first marker that is neither FF00 nor RSTn */
            size_t p = r.pos;
            while (p + 1 < len) {
                if (file[p] == 0xff) {
                    uint8_t x = file[p + 1];
                    if (x == 0x00 || (x >= 0xd0 && x <= 0xd7)) { p += 2; continue; }
                    if (x == 0xff) { p += 1; continue; } /* fill byte */
                    break;
                }
                p++;
            }
            if (p + 1 >= len) p = len; /* no terminating marker: the data simply ends */
            f->scan_len = p - f->scan_off;
            return ZJO_OK;
        }
        r.pos = seg_end; /* everything else (APPn, COM, ...) skipped by length, :631-640 */
    }
}

/* src/decoder.rs:186-241.  qz receives the block before dequantisation (zig-zag order, DC after
 * prediction); nat receives the reference's own result (dequantised, natural order). */
static void decode_dct_matrix(const hufftree_t *ac, const hufftree_t *dc, const uint8_t *q, br_t *b, int16_t *dc_pred,
                              int16_t *qz, int16_t *nat, uint64_t *nsym, uint64_t *nshort)
{
    int dc_bits = huff_read_code(dc, b, nsym, nshort);
    uint16_t dc_val = br_get_n_bits(b, (uint32_t)dc_bits);
    int16_t dc_coeff = (int16_t)(*dc_pred + sign_code((uint32_t)dc_bits, dc_val));
    *dc_pred = dc_coeff;
    int16_t blk[64];
    memset(blk, 0, sizeof blk);
    blk[0] = dc_coeff;
    int idx = 1;
    for (;;) {
        if (b->err) break;
        int symbol = huff_read_code(ac, b, nsym, nshort);
        if (b->err) break;
        if (symbol == 0) break; /* EOB */
        int ac_bits = symbol & 0xf;
        int run = symbol >> 4;
        uint16_t v = br_get_n_bits(b, (uint32_t)ac_bits);
        int16_t coef = sign_code((uint32_t)ac_bits, v);
        idx += run;
        if (idx >= 64) { b->err = ZJO_E_COEF; break; } /* reference: index panic at :228 */
        blk[idx] = coef;
        idx += 1;
        if (idx >= 64) break;
    }
    memcpy(qz, blk, sizeof blk);
    /* dequantize, src/decoder.rs:179-184: both operands in zig-zag order, i16 wrapping */
    for (int i = 0; i < 64; i++) blk[i] = (int16_t)(blk[i] * (int16_t)q[i]);
    /* zigzag_descan, src/decoder.rs:137-146 */
    for (int i = 0; i < 64; i++) nat[i] = blk[g_nat2zz[i]];
}

/* One restart interval (or the whole scan when there is none): src/decoder.rs:374-389 with the
 * MCU layout generalised per E2. */
static int decode_segment(const frame_t *f, const uint8_t *seg, size_t seglen, int64_t mcu0, int64_t nmcu,
                          int bpm, const int *blk_comp, int16_t *qcoef, int16_t *dqcoef, zjo_info *info)
{
    rd_t r = {seg, seglen, 0};
    br_t b = {&r, 0, 0, 0};
    int16_t dc_pred[3] = {0, 0, 0};
    for (int64_t m = mcu0; m < mcu0 + nmcu; m++) {
        for (int k = 0; k < bpm; k++) {
            int c = blk_comp[k];
            int64_t g = m * bpm + k;
            decode_dct_matrix(&f->ht[f->ta[c]][0], &f->ht[f->td[c]][1], f->qt[f->tq[c]], &b, &dc_pred[c],
                              qcoef + g * 64, dqcoef + g * 64, &info->nsymbols, &info->nshort);
            if (b.err) { info->err_block = g; return b.err; }
        }
    }
    info->scan_consumed += (int64_t)r.pos;
    return ZJO_OK;
}

/* src/decoder.rs:282-320 generalised per E2/E5: for 4:4:4 with w % 8 == 0 the loop below performs
 * exactly the reference's operations on exactly the reference's operands. */
static void to_rgb(const frame_t *f, int mcux, int mcuy, int bpm, const int16_t *dqcoef, uint8_t *rgb)
{
    int mw = 8 * f->hmax, mh = 8 * f->vmax;
    float coeffs[64], out[6][64];
    (void)mcuy;
    for (int my = 0; my < mcuy; my++)
        for (int mx = 0; mx < mcux; mx++) {
            const int16_t *mb = dqcoef + ((int64_t)(my * mcux + mx) * bpm) * 64;
            for (int k = 0; k < bpm; k++) {
                for (int i = 0; i < 64; i++) coeffs[i] = (float)mb[k * 64 + i];
                idct(coeffs, out[k]);
            }
            int ny = f->hs[0] * f->vs[0]; /* luma blocks come first, then Cb, Cr */
            for (int y2 = 0; y2 < mh; y2++) {
                int py = my * mh + y2;
                if (py >= f->h) break;
                for (int x2 = 0; x2 < mw; x2++) {
                    int px = mx * mw + x2;
                    if (px >= f->w) break;
                    int yb = (y2 / 8) * f->hs[0] + (x2 / 8);
                    float yp = out[yb][(y2 % 8) * 8 + (x2 % 8)] + 128.0f;
                    int cy = y2 / f->vmax, cx = x2 / f->hmax; /* chroma is 1x1: replicate */
                    float cb = 128.0f, cr = 128.0f;           /* E6: no chroma */
                    if (f->ncomp == 3) {
                        cb = out[ny][cy * 8 + cx] + 128.0f;
                        cr = out[ny + 1][cy * 8 + cx] + 128.0f;
                    }
                    ycbcr_to_rgb(yp, cb, cr, rgb + 3 * ((size_t)py * f->w + px));
                }
            }
        }
}

static int frame_geometry(const frame_t *f, int *mcux, int *mcuy, int *bpm, int *blk_comp)
{
    *mcux = (f->w + 8 * f->hmax - 1) / (8 * f->hmax);
    *mcuy = (f->h + 8 * f->vmax - 1) / (8 * f->vmax);
    int n = 0;
    for (int c = 0; c < f->ncomp; c++)
        for (int k = 0; k < f->hs[c] * f->vs[c]; k++) blk_comp[n++] = c;
    *bpm = n;
    for (int c = 0; c < f->ncomp; c++) {
        if (!f->qt_defined[f->tq[c]]) return ZJO_E_FORMAT;
        if (!f->ht[f->td[c]][1].defined || !f->ht[f->ta[c]][0].defined) return ZJO_E_FORMAT;
    }
    return ZJO_OK;
}

/* ---------------------------------------------------------------- public entry points */

/* Parse only: geometry and scan extent, so that callers can size their buffers. */
int zjo_probe(const uint8_t *file, size_t len, zjo_info *info)
{
    oracle_init();
    frame_t f;
    memset(info, 0, sizeof *info);
    int rc = parse_headers(file, len, &f);
    info->status = rc;
    if (rc) return rc;
    int mcux, mcuy, bpm, blk_comp[8];
    rc = frame_geometry(&f, &mcux, &mcuy, &bpm, blk_comp);
    info->status = rc;
    if (rc) return rc;
    info->w = f.w; info->h = f.h; info->ncomp = f.ncomp;
    for (int c = 0; c < 3; c++) { info->hs[c] = f.hs[c]; info->vs[c] = f.vs[c]; }
    info->mcux = mcux; info->mcuy = mcuy; info->blocks_per_mcu = bpm;
    info->restart_interval = f.restart_interval;
    info->nblocks = (int64_t)mcux * mcuy * bpm;
    info->err_block = -1;
    info->scan_off = (int64_t)f.scan_off; info->scan_len = (int64_t)f.scan_len;
    return ZJO_OK;
}

/* Full decode.  Any of qcoef / dqcoef / rgb may be NULL.
 *   qcoef  [nblocks][64] i16, zig-zag order, before dequantisation, DC after prediction
 *   dqcoef [nblocks][64] i16, natural order, dequantised  (== the reference's `blocks`)
 *   rgb    [h][w][3] u8                                    (== the reference's out.ppm payload)
 * Block order is scan order: MCU-major, inside an MCU the luma blocks row by row, then Cb, Cr. */
int zjo_decode(const uint8_t *file, size_t len, int16_t *qcoef, int16_t *dqcoef, uint8_t *rgb, zjo_info *info)
{
    int rc = zjo_probe(file, len, info);
    if (rc) return rc;
    frame_t f;
    parse_headers(file, len, &f);
    int mcux, mcuy, bpm, blk_comp[8];
    frame_geometry(&f, &mcux, &mcuy, &bpm, blk_comp);
    int64_t nblocks = info->nblocks;
    int16_t *q_own = NULL, *d_own = NULL;
    if (!qcoef) qcoef = q_own = (int16_t *)malloc((size_t)nblocks * 128);
    if (!dqcoef) dqcoef = d_own = (int16_t *)malloc((size_t)nblocks * 128);
    if (!qcoef || !dqcoef) { free(q_own); free(d_own); return info->status = ZJO_E_FORMAT; }

    const uint8_t *scan = file + f.scan_off;
    size_t slen = f.scan_len;
    for (size_t i = 0; i + 1 < slen; i++)
        if (scan[i] == 0xff && scan[i + 1] == 0x00) info->nff00++;
    int64_t total_mcu = (int64_t)mcux * mcuy;
    if (f.restart_interval == 0) {
        rc = decode_segment(&f, scan, slen, 0, total_mcu, bpm, blk_comp, qcoef, dqcoef, info);
    } else {
        /* E3: cut at every RSTn; the marker bytes belong to no segment */
        size_t start = 0;
        int64_t mcu = 0;
        rc = ZJO_OK;
        while (mcu < total_mcu && rc == ZJO_OK) {
            size_t p = start;
            while (p + 1 < slen && !(scan[p] == 0xff && scan[p + 1] >= 0xd0 && scan[p + 1] <= 0xd7)) p++;
            size_t end = (p + 1 < slen) ? p : slen;
            int64_t n = total_mcu - mcu < f.restart_interval ? total_mcu - mcu : f.restart_interval;
            rc = decode_segment(&f, scan + start, end - start, mcu, n, bpm, blk_comp, qcoef, dqcoef, info);
            mcu += n;
            start = end + 2;
            if (start > slen) start = slen;
        }
    }
    info->status = rc;
    if (rc == ZJO_OK && rgb) to_rgb(&f, mcux, mcuy, bpm, dqcoef, rgb);
    free(q_own);
    free(d_own);
    return rc;
}

/* Stage-2/3 only: dequantised natural-order blocks -> RGB, for checking the IDCT/colour kernel in
 * isolation on synthetic coefficient blocks (hs0/vs0 = luma sampling, chroma 1x1). */
int zjo_blocks_to_rgb(int w, int h, int hs0, int vs0, const int16_t *dqcoef, uint8_t *rgb)
{
    oracle_init();
    frame_t f;
    memset(&f, 0, sizeof f);
    f.w = w; f.h = h; f.ncomp = 3;
    f.hs[0] = hs0; f.vs[0] = vs0; f.hs[1] = f.vs[1] = f.hs[2] = f.vs[2] = 1;
    f.hmax = hs0; f.vmax = vs0;
    int mcux = (w + 8 * hs0 - 1) / (8 * hs0), mcuy = (h + 8 * vs0 - 1) / (8 * vs0);
    to_rgb(&f, mcux, mcuy, hs0 * vs0 + 2, dqcoef, rgb);
    return ZJO_OK;
}

/* Single pieces, exported for known-answer tests. */
void zjo_idct(const float *in, float *out) { oracle_init(); idct(in, out); }
int16_t zjo_sign_code(uint32_t n, uint16_t code) { return sign_code(n, code); }
void zjo_ycbcr_to_rgb(float y, float cb, float cr, uint8_t *px) { ycbcr_to_rgb(y, cb, cr, px); }
void zjo_tables(uint8_t *nat2zz, float *cos_table)
{
    oracle_init();
    memcpy(nat2zz, g_nat2zz, 64);
    memcpy(cos_table, g_cos, sizeof g_cos);
}

/* PPM writer: src/decoder.rs:580 (header) and :648 (payload). */
int zjo_write_ppm(const char *path, int w, int h, const uint8_t *rgb)
{
    FILE *fp = fopen(path, "wb");
    if (!fp) return -1;
    fprintf(fp, "P6\n%d %d\n255\n", w, h);
    size_t n = (size_t)w * h * 3;
    int ok = fwrite(rgb, 1, n, fp) == n;
    fclose(fp);
    return ok ? 0 : -1;
}
