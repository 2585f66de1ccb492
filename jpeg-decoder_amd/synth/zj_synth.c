/*
 * zj_synth.c -- deterministic synthetic baseline-JPEG generator (workload generator for the
 * benchmark and the tests; CPU only, no part of the decode path).
 *
 * Produces baseline sequential (SOF0) 8-bit, 3-component (or, on request, 1-component), single-scan files with
 * the T.81 Annex K example quantisation tables scaled by the IJG quality formula and the Annex K
 * typical Huffman tables, optional restart intervals (DRI + RST0..7 cycling).
 *
 * Content class "natural": per colour channel an 8x8-block-constant Gaussian field (sigma 40)
 * + per-pixel Gaussian noise (sigma 12) + a horizontal sinusoid (amplitude 30, period
 * 2*pi*23 px) around 128, clipped to [0,255]; RGB -> YCbCr by the JFIF matrix; chroma
 * subsampled by box averaging.  Class "smooth" drops the per-pixel noise, class "noise" is
 * uniform white noise.  The generator is xorshift64* seeded with `seed`, Box-Muller for normals.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- Annex K tables (T.81 K.1, K.2 quantisation; K.3-K.6 Huffman) ---- */
static const uint8_t k_qlum[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
                                   14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
                                   18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
                                   49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const uint8_t k_qchr[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
                                   24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                                   99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                   99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
static const uint8_t k_dc_lum_bits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t k_dc_chr_bits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t k_dc_vals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t k_ac_lum_bits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
static const uint8_t k_ac_lum_vals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71,
    0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72,
    0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
static const uint8_t k_ac_chr_bits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
static const uint8_t k_ac_chr_vals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22,
    0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1,
    0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
    0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a,
    0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

static const uint8_t k_zz2nat[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                     41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                     30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

/* ---- RNG ---- */
typedef struct { uint64_t s; int have; double spare; } rng_t;
static uint64_t rng_u64(rng_t *r)
{
    r->s ^= r->s >> 12; r->s ^= r->s << 25; r->s ^= r->s >> 27;
    return r->s * 0x2545F4914F6CDD1DULL;
}
static double rng_unit(rng_t *r) { return ((rng_u64(r) >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
static double rng_normal(rng_t *r)
{
    if (r->have) { r->have = 0; return r->spare; }
    double u = rng_unit(r), v = rng_unit(r);
    double m = sqrt(-2.0 * log(u));
    r->spare = m * sin(6.283185307179586 * v);
    r->have = 1;
    return m * cos(6.283185307179586 * v);
}

/* ---- output buffer ---- */
typedef struct { uint8_t *p; size_t n, cap; uint32_t acc; int nacc; } out_t;
static void put(out_t *o, int b)
{
    if (o->n == o->cap) { o->cap = o->cap ? o->cap * 2 : 1 << 16; o->p = (uint8_t *)realloc(o->p, o->cap); }
    o->p[o->n++] = (uint8_t)b;
}
static void put16(out_t *o, int v) { put(o, v >> 8); put(o, v & 0xff); }
static void putbits(out_t *o, uint32_t code, int len)
{
    o->acc = (o->acc << len) | (code & ((1u << len) - 1));
    o->nacc += len;
    while (o->nacc >= 8) {
        int b = (o->acc >> (o->nacc - 8)) & 0xff;
        put(o, b);
        if (b == 0xff) put(o, 0);
        o->nacc -= 8;
    }
}
static void flushbits(out_t *o)
{
    if (o->nacc > 0) putbits(o, (1u << (8 - o->nacc)) - 1, 8 - o->nacc); /* pad with ones */
    o->acc = 0; o->nacc = 0;
}

typedef struct { uint16_t code[256]; uint8_t len[256]; } enc_t;
static void mk_enc(enc_t *e, const uint8_t *bits, const uint8_t *vals)
{
    memset(e, 0, sizeof *e);
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
        for (int i = 0; i < bits[l - 1]; i++) { e->code[vals[k]] = (uint16_t)code; e->len[vals[k]] = (uint8_t)l; k++; code++; }
        code <<= 1;
    }
}

static int nbits_of(int v) { int a = v < 0 ? -v : v, n = 0; while (a) { n++; a >>= 1; } return n; }

static void encode_block(out_t *o, const int *q /* zig-zag */, int *pred, const enc_t *dc, const enc_t *ac)
{
    int diff = q[0] - *pred;
    *pred = q[0];
    int s = nbits_of(diff);
    putbits(o, dc->code[s], dc->len[s]);
    if (s) putbits(o, (uint32_t)(diff < 0 ? diff + (1 << s) - 1 : diff), s);
    int run = 0;
    for (int i = 1; i < 64; i++) {
        if (q[i] == 0) { run++; continue; }
        while (run > 15) { putbits(o, ac->code[0xf0], ac->len[0xf0]); run -= 16; }
        s = nbits_of(q[i]);
        int sym = (run << 4) | s;
        putbits(o, ac->code[sym], ac->len[sym]);
        putbits(o, (uint32_t)(q[i] < 0 ? q[i] + (1 << s) - 1 : q[i]), s);
        run = 0;
    }
    if (run) putbits(o, ac->code[0], ac->len[0]);
}

static void fdct_quant(const float *px /* 8x8, level shifted */, const uint8_t *qt_nat, int *qzz)
{
    static float cs[8][8];
    static int init = 0;
    if (!init) {
        for (int k = 0; k < 8; k++)
            for (int n = 0; n < 8; n++) cs[k][n] = (float)(cos((2 * n + 1) * k * 3.14159265358979323846 / 16.0) * (k ? 0.5 : 0.35355339059327376));
        init = 1;
    }
    float t[64], f[64];
    for (int y = 0; y < 8; y++)
        for (int k = 0; k < 8; k++) {
            float s = 0;
            for (int n = 0; n < 8; n++) s += px[y * 8 + n] * cs[k][n];
            t[y * 8 + k] = s;
        }
    for (int k = 0; k < 8; k++)
        for (int x = 0; x < 8; x++) {
            float s = 0;
            for (int n = 0; n < 8; n++) s += t[n * 8 + x] * cs[k][n];
            f[k * 8 + x] = s;
        }
    for (int i = 0; i < 64; i++) {
        int nat = k_zz2nat[i];
        float v = f[nat] / (float)qt_nat[nat];
        int q = (int)(v < 0 ? v - 0.5f : v + 0.5f);
        if (q > 1023) q = 1023;
        if (q < -1023) q = -1023;
        qzz[i] = q;
    }
    if (qzz[0] > 1016) qzz[0] = 1016; /* keeps DC differences within 11 bits */
    if (qzz[0] < -1016) qzz[0] = -1016;
}

static void scale_qt(const uint8_t *base, int quality, uint8_t *out)
{
    if (quality < 1) quality = 1;
    if (quality > 100) quality = 100;
    int scale = quality < 50 ? 5000 / quality : 200 - quality * 2;
    for (int i = 0; i < 64; i++) {
        int v = (base[i] * scale + 50) / 100;
        if (v < 1) v = 1;
        if (v > 255) v = 255;
        out[i] = (uint8_t)v;
    }
}

/* Synthesise the planar YCbCr source image (full resolution, w x h each). */
static void make_content(int w, int h, int cls, uint64_t seed, float *Y, float *Cb, float *Cr)
{
    rng_t r = {seed * 0x9E3779B97F4A7C15ULL + 0xD1B54A32D192ED03ULL, 0, 0.0};
    for (int i = 0; i < 8; i++) rng_u64(&r);
    int bw = (w + 7) / 8, bh = (h + 7) / 8;
    float *base = (float *)malloc(sizeof(float) * (size_t)bw * bh * 3);
    for (size_t i = 0; i < (size_t)bw * bh * 3; i++) base[i] = (float)(40.0 * rng_normal(&r));
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float rgb[3];
            float sn = (float)(30.0 * sin(x / 23.0));
            for (int c = 0; c < 3; c++) {
                float v;
                if (cls == 2) v = (float)(rng_unit(&r) * 255.0);
                else {
                    v = base[((size_t)(y / 8) * bw + x / 8) * 3 + c] + sn + 128.0f;
                    if (cls == 0) v += (float)(12.0 * rng_normal(&r));
                }
                v = v < 0 ? 0 : (v > 255 ? 255 : v);
                rgb[c] = floorf(v);
            }
            size_t i = (size_t)y * w + x;
            Y[i] = 0.299f * rgb[0] + 0.587f * rgb[1] + 0.114f * rgb[2];
            Cb[i] = -0.168736f * rgb[0] - 0.331264f * rgb[1] + 0.5f * rgb[2] + 128.0f;
            Cr[i] = 0.5f * rgb[0] - 0.418688f * rgb[1] - 0.081312f * rgb[2] + 128.0f;
        }
    free(base);
}

static void fetch_block(const float *plane, int pw, int ph, int bx, int by, int sx, int sy, float *blk)
{
    /* 8x8 samples at subsampling (sx, sy): box average of sx*sy source pixels, edge replicated */
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) {
            float s = 0;
            for (int j = 0; j < sy; j++)
                for (int i = 0; i < sx; i++) {
                    int px = (bx * 8 + x) * sx + i, py = (by * 8 + y) * sy + j;
                    if (px >= pw) px = pw - 1;
                    if (py >= ph) py = ph - 1;
                    s += plane[(size_t)py * pw + px];
                }
            blk[y * 8 + x] = s / (float)(sx * sy) - 128.0f;
        }
}

/*
 * Encode one synthetic image.  hs, vs: luma sampling factors (1 or 2); chroma is 1x1.
 * hs = vs = 0: a one-component (grayscale) file that codes the luma plane of the same content.
 * content class: 0 natural, 1 smooth, 2 white noise.  restart_interval in MCUs (0 = none).
 * Table numbering (added to the content class): +0x100 the luma DC table is written as DC table 1 and the
 * chroma DC table as DC table 0; +0x200 the same for the AC tables; +0x400 the luma quantisation table is
 * table 1 and the chroma one table 0.  The frame and scan headers name the tables accordingly, so the
 * decoded image is the same; only a decoder that reads the selectors gets it right.
 * Returns a malloc'd buffer (free with zj_synth_free) and its size in *out_len.
 */
uint8_t *zj_synth_encode(int w, int h, int hs, int vs, int quality, int content_class, uint64_t seed,
                         int restart_interval, size_t *out_len)
{
    const int gray = (hs == 0 && vs == 0);
    const int dc_y = (content_class >> 8) & 1, ac_y = (content_class >> 9) & 1, q_y = (content_class >> 10) & 1;
    content_class &= 0xff;
    if (gray) hs = vs = 1;
    if (w < 1 || h < 1 || w > 65535 || h > 65535 || hs < 1 || hs > 2 || vs < 1 || vs > 2) return NULL;
    float *Y = (float *)malloc(sizeof(float) * (size_t)w * h);
    float *Cb = (float *)malloc(sizeof(float) * (size_t)w * h);
    float *Cr = (float *)malloc(sizeof(float) * (size_t)w * h);
    if (!Y || !Cb || !Cr) { free(Y); free(Cb); free(Cr); return NULL; }
    make_content(w, h, content_class, seed, Y, Cb, Cr);

    uint8_t ql[64], qc[64];
    scale_qt(k_qlum, quality, ql);
    scale_qt(k_qchr, quality, qc);
    enc_t dcl, dcc, acl, acc;
    mk_enc(&dcl, k_dc_lum_bits, k_dc_vals);
    mk_enc(&dcc, k_dc_chr_bits, k_dc_vals);
    mk_enc(&acl, k_ac_lum_bits, k_ac_lum_vals);
    mk_enc(&acc, k_ac_chr_bits, k_ac_chr_vals);

    out_t o;
    memset(&o, 0, sizeof o);
    put16(&o, 0xffd8);
    put16(&o, 0xffe0); put16(&o, 16); /* APP0 JFIF */
    put(&o, 'J'); put(&o, 'F'); put(&o, 'I'); put(&o, 'F'); put(&o, 0);
    put(&o, 1); put(&o, 1); put(&o, 0); put16(&o, 1); put16(&o, 1); put(&o, 0); put(&o, 0);
    put16(&o, 0xffdb); put16(&o, 2 + 65 * 2); /* one DQT segment, two tables, zig-zag order */
    put(&o, q_y); for (int i = 0; i < 64; i++) put(&o, ql[k_zz2nat[i]]);
    put(&o, 1 - q_y); for (int i = 0; i < 64; i++) put(&o, qc[k_zz2nat[i]]);
    if (gray) {
        put16(&o, 0xffc0); put16(&o, 11); put(&o, 8); put16(&o, h); put16(&o, w); put(&o, 1);
        put(&o, 1); put(&o, 0x11); put(&o, q_y);
    } else {
        put16(&o, 0xffc0); put16(&o, 17); put(&o, 8); put16(&o, h); put16(&o, w); put(&o, 3);
        put(&o, 1); put(&o, (hs << 4) | vs); put(&o, q_y);
        put(&o, 2); put(&o, 0x11); put(&o, 1 - q_y);
        put(&o, 3); put(&o, 0x11); put(&o, 1 - q_y);
    }
    const uint8_t *hb[4] = {k_dc_lum_bits, k_ac_lum_bits, k_dc_chr_bits, k_ac_chr_bits};
    const uint8_t *hv[4] = {k_dc_vals, k_ac_lum_vals, k_dc_vals, k_ac_chr_vals};
    const int hinfo[4] = {0x00 | dc_y, 0x10 | ac_y, 0x00 | (1 - dc_y), 0x10 | (1 - ac_y)};
    for (int t = 0; t < 4; t++) { /* four DHT segments */
        int n = 0;
        for (int i = 0; i < 16; i++) n += hb[t][i];
        put16(&o, 0xffc4); put16(&o, 2 + 17 + n); put(&o, hinfo[t]);
        for (int i = 0; i < 16; i++) put(&o, hb[t][i]);
        for (int i = 0; i < n; i++) put(&o, hv[t][i]);
    }
    if (restart_interval > 0) { put16(&o, 0xffdd); put16(&o, 4); put16(&o, restart_interval); }
    if (gray) {
        put16(&o, 0xffda); put16(&o, 8); put(&o, 1);
        put(&o, 1); put(&o, (dc_y << 4) | ac_y);
    } else {
        put16(&o, 0xffda); put16(&o, 12); put(&o, 3);
        const int sy = (dc_y << 4) | ac_y, sc = ((1 - dc_y) << 4) | (1 - ac_y);
        put(&o, 1); put(&o, sy); put(&o, 2); put(&o, sc); put(&o, 3); put(&o, sc);
    }
    put(&o, 0); put(&o, 63); put(&o, 0);

    int mcux = (w + 8 * hs - 1) / (8 * hs), mcuy = (h + 8 * vs - 1) / (8 * vs);
    int pred[3] = {0, 0, 0};
    int qz[64];
    float blk[64];
    long mcu_count = 0;
    int rst = 0;
    long total = (long)mcux * mcuy;
    for (int my = 0; my < mcuy; my++)
        for (int mx = 0; mx < mcux; mx++) {
            if (restart_interval > 0 && mcu_count > 0 && mcu_count % restart_interval == 0) {
                flushbits(&o);
                put(&o, 0xff); put(&o, 0xd0 + (rst & 7));
                rst++;
                pred[0] = pred[1] = pred[2] = 0;
            }
            for (int v = 0; v < vs; v++)
                for (int u = 0; u < hs; u++) {
                    fetch_block(Y, w, h, mx * hs + u, my * vs + v, 1, 1, blk);
                    fdct_quant(blk, ql, qz);
                    encode_block(&o, qz, &pred[0], &dcl, &acl);
                }
            if (!gray) {
                fetch_block(Cb, w, h, mx, my, hs, vs, blk);
                fdct_quant(blk, qc, qz);
                encode_block(&o, qz, &pred[1], &dcc, &acc);
                fetch_block(Cr, w, h, mx, my, hs, vs, blk);
                fdct_quant(blk, qc, qz);
                encode_block(&o, qz, &pred[2], &dcc, &acc);
            }
            mcu_count++;
        }
    (void)total;
    flushbits(&o);
    put16(&o, 0xffd9);
    free(Y); free(Cb); free(Cr);
    *out_len = o.n;
    return o.p;
}

void zj_synth_free(uint8_t *p) { free(p); }
