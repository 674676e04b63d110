/*
 * sangnom_vec.c -- a vectorisable CPU port of the SangNom2 opt=0 path for 8-bit Y clips.
 * TEST / BENCHMARK INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg and tests/test_oracle.py).
 *
 * Same arithmetic as sangnom_oracle.c (and therefore as /root/reference/src/SangNom2.cpp:74-257 with
 * T = uint8_t: wrap, not saturate), but written as straight loops over padded line buffers that gcc
 * -O3 -mavx2 turns into SIMD code: it stands in for the reference's opt=1 SSE2 path as the CPU timing
 * baseline (that path is not bit-identical to opt=0 and cannot be built here).
 * tests/test_oracle.py holds it bit-identical to the scalar oracle, pool contents included.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PAD 32 /* elements of edge replication kept on each side of a padded line */

typedef struct snv_ctx {
    int w, h, order, thr;
    int stride_e, bh;
    uint8_t* pool;      /* 9 x (bh + 1) x stride_e, zero-filled */
    uint8_t* c_pad;     /* padded kept lines c and n */
    uint8_t* n_pad;
    uint16_t* s_pad;    /* padded 3-row sums of stage 2 */
    uint8_t* fb[8];     /* f1, f2, b1, b2 of the pair (stage 1 / 3 share the code) */
} snv_ctx;

static void pad_line(uint8_t* dst, const uint8_t* src, int w)
{
    memcpy(dst + PAD, src, (size_t)w);
    memset(dst, src[0], PAD);
    memset(dst + PAD + w, src[w - 1], PAD);
}

/* calculateSangNom, SangNom2.cpp:60-65: T(IType(4*p1 + 5*p2 - p3) >> 3), arithmetic shift, wraps to uint8_t */
static inline uint8_t sg(int p1, int p2, int p3) { return (uint8_t)((int16_t)(4 * p1 + 5 * p2 - p3) >> 3); }
static inline uint8_t adiff(int a, int b) { return (uint8_t)(a > b ? a - b : b - a); }
static inline uint8_t avg(int a, int b) { return (uint8_t)((a + b + 1) >> 1); }

static void sangnom_values(const snv_ctx* x, const uint8_t* restrict c, const uint8_t* restrict n)
{
    uint8_t* restrict f1 = x->fb[0];
    uint8_t* restrict f2 = x->fb[1];
    uint8_t* restrict b1 = x->fb[2];
    uint8_t* restrict b2 = x->fb[3];
    const int w = x->w;
    for (int i = 0; i < w; ++i) {
        f1[i] = sg(c[i - 1], c[i], c[i + 1]);
        f2[i] = sg(n[i + 1], n[i], n[i - 1]);
        b1[i] = sg(c[i + 1], c[i], c[i - 1]);
        b2[i] = sg(n[i - 1], n[i], n[i + 1]);
    }
}

/* prepareBuffers_c, SangNom2.cpp:74-124 */
static void stage1(snv_ctx* x, const uint8_t* kept, int pitch2)
{
    const int w = x->w, nr = x->h / 2 - 1;
    const size_t bsz = (size_t)x->stride_e * (x->bh + 1);
    for (int y = 0; y < nr; ++y) {
        pad_line(x->c_pad, kept + (size_t)y * pitch2, w);
        pad_line(x->n_pad, kept + (size_t)(y + 1) * pitch2, w);
        const uint8_t* restrict c = x->c_pad + PAD;
        const uint8_t* restrict n = x->n_pad + PAD;
        sangnom_values(x, c, n);
        uint8_t* row = x->pool + (size_t)(y + 1) * x->stride_e;
        uint8_t* restrict r0 = row, * restrict r1 = row + bsz, * restrict r2 = row + 2 * bsz, * restrict r3 = row + 3 * bsz;
        uint8_t* restrict r4 = row + 4 * bsz, * restrict r5 = row + 5 * bsz, * restrict r6 = row + 6 * bsz;
        uint8_t* restrict r7 = row + 7 * bsz, * restrict r8 = row + 8 * bsz;
        const uint8_t* restrict f1 = x->fb[0], * restrict f2 = x->fb[1], * restrict b1 = x->fb[2], * restrict b2 = x->fb[3];
#pragma GCC ivdep
        for (int i = 0; i < w; ++i) {
            r0[i] = adiff(c[i - 3], n[i + 3]);
            r1[i] = adiff(c[i - 2], n[i + 2]);
            r2[i] = adiff(c[i - 1], n[i + 1]);
            r4[i] = adiff(c[i], n[i]);
            r6[i] = adiff(c[i + 1], n[i - 1]);
            r7[i] = adiff(c[i + 2], n[i - 2]);
            r8[i] = adiff(c[i + 3], n[i - 3]);
            r3[i] = adiff(f1[i], f2[i]);
            r5[i] = adiff(b1[i], b2[i]);
        }
    }
}

/* processBuffers_c, SangNom2.cpp:126-159: in place, rows 1..bh-1, over the whole pool stride */
static void stage2(snv_ctx* x)
{
    const int se = x->stride_e;
    for (int b = 0; b < 9; ++b) {
        uint8_t* buf = x->pool + (size_t)b * se * (x->bh + 1);
        for (int r = 1; r < x->bh; ++r) {
            const uint8_t* restrict pp = buf + (size_t)(r - 1) * se;
            uint8_t* restrict pc = buf + (size_t)r * se;
            const uint8_t* restrict pn = buf + (size_t)(r + 1) * se;
            uint16_t* restrict s = x->s_pad + PAD;
            for (int i = 0; i < se; ++i) s[i] = (uint16_t)(pp[i] + pc[i] + pn[i]);
            for (int k = 1; k <= 3; ++k) {
                s[-k] = s[0];
                s[se - 1 + k] = s[se - 1];
            }
            for (int i = 0; i < se; ++i)
                pc[i] = (uint8_t)((uint16_t)(s[i - 3] + s[i - 2] + s[i - 1] + s[i] + s[i + 1] + s[i + 2] + s[i + 3]) >> 4);
        }
    }
}

/* finalizePlane_c, SangNom2.cpp:161-257 */
static void stage3(snv_ctx* x, uint8_t* kept, int pitch2, uint8_t* out)
{
    const int w = x->w, nr = x->h / 2 - 1, thr = x->thr;
    const size_t bsz = (size_t)x->stride_e * (x->bh + 1);
    for (int y = 0; y < nr; ++y) {
        pad_line(x->c_pad, kept + (size_t)y * pitch2, w);
        pad_line(x->n_pad, kept + (size_t)(y + 1) * pitch2, w);
        const uint8_t* restrict c = x->c_pad + PAD;
        const uint8_t* restrict n = x->n_pad + PAD;
        sangnom_values(x, c, n);
        const uint8_t* v = x->pool + (size_t)(y + 1) * x->stride_e;
        const uint8_t* restrict p0 = v, * restrict p1 = v + bsz, * restrict p2 = v + 2 * bsz, * restrict p3 = v + 3 * bsz;
        const uint8_t* restrict p4 = v + 4 * bsz, * restrict p5 = v + 5 * bsz, * restrict p6 = v + 6 * bsz;
        const uint8_t* restrict p7 = v + 7 * bsz, * restrict p8 = v + 8 * bsz;
        const uint8_t* restrict f1 = x->fb[0], * restrict f2 = x->fb[1], * restrict b1 = x->fb[2], * restrict b2 = x->fb[3];
        uint8_t* restrict o = out + (size_t)y * pitch2;
#pragma GCC ivdep
        for (int i = 0; i < w; ++i) {
            const uint8_t v0 = p0[i], v1 = p1[i], v2 = p2[i], v3 = p3[i], v4 = p4[i];
            const uint8_t v5 = p5[i], v6 = p6[i], v7 = p7[i], v8 = p8[i];
            uint8_t m = v0 < v1 ? v0 : v1;
            m = v2 < m ? v2 : m;
            m = v3 < m ? v3 : m;
            m = v4 < m ? v4 : m;
            m = v5 < m ? v5 : m;
            m = v6 < m ? v6 : m;
            m = v7 < m ? v7 : m;
            m = v8 < m ? v8 : m;
            /* the ladder of SangNom2.cpp:211-249, lowest rung first so that the highest matching rung wins */
            const uint8_t a0 = avg(c[i - 3], n[i + 3]), a8 = avg(c[i + 3], n[i - 3]), a1 = avg(c[i - 2], n[i + 2]);
            const uint8_t a7 = avg(c[i + 2], n[i - 2]), a2 = avg(c[i - 1], n[i + 1]), a6 = avg(c[i + 1], n[i - 1]);
            const uint8_t a3 = avg(f1[i], f2[i]), a5 = avg(b1[i], b2[i]), a4 = avg(c[i], n[i]);
            uint8_t res = a0;
            res = v8 == m ? a8 : res;
            res = v1 == m ? a1 : res;
            res = v7 == m ? a7 : res;
            res = v2 == m ? a2 : res;
            res = v6 == m ? a6 : res;
            res = v3 == m ? a3 : res;
            res = v5 == m ? a5 : res;
            res = ((v4 == m) | (m > thr)) ? a4 : res;
            o[i] = res;
        }
    }
}

snv_ctx* snv_create_y8(int width, int height, int order, int aa)
{
    snv_ctx* x = (snv_ctx*)calloc(1, sizeof *x);
    if (!x) return NULL;
    x->w = width;
    x->h = height;
    x->order = order;
    x->thr = (int)(uint8_t)((float)aa * 21.0f / 16.0f); /* SangNom2.cpp:280-282, cast to T at :272 */
    x->stride_e = (width + 31) / 32 * 32;
    x->bh = (height + 1) >> 1;
    x->pool = (uint8_t*)calloc((size_t)9 * (x->bh + 1) * x->stride_e, 1);
    x->c_pad = (uint8_t*)malloc((size_t)width + 2 * PAD);
    x->n_pad = (uint8_t*)malloc((size_t)width + 2 * PAD);
    x->s_pad = (uint16_t*)malloc(((size_t)x->stride_e + 2 * PAD) * sizeof(uint16_t));
    for (int i = 0; i < 4; ++i) x->fb[i] = (uint8_t*)malloc((size_t)width + PAD);
    return x;
}

void snv_destroy(snv_ctx* x)
{
    if (!x) return;
    free(x->pool);
    free(x->c_pad);
    free(x->n_pad);
    free(x->s_pad);
    for (int i = 0; i < 4; ++i) free(x->fb[i]);
    free(x);
}

const uint8_t* snv_pool(const snv_ctx* x) { return x->pool; }

/* GetFrame, SangNom2.cpp:332-397 (one 8-bit plane, no dh) followed by the three stages */
void snv_process(snv_ctx* x, const uint8_t* src, int src_pitch, uint8_t* dst, int dst_pitch, int parity)
{
    const int off = x->order == 0 ? (parity ? 0 : 1) : x->order == 1 ? 0 : 1;
    for (int y = off; y < x->h; y += 2) memcpy(dst + (size_t)y * dst_pitch, src + (size_t)y * src_pitch, (size_t)x->w);
    if (off == 0) memcpy(dst + (size_t)(x->h - 1) * dst_pitch, dst + (size_t)(x->h - 2) * dst_pitch, (size_t)x->w);
    else memcpy(dst, dst + dst_pitch, (size_t)x->w);
    uint8_t* kept = dst + (size_t)off * dst_pitch;
    stage1(x, kept, 2 * dst_pitch);
    stage2(x);
    stage3(x, kept, 2 * dst_pitch, kept + dst_pitch);
}
