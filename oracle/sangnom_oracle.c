/*
 * sangnom_oracle.c -- CPU oracle for the SangNom2 hot path.  TEST INFRASTRUCTURE ONLY.
 * See sangnom_oracle.h for scope, citations and the "PARITY UNPINNED" statement.
 *
 * Build (oracle/Makefile): gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC
 * -ffp-contract=off keeps the float path free of FMA contraction, matching the reference's
 * x86-64 baseline build (no -march, CMakeLists.txt:21-37).
 */
#include "sangnom_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct sno_ctx {
    sno_config cfg;
    int out_height;   /* vi.height after the dh doubling, SangNom2.cpp:284-285            */
    int stride_e;     /* bufferStride / sizeof(T), SangNom2.cpp:287                        */
    int bh;           /* bufferHeight = (vi.height + 1) >> 1, SangNom2.cpp:288             */
    float aaf[3];     /* SangNom2.cpp:280-282                                              */
    int process[3];   /* processPlane, SangNom2.cpp:276                                    */
    void* pool;       /* 9 * (bh + 1) * stride_e elements, zero-filled (convention)        */
    void* line;       /* bufferLine_, stride_e elements of IType, SangNom2.cpp:303         */
};

#define CAT2(a, b) a##b
#define CAT(a, b) CAT2(a, b)

#define T uint8_t
#define IT int16_t
#define SNO_FLOAT 0
#define FN(n) CAT(n, _u8)
#include "sangnom_oracle_impl.inc"
#undef T
#undef IT
#undef SNO_FLOAT
#undef FN

#define T uint16_t
#define IT int32_t
#define SNO_FLOAT 0
#define FN(n) CAT(n, _u16)
#include "sangnom_oracle_impl.inc"
#undef T
#undef IT
#undef SNO_FLOAT
#undef FN

#define T float
#define IT float
#define SNO_FLOAT 1
#define FN(n) CAT(n, _f32)
#include "sangnom_oracle_impl.inc"
#undef T
#undef IT
#undef SNO_FLOAT
#undef FN

/* Create_SangNom2's checks in the reference's order, SangNom2.cpp:407-422.  RGB / non-planar
 * clips cannot be expressed in sno_config (it only describes planar Y/YUV), and opt is not part
 * of this path, so those two checks have no counterpart here. */
int sno_validate(const sno_config* c, char* msg, size_t n)
{
    const char* e = NULL;
    const int is420 = c->planes >= 3 && c->subw == 1 && c->subh == 1;
    if (c->height % 2 != 0) e = "SangNom2: height must be even.";
    else if (is420 && c->height % 4) e = "SangNom2: height must be mod4.";
    else if (c->order < 0 || c->order > 2) e = "SangNom2: order must be between 0..2.";
    else if (c->aa < 0 || c->aa > 128) e = "SangNom2: aa must be between 0..128.";
    else if (c->aac < 0 || c->aac > 128) e = "SangNom2: aac must be between 0..128.";
    if (e) {
        if (msg && n) snprintf(msg, n, "%s", e);
        return 1;
    }
    if (msg && n) msg[0] = 0;
    return 0;
}

sno_ctx* sno_create(const sno_config* cfg)
{
    if (!cfg || cfg->width <= 0 || cfg->height <= 0) return NULL;
    if (cfg->bytes != 1 && cfg->bytes != 2 && cfg->bytes != 4) return NULL;
    if (cfg->planes != 1 && cfg->planes != 3) return NULL;
    sno_ctx* ctx = (sno_ctx*)calloc(1, sizeof(*ctx));
    if (!ctx) return NULL;
    ctx->cfg = *cfg;
    const int aa[3] = { cfg->aa, cfg->aac, cfg->aac };
    for (int i = 0; i < (cfg->planes < 3 ? cfg->planes : 3); ++i) {
        if (cfg->bytes < 4)
            ctx->aaf[i] = (aa[i] * 21.0f / 16.0f) * (float)(1 << (cfg->bits - 8));
        else
            ctx->aaf[i] = (aa[i] * 21.0f / 16.0f) / 256.0f;
    }
    ctx->process[0] = cfg->luma;
    ctx->process[1] = ctx->process[2] = cfg->chroma;
    ctx->out_height = cfg->dh ? cfg->height * 2 : cfg->height;
    ctx->stride_e = (cfg->width + 31) & ~31;
    ctx->bh = (ctx->out_height + 1) >> 1;
    const size_t pool_elems = (size_t)ctx->stride_e * (ctx->bh + 1) * 9;
    ctx->pool = calloc(pool_elems, (size_t)cfg->bytes);
    ctx->line = calloc((size_t)ctx->stride_e, cfg->bytes == 4 ? 4 : (size_t)cfg->bytes * 2);
    if (!ctx->pool || !ctx->line) {
        sno_destroy(ctx);
        return NULL;
    }
    return ctx;
}

void sno_destroy(sno_ctx* ctx)
{
    if (!ctx) return;
    free(ctx->pool);
    free(ctx->line);
    free(ctx);
}

int sno_out_height(const sno_ctx* c) { return c->out_height; }
int sno_plane_width(const sno_ctx* c, int p) { return p == 0 ? c->cfg.width : c->cfg.width >> c->cfg.subw; }
int sno_plane_height_in(const sno_ctx* c, int p) { return p == 0 ? c->cfg.height : c->cfg.height >> c->cfg.subh; }
int sno_plane_height_out(const sno_ctx* c, int p) { return p == 0 ? c->out_height : c->out_height >> c->cfg.subh; }
int sno_pool_stride(const sno_ctx* c) { return c->stride_e; }
int sno_pool_rows(const sno_ctx* c) { return c->bh + 1; }
void* sno_pool(sno_ctx* c) { return c->pool; }

double sno_threshold(const sno_ctx* c, int plane)
{
    switch (c->cfg.bytes) {
    case 1: return (double)(uint8_t)c->aaf[plane];
    case 2: return (double)(uint16_t)c->aaf[plane];
    default: return (double)c->aaf[plane];
    }
}

void sno_plane(sno_ctx* ctx, void* dst, int dst_stride, int w, int h, int offset, int plane)
{
    switch (ctx->cfg.bytes) {
    case 1: plane_u8(ctx, dst, dst_stride, w, h, offset, plane); break;
    case 2: plane_u16(ctx, dst, dst_stride, w, h, offset, plane); break;
    default: plane_f32(ctx, dst, dst_stride, w, h, offset, plane); break;
    }
}

/* env->BitBlt: plain pitched row copy. */
static void blit(uint8_t* d, ptrdiff_t dp, const uint8_t* s, ptrdiff_t sp, size_t row_bytes, int rows)
{
    for (int y = 0; y < rows; ++y)
        memcpy(d + dp * y, s + sp * y, row_bytes);
}

/* SangNom2::GetFrame, SangNom2.cpp:332-397. */
int sno_process(sno_ctx* ctx, const void* const src[3], const int src_pitch[3],
                void* const dst[3], const int dst_pitch[3], int parity)
{
    const sno_config* c = &ctx->cfg;
    int offset;
    switch (c->order) {
    case 0: offset = parity ? 0 : 1; break;
    case 1: offset = 0; break;
    default: offset = 1; break;
    }
    const int planecount = c->planes < 3 ? c->planes : 3;
    for (int i = 0; i < planecount; ++i) {
        const uint8_t* srcp = (const uint8_t*)src[i];
        uint8_t* dstp = (uint8_t*)dst[i];
        const ptrdiff_t sp = src_pitch[i], dp = dst_pitch[i];
        const int src_h = sno_plane_height_in(ctx, i);
        const int dst_h = sno_plane_height_out(ctx, i);
        const int w = sno_plane_width(ctx, i);
        const size_t row_bytes = (size_t)w * c->bytes;

        if (c->dh) {
            /* every source line is a kept line of the double-height output */
            blit(dstp + offset * dp, dp * 2, srcp, sp, row_bytes, src_h);
        } else {
            if (!ctx->process[i]) {
                /* reference: one memcpy of src_pitch*src_height bytes (assumes equal pitches,
                 * SangNom2.cpp:372); restated as a pitched row copy, identical when they match */
                blit(dstp, dp, srcp, sp, row_bytes, src_h);
                continue;
            }
            blit(dstp + offset * dp, dp * 2, srcp + offset * sp, sp * 2, row_bytes, src_h / 2);
        }
        /* the one line that cannot be interpolated */
        if (offset == 0)
            memcpy(dstp + (dst_h - 1) * dp, dstp + (dst_h - 2) * dp, row_bytes);
        else
            memcpy(dstp, dstp + dp, row_bytes);

        sno_plane(ctx, dstp, (int)(dp / c->bytes), w, dst_h, offset, i);
    }
    return 0;
}
