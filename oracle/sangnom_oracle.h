/*
 * sangnom_oracle.h -- CPU oracle for the SangNom2 hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference's opt=0 (C++) path:
 *   GetFrame frame assembly      /root/reference/src/SangNom2.cpp:332-397
 *   prepareBuffers_c  (stage 1)  /root/reference/src/SangNom2.cpp:74-124
 *   processBuffers_c  (stage 2)  /root/reference/src/SangNom2.cpp:126-159
 *   finalizePlane_c   (stage 3)  /root/reference/src/SangNom2.cpp:161-257
 *   pool geometry / thresholds   /root/reference/src/SangNom2.cpp:275-310
 *
 * PARITY UNPINNED: the reference holds no tests, fixtures or golden vectors, and it cannot be
 * built in this image (it needs the third-party AviSynth+ SDK header avisynth.h, which is not in
 * /root/reference and not installed; writing a stand-in for it is not allowed).  The oracle is
 * therefore a careful restatement checked only against (a) hand-derived known answers, (b) an
 * independent numpy restatement (oracle/sangnom_numpy.py) and (c) SURVEY.md Appendix A.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything in
 * oracle/.  The product (libsangnom_hip.so) never links, loads or calls it.
 *
 * Conventions that make the reference's behaviour *defined* (SURVEY.md section 8c):
 *   - the scratch pool is zero-filled at creation (the reference leaves it uninitialised,
 *     SangNom2.cpp:295-299,305-306);
 *   - one context == one filter instance; frames are processed in call order and the pool
 *     carries over between planes and frames exactly as in the reference.
 */
#ifndef SANGNOM_ORACLE_H
#define SANGNOM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sno_config {
    int width;   /* luma width of the input clip, pixels            */
    int height;  /* luma height of the INPUT clip (before dh x2)    */
    int bytes;   /* bytes per sample: 1, 2 or 4 (float)             */
    int bits;    /* bits per sample: 8..16, or 32 for float         */
    int planes;  /* 1 (Y) or 3 (YUV); a 4th plane is never touched  */
    int subw;    /* log2 horizontal chroma subsampling (0 or 1)     */
    int subh;    /* log2 vertical chroma subsampling (0 or 1)       */
    int order;   /* 0 = by parity, 1 = keep top, 2 = keep bottom    */
    int aa;      /* luma threshold 0..128                           */
    int aac;     /* chroma threshold 0..128                         */
    int dh;      /* double height                                   */
    int luma;    /* process luma                                    */
    int chroma;  /* process chroma                                  */
} sno_config;

typedef struct sno_ctx sno_ctx;

/* Mirrors Create_SangNom2's validation (SangNom2.cpp:407-422).  Returns 0 if the configuration
 * is accepted, otherwise non-zero with the reference's message text in msg. */
int sno_validate(const sno_config* cfg, char* msg, size_t msg_len);

sno_ctx* sno_create(const sno_config* cfg);
void sno_destroy(sno_ctx* ctx);

/* One GetFrame: src planes (input geometry) -> dst planes (output geometry).  Pitches in bytes.
 * parity is child->GetParity(n) and only matters for order == 0. */
int sno_process(sno_ctx* ctx, const void* const src[3], const int src_pitch[3],
                void* const dst[3], const int dst_pitch[3], int parity);

/* Geometry helpers shared with the tests. */
int sno_out_height(const sno_ctx* ctx);                 /* luma output height                    */
int sno_plane_width(const sno_ctx* ctx, int plane);     /* pixels                                */
int sno_plane_height_in(const sno_ctx* ctx, int plane); /* input rows                            */
int sno_plane_height_out(const sno_ctx* ctx, int plane);/* output rows                           */
int sno_pool_stride(const sno_ctx* ctx);                /* elements per pool row (stride_e)      */
int sno_pool_rows(const sno_ctx* ctx);                  /* bufferHeight + 1                      */
void* sno_pool(sno_ctx* ctx);                           /* 9 x rows x stride_e elements          */
double sno_threshold(const sno_ctx* ctx, int plane);    /* aaf[plane] after conversion to T      */

/* The three stages on one plane, exposed for unit tests.  dst points at the first line of the
 * output plane which already holds the kept field; stride in elements. */
void sno_plane(sno_ctx* ctx, void* dst, int dst_stride, int w, int h, int offset, int plane);

#ifdef __cplusplus
}
#endif
#endif
