/*
 * TEST INFRASTRUCTURE ONLY -- a stand-in for libhavoc_mi355x.so on machines WITHOUT a GPU, so that the HOST LOGIC layered on
 * the C ABI (libhavoc_classic.so's precompute-and-serve layer, libhavoc_search.so's batch client) can be exercised by the
 * CPU test suite (-m "not gpu").  "Device memory" is host memory and every "kernel" is a loop over the CPU oracle
 * (oracle/havoc_oracle.c).  It is built by tests/test_search.py into tests/_build/ with the soname of the real library and
 * loaded, in a SUBPROCESS of the test only, before the host library under test.  The product never builds, links or loads
 * it; on a GPU box the same tests run against the real library (-m gpu).  Only the entry points those two host libraries
 * use are implemented; anything else is absent on purpose (an unresolved symbol is a test failure, not a silent CPU path).
 */
#include "../include/havoc_mi355x.h"
#include "../oracle/havoc_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct havoc_mi355x_ctx { int device; long launches; };

static long g_launches = 0;
long mock_launches(void) { return g_launches; }

/* HAVOC_MOCK_HISTOGRAM=<file>: the jobs of every call by entry point and block size, written when the context is destroyed -- how
 * profiles/measure_call_mix.py reads the call mix of the reference's own encoder (run over libhavoc_classic.so + this stand-in) */
enum { H_SAD, H_SAD4, H_SATD, H_PRED_UNI8, H_PRED_UNI4, H_PRED_BI8, H_PRED_BI4, H_SUBTRACT_BI, H_INTRA, H_TRANSFORM, H_INVERSE, H_SSD, H_QUANTIZE, H_RDOQ,
       H_UNI8_COPY, H_UNI8_H, H_UNI8_V, H_UNI8_HV, H_UNI4_COPY, H_UNI4_H, H_UNI4_V, H_UNI4_HV, H_TRANSFORM_DST, H_DEQUANT_NONZERO, H_COUNT };   /* sub-tallies: interpolation by phase class, DST-VII transforms */
static const char *const g_hname[H_COUNT] = {"sad", "sad4", "satd", "pred_uni8", "pred_uni4", "pred_bi8", "pred_bi4", "subtract_bi", "intra", "transform",
                                             "inverse_transform", "ssd", "quantize", "rdoq", "uni8_copy", "uni8_h", "uni8_v", "uni8_hv", "uni4_copy",
                                             "uni4_h", "uni4_v", "uni4_hv", "transform_dst", "dequant_nonzero"};
static long g_hist[H_COUNT][65][65];
static void tally(int fn, int w, int h) { if (w >= 0 && w <= 64 && h >= 0 && h <= 64) __sync_fetch_and_add(&g_hist[fn][w][h], 1); }   /* the encoder calls from several threads */
static void write_histogram(void)
{
    const char *path = getenv("HAVOC_MOCK_HISTOGRAM");
    if (!path) return;
    FILE *f = fopen(path, "w");
    if (!f) return;
    fprintf(f, "{");
    for (int fn = 0, first = 1; fn < H_COUNT; ++fn)
        for (int w = 0; w <= 64; ++w)
            for (int h = 0; h <= 64; ++h)
                if (g_hist[fn][w][h])
                {
                    fprintf(f, "%s\"%s %dx%d\": %ld", first ? "" : ", ", g_hname[fn], w, h, g_hist[fn][w][h]);
                    first = 0;
                }
    fprintf(f, "}\n");
    fclose(f);
}

const char *havoc_mi355x_last_error(void) { return "mock device"; }
const char *havoc_mi355x_version(void) { return "MOCK (tests only)"; }

int havoc_mi355x_create(havoc_mi355x_ctx **ctx, int device, void *stream)
{
    (void)stream;
    *ctx = (havoc_mi355x_ctx *)calloc(1, sizeof(**ctx));
    (*ctx)->device = device;
    return 0;
}
void havoc_mi355x_destroy(havoc_mi355x_ctx *ctx) { write_histogram(); free(ctx); }
int havoc_mi355x_sync(havoc_mi355x_ctx *ctx) { (void)ctx; return 0; }
int havoc_mi355x_sync_spin(havoc_mi355x_ctx *ctx) { (void)ctx; return 0; }
/* fork / join lanes: the stand-in runs every launch at once, in call order */
int havoc_mi355x_fork(havoc_mi355x_ctx *ctx, int nlanes) { (void)ctx; return nlanes >= 1 && nlanes <= 8 ? 0 : -1; }
int havoc_mi355x_lane(havoc_mi355x_ctx *ctx, int lane) { (void)ctx; return lane >= 0 && lane < 8 ? 0 : -1; }
int havoc_mi355x_join(havoc_mi355x_ctx *ctx) { (void)ctx; return 0; }
int havoc_mi355x_malloc(havoc_mi355x_ctx *ctx, void **p, size_t n) { (void)ctx; *p = calloc(1, n + 64); return *p ? 0 : -1; }
int havoc_mi355x_free(havoc_mi355x_ctx *ctx, void *p) { (void)ctx; free(p); return 0; }
int havoc_mi355x_h2d(havoc_mi355x_ctx *ctx, void *d, const void *h, size_t n) { (void)ctx; memcpy(d, h, n); return 0; }
int havoc_mi355x_d2h(havoc_mi355x_ctx *ctx, void *h, const void *d, size_t n) { (void)ctx; memcpy(h, d, n); return 0; }
int havoc_mi355x_h2d_async(havoc_mi355x_ctx *ctx, void *d, const void *h, size_t n) { (void)ctx; memcpy(d, h, n); return 0; }
int havoc_mi355x_d2h_async(havoc_mi355x_ctx *ctx, void *h, const void *d, size_t n) { (void)ctx; memcpy(h, d, n); return 0; }
int havoc_mi355x_host_alloc(havoc_mi355x_ctx *ctx, size_t n, void **h, void **d)
{
    (void)ctx;
    *h = calloc(1, n + 64);
    *d = *h;   /* the device view of pinned memory is the same address here */
    return *h ? 0 : -1;
}
int havoc_mi355x_host_free(havoc_mi355x_ctx *ctx, void *h) { (void)ctx; free(h); return 0; }

#define AT(base, off, S) ((const char *)(base) + (long)(off) * (S))

int havoc_mi355x_sad(havoc_mi355x_ctx *ctx, int S, const void *src, intptr_t ss, const void *ref, intptr_t rs, const havoc_mi355x_pair_job *j, int n, int32_t *out)
{
    (void)ctx; ++g_launches;
    for (int i = 0; i < n; ++i) tally(H_SAD, j[i].w, j[i].h);
    for (int i = 0; i < n; ++i) out[i] = oracle_sad(AT(src, j[i].a_off, S), ss, AT(ref, j[i].b_off, S), rs, j[i].w, j[i].h, S);
    return 0;
}

int havoc_mi355x_sad4(havoc_mi355x_ctx *ctx, int S, const void *src, intptr_t ss, const void *ref, intptr_t rs, const havoc_mi355x_sad4_job *j, int n, int32_t *out)
{
    (void)ctx; ++g_launches;
    for (int i = 0; i < n; ++i)
    {
        const void *r[4];
        int v[4];
        tally(H_SAD4, j[i].w, j[i].h);
        for (int k = 0; k < 4; ++k) r[k] = AT(ref, j[i].ref_off[k], S);
        oracle_sad4(AT(src, j[i].src_off, S), ss, r, rs, v, j[i].w, j[i].h, S);
        for (int k = 0; k < 4; ++k) out[4 * i + k] = v[k];
    }
    return 0;
}

int havoc_mi355x_sad_surface(havoc_mi355x_ctx *ctx, int S, int range, int max_w, int max_h, const void *src, intptr_t ss, const void *ref, intptr_t rs,
                             const havoc_mi355x_surface_job *j, int n, int32_t *out)
{
    (void)ctx; (void)max_w; (void)max_h; ++g_launches;
    const int side = 2 * range + 1;
    for (int i = 0; i < n; ++i)
        for (int dy = -range; dy <= range; ++dy)
            for (int dx = -range; dx <= range; ++dx)
                out[j[i].out_off + (dy + range) * side + dx + range] =
                    oracle_sad(AT(src, j[i].src_off, S), ss, AT(ref, j[i].ref_off + (long)dy * rs + dx, S), rs, j[i].w, j[i].h, S);
    return 0;
}

int havoc_mi355x_satd(havoc_mi355x_ctx *ctx, int S, int max_w, int max_h, const void *a, intptr_t sa, const void *b, intptr_t sb,
                      const havoc_mi355x_pair_job *j, int n, int32_t *out)
{
    (void)ctx; (void)max_w; (void)max_h; ++g_launches;
    for (int i = 0; i < n; ++i) tally(H_SATD, j[i].w, j[i].h);
    for (int i = 0; i < n; ++i) out[i] = oracle_pu_satd(AT(a, j[i].a_off, S), sa, AT(b, j[i].b_off, S), sb, j[i].w, j[i].h, S);
    return 0;
}

int havoc_mi355x_satd_multi(havoc_mi355x_ctx *ctx, int S, int max_w, int max_h, const void *a, intptr_t sa, const void *b, intptr_t sb,
                            const havoc_mi355x_satd_multi_job *j, int n, int32_t *out)
{
    (void)ctx; (void)max_w; (void)max_h; ++g_launches;
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 16; ++k)
            out[16 * i + k] = k < j[i].count ? oracle_pu_satd(AT(a, j[i].a_off, S), sa, AT(b, j[i].b_off[k], S), sb, j[i].w, j[i].h, S) : 0;
    return 0;
}

int havoc_mi355x_pred_uni(havoc_mi355x_ctx *ctx, int S, int taps, int bitDepth, int max_w, int max_h, void *dst, intptr_t sd, const void *ref, intptr_t sr,
                          const havoc_mi355x_pred_uni_job *j, int n)
{
    (void)ctx; (void)max_w; (void)max_h; ++g_launches;
    for (int i = 0; i < n; ++i) tally(taps == 8 ? H_PRED_UNI8 : H_PRED_UNI4, j[i].w, j[i].h);
    for (int i = 0; i < n; ++i) tally((taps == 8 ? H_UNI8_COPY : H_UNI4_COPY) + (j[i].xFrac != 0) + 2 * (j[i].yFrac != 0), j[i].w, j[i].h);
    for (int i = 0; i < n; ++i)
        oracle_pred_uni((char *)dst + (long)j[i].dst_off * S, sd, AT(ref, j[i].ref_off, S), sr, j[i].w, j[i].h, j[i].xFrac, j[i].yFrac, bitDepth, taps, S);
    return 0;
}

int havoc_mi355x_subtract_bi(havoc_mi355x_ctx *ctx, int S, int bitDepth, void *dst, intptr_t sd, const void *pred, intptr_t sp, const void *src, intptr_t ss,
                             const havoc_mi355x_subtract_bi_job *j, int n)
{
    (void)ctx; ++g_launches;
    for (int i = 0; i < n; ++i) tally(H_SUBTRACT_BI, j[i].w, j[i].h);
    for (int i = 0; i < n; ++i)
        oracle_subtract_bi((char *)dst + (long)j[i].dst_off * S, sd, AT(pred, j[i].pred_off, S), sp, AT(src, j[i].src_off, S), ss, j[i].w, j[i].h, bitDepth, S);
    return 0;
}

/* plane[4*yFrac + xFrac][y][x] = HavocPredUni sample at (x, y): by 64 x 64 blocks, as tests/suite.py's LoopImpl does */
int havoc_mi355x_interp_planes(havoc_mi355x_ctx *ctx, int S, int bitDepth, void *planes, intptr_t plane_elems, const void *ref, intptr_t stride, int x0, int y0,
                               int width, int height)
{
    (void)ctx; ++g_launches;
    for (int yf = 0; yf < 4; ++yf)
        for (int xf = 0; xf < 4; ++xf)
        {
            if (!xf && !yf) continue;
            char *pl = (char *)planes + (long)(4 * yf + xf) * plane_elems * S;
            for (int by = y0; by < y0 + height; by += 64)
                for (int bx = x0; bx < x0 + width; bx += 64)
                {
                    const int w = x0 + width - bx < 64 ? x0 + width - bx : 64, h = y0 + height - by < 64 ? y0 + height - by : 64;
                    oracle_pred_uni(pl + ((long)by * stride + bx) * S, stride, AT(ref, (long)by * stride + bx, S), stride, w, h, xf, yf, bitDepth, 8, S);
                }
        }
    return 0;
}


/* 35-mode intra SATD stage: prediction then Hadamard tiles, per mode (what the fused kernel computes) */
int havoc_mi355x_intra_satd35(havoc_mi355x_ctx *ctx, int S, int bitDepth, int log2, const void *src, intptr_t ss, const void *nb,
                              const havoc_mi355x_intra_search_job *j, int n, int32_t *cost)
{
    (void)ctx; ++g_launches;
    const int N = 1 << log2, ts = log2 == 2 ? 4 : 8;
    uint16_t pred16[32 * 32];
    for (int i = 0; i < n; ++i)
    {
        const unsigned long long mask = (unsigned long long)j[i].filt_lo | ((unsigned long long)j[i].filt_hi << 32);
        for (int mode = 0; mode < 35; ++mode)
        {
            const long no = ((mask >> mode) & 1) ? j[i].nbf_off : j[i].nb_off;
            oracle_intra(pred16, 32, AT(nb, no, S), log2, mode, (j[i].edge && log2 < 5) ? 1 : 0, bitDepth, S);
            int c = 0;
            for (int y = 0; y < N; y += ts)
                for (int x = 0; x < N; x += ts)
                    c += oracle_satd(AT(src, j[i].src_off + (long)y * ss + x, S), ss, (const char *)pred16 + ((long)y * 32 + x) * S, 32, ts, S);
            cost[35 * i + mode] = c;
        }
    }
    return 0;
}

int havoc_mi355x_intra(havoc_mi355x_ctx *ctx, int S, int bitDepth, int log2, void *dst, intptr_t sd, const void *nb, const havoc_mi355x_intra_job *j, int n)
{
    (void)ctx; ++g_launches;
    for (int i = 0; i < n; ++i) tally(H_INTRA, 1 << log2, 1 << log2);
    for (int i = 0; i < n; ++i)
        oracle_intra((char *)dst + (long)j[i].dst_off * S, sd, AT(nb, j[i].nb_off, S), log2, j[i].mode, (j[i].edge && log2 < 5) ? 1 : 0, bitDepth, S);
    return 0;
}

/* ---- the remaining per-call entry points of libhavoc_classic.so's one-job path: reached when the reference's own encoder is run through
 * the table API on the stand-in device (tests/test_reference_encoder.py) ---- */
int havoc_mi355x_ssd(havoc_mi355x_ctx *ctx, int S, const void *a, intptr_t sa, const void *b, intptr_t sb, const havoc_mi355x_pair_job *j, int n, uint32_t *out)
{
    (void)ctx; ++g_launches;
    for (int i = 0; i < n; ++i) tally(H_SSD, j[i].w, j[i].h);
    for (int i = 0; i < n; ++i) out[i] = oracle_ssd(AT(a, j[i].a_off, S), sa, AT(b, j[i].b_off, S), sb, j[i].w, j[i].h, S);
    return 0;
}

int havoc_mi355x_ssd_linear(havoc_mi355x_ctx *ctx, const uint8_t *a, const uint8_t *b, int size, int32_t *out)
{
    (void)ctx; ++g_launches;
    *out = oracle_ssd_linear(a, b, size);
    return 0;
}

int havoc_mi355x_pred_bi(havoc_mi355x_ctx *ctx, int S, int taps, int bitDepth, int max_w, int max_h, void *dst, intptr_t sd, const void *ref, intptr_t sr,
                         const havoc_mi355x_pred_bi_job *j, int n)
{
    (void)ctx; (void)max_w; (void)max_h; ++g_launches;
    for (int i = 0; i < n; ++i) tally(taps == 8 ? H_PRED_BI8 : H_PRED_BI4, j[i].w, j[i].h);
    for (int i = 0; i < n; ++i)
        oracle_pred_bi((char *)dst + (long)j[i].dst_off * S, sd, AT(ref, j[i].ref0_off, S), AT(ref, j[i].ref1_off, S), sr, j[i].w, j[i].h, j[i].xFrac0,
                       j[i].yFrac0, j[i].xFrac1, j[i].yFrac1, bitDepth, taps, S);
    return 0;
}

int havoc_mi355x_transform(havoc_mi355x_ctx *ctx, int bitDepth, int trType, int log2, int16_t *coeffs, const int16_t *res, intptr_t stride_res,
                           const havoc_mi355x_tu_job *j, int n)
{
    (void)ctx; ++g_launches;
    for (int i = 0; i < n; ++i) tally(H_TRANSFORM, 1 << log2, 1 << log2);
    if (trType) for (int i = 0; i < n; ++i) tally(H_TRANSFORM_DST, 1 << log2, 1 << log2);
    for (int i = 0; i < n; ++i) oracle_transform(coeffs + j[i].coef_off, res + j[i].res_off, stride_res, log2, trType, bitDepth);
    return 0;
}

int havoc_mi355x_inverse_transform(havoc_mi355x_ctx *ctx, int bitDepth, int trType, int log2, int16_t *res, const int16_t *coeffs, const havoc_mi355x_tu_job *j, int n)
{
    (void)ctx; ++g_launches;
    for (int i = 0; i < n; ++i) oracle_inverse_transform(res + j[i].res_off, coeffs + j[i].coef_off, log2, trType, bitDepth);
    return 0;
}

int havoc_mi355x_inverse_transform_add(havoc_mi355x_ctx *ctx, int S, int bitDepth, int trType, int log2, void *dst, intptr_t sd, const void *pred, intptr_t sp,
                                       const int16_t *coeffs, const havoc_mi355x_tu_job *j, int n)
{
    (void)ctx; ++g_launches;
    for (int i = 0; i < n; ++i) tally(H_INVERSE, 1 << log2, 1 << log2);
    for (int i = 0; i < n; ++i)
        oracle_inverse_transform_add((char *)dst + (long)j[i].dst_off * S, sd, AT(pred, j[i].pred_off, S), sp, coeffs + j[i].coef_off, log2, trType, bitDepth, S);
    return 0;
}

int havoc_mi355x_quantize(havoc_mi355x_ctx *ctx, int16_t *dst, const int16_t *src, const havoc_mi355x_quant_job *j, int n, int32_t *cbf)
{
    (void)ctx; ++g_launches;
    for (int i = 0; i < n; ++i) tally(H_QUANTIZE, j[i].n > 64 ? 64 : j[i].n, 0);
    for (int i = 0; i < n; ++i) cbf[i] = oracle_quantize(dst + j[i].dst_off, src + j[i].src_off, j[i].scale, j[i].shift, j[i].offset, j[i].n);
    return 0;
}

int havoc_mi355x_quantize_inverse(havoc_mi355x_ctx *ctx, int16_t *dst, const int16_t *src, const havoc_mi355x_quant_job *j, int n)
{
    (void)ctx; ++g_launches;
    for (int i = 0; i < n; ++i)
    {   /* how dense the quantised levels of the reference's own encode are: "dequant_nonzero <side>x<class>", class 0 = no level, k = 2^(k-1) .. 2^k - 1 levels */
        int nz = 0, side = 4, cls = 0;
        for (int k = 0; k < j[i].n; ++k) nz += src[j[i].src_off + k] != 0;
        while (side * side < j[i].n) side *= 2;
        while ((1 << cls) <= nz) ++cls;
        tally(H_DEQUANT_NONZERO, side, cls);
    }
    for (int i = 0; i < n; ++i) oracle_quantize_inverse(dst + j[i].dst_off, src + j[i].src_off, j[i].scale, j[i].shift, j[i].n);
    return 0;
}

int havoc_mi355x_quantize_reconstruct(havoc_mi355x_ctx *ctx, int log2, uint8_t *rec, intptr_t sr, const uint8_t *pred, intptr_t sp, const int16_t *res,
                                      const havoc_mi355x_tu_job *j, int n)
{
    (void)ctx; ++g_launches;
    for (int i = 0; i < n; ++i) oracle_quantize_reconstruct(rec + j[i].dst_off, sr, pred + j[i].pred_off, sp, res + j[i].res_off, 1 << log2);
    return 0;
}

int havoc_mi355x_level_stats(havoc_mi355x_ctx *ctx, const int16_t *levels, const int32_t *j, int n, int32_t *out)
{
    (void)ctx; ++g_launches;
    for (int i = 0; i < n; ++i)
    {
        int nz = 0, sum = 0;
        for (int k = 0; k < j[2 * i + 1]; ++k)
        {
            const int v = levels[j[2 * i] + k];
            nz += v != 0;
            sum += v < 0 ? -v : v;
        }
        out[2 * i] = nz;
        out[2 * i + 1] = sum;
    }
    return 0;
}

/* ---- the TU chain either side of RDOQ and RDOQ itself (libhavoc_search.so: havoc_search_rqt) ---- */
int havoc_mi355x_tu_forward(havoc_mi355x_ctx *ctx, int S, int bitDepth, int trType, int log2, int16_t *coeffs, const void *src, intptr_t ss, const void *pred,
                            intptr_t sp, const havoc_mi355x_tu_fused_job *j, int n)
{
    (void)ctx; ++g_launches;
    const int N = 1 << log2;
    int16_t res[32 * 32];
    for (int i = 0; i < n; ++i)
    {
        oracle_residual(res, N, AT(src, j[i].src_off, S), ss, AT(pred, j[i].pred_off, S), sp, N, N, S);
        oracle_transform(coeffs + j[i].coef_off, res, N, log2, trType, bitDepth);
    }
    return 0;
}

int havoc_mi355x_tu_reconstruct(havoc_mi355x_ctx *ctx, int S, int bitDepth, int trType, int log2, int scale, int shift, void *rec, intptr_t sr, const void *pred,
                                intptr_t sp, const void *src, intptr_t ss, const int16_t *levels, const havoc_mi355x_tu_fused_job *j, int n, uint32_t *ssd)
{
    (void)ctx; ++g_launches;
    const int N = 1 << log2;
    int16_t deq[32 * 32];
    for (int i = 0; i < n; ++i)
    {
        oracle_quantize_inverse(deq, levels + j[i].coef_off, scale, shift, N * N);
        oracle_inverse_transform_add((char *)rec + (long)j[i].rec_off * S, sr, AT(pred, j[i].pred_off, S), sp, deq, log2, trType, bitDepth, S);
        ssd[i] = oracle_ssd(AT(src, j[i].src_off, S), ss, AT(rec, j[i].rec_off, S), sr, N, N, S);
    }
    return 0;
}

/* the 35-mode stage of one partition in one call (csrc/kernels_tu_fused.hip: k_intra_measure): the three entry points it stands for, one after the other */
int havoc_mi355x_intra_measure(havoc_mi355x_ctx *ctx, int S, int bitDepth, int log2, int16_t *coeffs, int16_t *coeffs_dct, int32_t *satd, void *rec0, uint32_t *ssd0,
                               const void *src, intptr_t ss, const void *pred, intptr_t sp, const havoc_mi355x_tu_fused_job *j, int n, int with_satd)
{
    (void)ctx; ++g_launches;
    const int N = 1 << log2, TS = N >= 8 ? 8 : 4, TX = N / TS;
    int16_t res[32 * 32], zero[32 * 32];
    memset(zero, 0, sizeof(zero));
    for (int i = 0; i < n; ++i)
    {
        const char *a = AT(src, j[i].src_off, S), *b = AT(pred, j[i].pred_off, S);
        if (with_satd)
            for (int t = 0; t < TX * TX; ++t)
                satd[i * TX * TX + t] = oracle_satd(a + ((long)(t / TX) * TS * ss + (t % TX) * TS) * S, ss, b + ((long)(t / TX) * TS * sp + (t % TX) * TS) * S, sp, TS, S);
        oracle_residual(res, N, a, ss, b, sp, N, N, S);
        oracle_transform(coeffs + j[i].coef_off, res, N, log2, log2 == 2 ? 1 : 0, bitDepth);
        if (log2 == 2) oracle_transform(coeffs_dct + j[i].coef_off, res, N, log2, 0, bitDepth);
        oracle_inverse_transform_add((char *)rec0 + (long)j[i].rec_off * S, N, b, sp, zero, log2, log2 == 2 ? 1 : 0, bitDepth, S);
        ssd0[i] = oracle_ssd(a, ss, AT(rec0, j[i].rec_off, S), N, N, N, S);
    }
    return 0;
}

/* the device-side intra decisions (csrc/kernels_decide.hip), one partition after the other: the same rules as search/decision.hpp: intraModeOrder and
 * search/tu_decision.hpp: decideIntraRd, which tests/test_search.py compares them with */
int havoc_mi355x_intra_order(havoc_mi355x_ctx *ctx, const int32_t *satd35, const havoc_mi355x_intra_mpm *mpm, int n, int32_t lambda_q16, int32_t *order, int32_t *count,
                             int32_t *slot, int32_t *total)
{
    (void)ctx; ++g_launches;
    total[0] = total[1] = 0;
    for (int i = 0; i < n; ++i)
    {
        const havoc_mi355x_intra_mpm *c = &mpm[i];
        int64_t costs[35];
        for (int m = 0; m < 35; ++m) costs[m] = 0;
        costs[c->cand_mode_list[0]] = c->rate_a_minus_c;
        costs[c->cand_mode_list[1]] = c->rate_b_minus_c;
        costs[c->cand_mode_list[2]] = c->rate_b_minus_c;
        for (int m = 0; m < 35; ++m) costs[m] += (int64_t)lambda_q16 * satd35[35 * i + m];
        int cnt = 0, forced = 0;
        for (int j = 0; j < c->max_refine + forced; ++j)
        {
            if (cnt == HAVOC_MI355X_INTRA_MAX_ORDER) { total[1] = 1; break; }
            int mode = 0;
            for (int m = 1; m < 35; ++m) if (costs[m] < costs[mode]) mode = m;
            costs[mode] = INT64_MAX;
            if (j == c->max_refine - 1)
                for (int k = 0; k < c->neighbour_modes; ++k)
                    if (costs[c->cand_mode_list[k]] != INT64_MAX) { costs[c->cand_mode_list[k]] = 0; ++forced; }
            order[HAVOC_MI355X_INTRA_MAX_ORDER * i + cnt++] = mode;
        }
        count[i] = cnt;
        slot[i] = total[0];
        total[0] += cnt;
    }
    return 0;
}

int havoc_mi355x_intra_expand(havoc_mi355x_ctx *ctx, const havoc_mi355x_intra_search_job *parts, const int32_t *order, const int32_t *count, const int32_t *slot,
                              const int32_t *ctx_index, int n, int log2, int quant_scale, int quant_shift, int inv_scale, int lambda_q16, int sdh_factor, int sdh,
                              havoc_mi355x_intra_job *ij, havoc_mi355x_tu_fused_job *tj, havoc_mi355x_rdoq_job *rj, int32_t *sj, int32_t *owner)
{
    (void)ctx; ++g_launches;
    const int area = 1 << (2 * log2);
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < count[i]; ++k)
        {
            const int c = slot[i] + k, mode = order[HAVOC_MI355X_INTRA_MAX_ORDER * i + k];
            const uint64_t mask = (uint64_t)parts[i].filt_lo | ((uint64_t)parts[i].filt_hi << 32);
            memset(&ij[c], 0, sizeof(ij[c]));
            ij[c].dst_off = c * area;
            ij[c].nb_off = ((mask >> mode) & 1) ? parts[i].nbf_off : parts[i].nb_off;
            ij[c].log2 = log2;
            ij[c].mode = mode;
            ij[c].edge = parts[i].edge;
            tj[c].coef_off = tj[c].pred_off = tj[c].rec_off = c * area;
            tj[c].src_off = parts[i].src_off;
            memset(&rj[c], 0, sizeof(rj[c]));
            rj[c].dst_off = rj[c].src_off = c * area;
            rj[c].quant_scale = quant_scale;
            rj[c].quant_shift = quant_shift;
            rj[c].inv_scale = inv_scale;
            rj[c].lambda_q16 = lambda_q16;
            rj[c].sdh_factor = sdh_factor;
            rj[c].ctx_index = ctx_index[i];
            rj[c].scan_idx = (uint8_t)((log2 == 2 || log2 == 3) ? ((mode >= 6 && mode <= 14) ? 2 : ((mode >= 22 && mode <= 30) ? 1 : 0)) : 0);
            rj[c].is_intra = 1;
            rj[c].sdh = (uint8_t)(sdh != 0);
            sj[2 * c] = c * area;
            sj[2 * c + 1] = area;
            owner[c] = i;
        }
    return 0;
}

int havoc_mi355x_intra_decide(havoc_mi355x_ctx *ctx, const havoc_mi355x_intra_mpm *mpm, const int32_t *order, const int32_t *count, const int32_t *slot, const int32_t *cbf,
                              const uint32_t *ssd, const int32_t *stats, const havoc_mi355x_tu_fused_job *tj, int n, int log2, int32_t reciprocal_lambda_q16,
                              havoc_mi355x_intra_choice *out, havoc_mi355x_tu_fused_job *fin)
{
    (void)ctx; ++g_launches;
    for (int i = 0; i < n; ++i)
    {
        const havoc_mi355x_intra_mpm *c = &mpm[i];
        havoc_mi355x_intra_choice r;
        memset(&r, 0, sizeof(r));
        r.mode = -1;
        r.cost = INT64_MAX;
        for (int j = 0; j < count[i]; ++j)
        {
            const int s = slot[i] + j, mode = order[HAVOC_MI355X_INTRA_MAX_ORDER * i + j];
            const int64_t mode_rate = mode == c->cand_mode_list[0] ? c->rate_a_minus_c : ((mode == c->cand_mode_list[1] || mode == c->cand_mode_list[2]) ? c->rate_b_minus_c : 0);
            const int64_t cost = mode_rate + ((int64_t)(1 + (cbf[s] ? 2 * stats[2 * s] + stats[2 * s + 1] : 0)) << 16) + (int64_t)reciprocal_lambda_q16 * (int32_t)ssd[s];
            ++r.evaluated;
            if (cost < r.cost)
            {
                r.mode = mode; r.index = j; r.cost = cost; r.cbf = cbf[s]; r.ssd = ssd[s]; r.nonzero = stats[2 * s]; r.sum_abs = stats[2 * s + 1];
            }
        }
        out[i] = r;
        fin[i] = tj[slot[i] + (r.index > 0 ? r.index : 0)];
        fin[i].rec_off = i << (2 * log2);
    }
    return 0;
}

/* the device-resident search is a kernel: the mock has nothing to put behind it (CPU tests never call it) */
int havoc_mi355x_search_motion_uni(havoc_mi355x_ctx *ctx, int S, const havoc_mi355x_search_params *params, const void *src, int64_t so, intptr_t ss, const void *ref,
                                   int64_t ro, intptr_t rs, const void *phase, intptr_t pe, int64_t po, const void *pus, int n, void *out)
{
    (void)ctx; (void)S; (void)params; (void)src; (void)so; (void)ss; (void)ref; (void)ro; (void)rs; (void)phase; (void)pe; (void)po; (void)pus; (void)n; (void)out;
    return HAVOC_MI355X_EINVAL;      /* a kernel: nothing behind it in the mock */
}
int havoc_mi355x_search_motion_bi(havoc_mi355x_ctx *ctx, int S, const havoc_mi355x_search_params *params, const void *src, int64_t so, intptr_t ss, const void *ref,
                                  int64_t ro, intptr_t rs, const void *phase, intptr_t pe, int64_t po, const void *phase_other, int64_t poo, const void *pus,
                                  const int16_t *start, int n, void *out)
{
    (void)ctx; (void)S; (void)params; (void)src; (void)so; (void)ss; (void)ref; (void)ro; (void)rs; (void)phase; (void)pe; (void)po; (void)phase_other; (void)poo;
    (void)pus; (void)start; (void)n; (void)out;
    return HAVOC_MI355X_EINVAL;      /* a kernel: nothing behind it in the mock */
}
/* the intra chain's device-side steps (round 4), restated on the host so that libhavoc_search's havoc_search_intra_chain runs over this stand-in (tests/intra_chain_runner.py):
 * reference samples of a partition from the running reconstruction with the substitution process of HEVC 8.4.4.2.2 (availability: the owner of the sample's 4x4 cell precedes
 * the partition in coding order), their [1 2 1]-filtered copy, candModeList from the neighbours' modes (turing/CandModeList.h:33-95); the champions back into the picture */
static int chain_sample(const void *rec, int S, long at) { return S == 1 ? ((const uint8_t *)rec)[at] : ((const uint16_t *)rec)[at]; }
static void chain_store(void *p, int S, long at, int v) { if (S == 1) ((uint8_t *)p)[at] = (uint8_t)v; else ((uint16_t *)p)[at] = (uint16_t)v; }
static int chain_available(const havoc_mi355x_intra_chain_layout *L, const int32_t *owner, int index, int x, int y)
{
    return x >= 0 && y >= 0 && x < L->pic_width && y < L->pic_height && owner[(y >> 2) * L->cells_per_row + (x >> 2)] < index;
}
int havoc_mi355x_intra_gather(havoc_mi355x_ctx *ctx, int S, const havoc_mi355x_intra_chain_layout *L, const void *rec, const int32_t *owner, const uint8_t *modes,
                              const havoc_mi355x_intra_chain_part *parts, int n, const havoc_mi355x_intra_search_job *jobs, void *nb, havoc_mi355x_intra_mpm *mpm)
{
    (void)ctx; ++g_launches;
    for (int i = 0; i < n; ++i)
    {
        const havoc_mi355x_intra_chain_part *p = &parts[i];
        const int nn = 1 << p->log2, len = 4 * nn + 1;
        int val[132], have[132], any = 0;
        for (int k = 0; k < len; ++k)
        {
            const int x = k <= 2 * nn ? p->x0 - 1 : p->x0 + k - 2 * nn - 1, y = k < 2 * nn ? p->y0 + 2 * nn - 1 - k : p->y0 - 1;
            have[k] = chain_available(L, owner, p->index, x, y);
            val[k] = have[k] ? chain_sample(rec, S, (long)(y + L->pad) * L->stride + x + L->pad) : 0;
            any |= have[k];
        }
        if (!any)
            for (int k = 0; k < len; ++k) val[k] = 1 << (L->bit_depth - 1);
        else
        {
            int first = 0;
            while (!have[first]) ++first;
            val[0] = val[first];
            for (int k = 1; k < len; ++k)
                if (!have[k]) val[k] = val[k - 1];
        }
        const long base = jobs[i].nb_off - (2 * nn + 1), basef = jobs[i].nbf_off - (2 * nn + 1);
        /* IntraReferenceSamples.h:382-402: flat edges of a 32x32 block -> bi-linear between the corner and the two ends */
        const int corner = val[2 * nn], thr = 1 << (L->bit_depth - 5);
        const int strong = L->strong_intra_smoothing == 1 && nn == 32 && abs(corner + val[4 * nn] - 2 * val[3 * nn]) < thr && abs(corner + val[0] - 2 * val[nn]) < thr;
        for (int k = 0; k < len; ++k)
        {
            chain_store(nb, S, base + k, val[k]);
            int f;
            if (strong) f = k < 64 ? (k * corner + (64 - k) * val[0] + 32) >> 6 : k == 64 ? corner : ((128 - k) * corner + (k - 64) * val[128] + 32) >> 6;
            else f = (k == 0 || k == len - 1) ? val[k] : (val[k - 1] + 2 * val[k] + val[k + 1] + 2) >> 2;
            chain_store(nb, S, basef + k, f);
        }
        const int a = chain_available(L, owner, p->index, p->x0 - 1, p->y0) ? modes[(p->y0 >> 2) * L->cells_per_row + ((p->x0 - 1) >> 2)] : 1;
        const int b = chain_available(L, owner, p->index, p->x0, p->y0 - 1) && (p->y0 - 1) >= ((p->y0 >> L->ctb_log2) << L->ctb_log2)
                          ? modes[((p->y0 - 1) >> 2) * L->cells_per_row + (p->x0 >> 2)] : 1;
        havoc_mi355x_intra_mpm *c = &mpm[i];
        if (a == b)
        {
            c->neighbour_modes = 1;
            if (a < 2) { c->cand_mode_list[0] = 0; c->cand_mode_list[1] = 1; c->cand_mode_list[2] = 26; }
            else { c->cand_mode_list[0] = a; c->cand_mode_list[1] = ((a + 29) % 32) + 2; c->cand_mode_list[2] = ((a - 1) % 32) + 2; }
        }
        else
        {
            c->neighbour_modes = 2;
            c->cand_mode_list[0] = a; c->cand_mode_list[1] = b;
            c->cand_mode_list[2] = (a != 0 && b != 0) ? 0 : ((a != 1 && b != 1) ? 1 : 26);
        }
    }
    return 0;
}
int havoc_mi355x_intra_commit(havoc_mi355x_ctx *ctx, int S, const havoc_mi355x_intra_chain_layout *L, void *rec, uint8_t *modes, const havoc_mi355x_intra_chain_part *parts,
                              int n, const void *blocks, const int32_t *mode, int mode_stride)
{
    (void)ctx; ++g_launches;
    for (int i = 0; i < n; ++i)
    {
        const havoc_mi355x_intra_chain_part *p = &parts[i];
        const int nn = 1 << p->log2;
        for (int y = 0; y < nn; ++y)
            for (int x = 0; x < nn; ++x)
                chain_store(rec, S, (long)(p->y0 + y + L->pad) * L->stride + p->x0 + x + L->pad, chain_sample(blocks, S, (long)i * nn * nn + y * nn + x));
        for (int y = 0; y < nn / 4; ++y)
            for (int x = 0; x < nn / 4; ++x) modes[((p->y0 >> 2) + y) * L->cells_per_row + (p->x0 >> 2) + x] = (uint8_t)mode[(long)i * mode_stride];
    }
    return 0;
}
int havoc_mi355x_intra_fill_spare(havoc_mi355x_ctx *ctx, const int32_t *total, int capacity, int log2, havoc_mi355x_intra_job *ij, havoc_mi355x_tu_fused_job *tj,
                                  havoc_mi355x_rdoq_job *rj, int32_t *sj, int32_t *owner)
{
    (void)ctx; ++g_launches;
    const int area = 1 << (2 * log2);
    if (total[0] <= 0) return 0;
    for (int c = total[0]; c < capacity; ++c)
    {
        ij[c] = ij[0];
        ij[c].dst_off = c * area;
        tj[c].coef_off = c * area; tj[c].src_off = tj[0].src_off; tj[c].pred_off = c * area; tj[c].rec_off = c * area;
        rj[c] = rj[0];
        rj[c].dst_off = rj[c].src_off = c * area;
        sj[2 * c] = c * area;
        sj[2 * c + 1] = area;
        owner[c] = owner[0];
    }
    return 0;
}
size_t havoc_mi355x_search_workspace(int width, int height) { (void)width; (void)height; return 256; }
int havoc_mi355x_search_picture_uni(havoc_mi355x_ctx *ctx, int S, const havoc_mi355x_search_params *params, const int64_t mvp_rate[2], const void *src, int64_t so,
                                    intptr_t ss, const void *ref, const int64_t ro[2], intptr_t rs, const void *phase, intptr_t pe, const int64_t po[2], const void *pus,
                                    const int32_t *first, int cx, int cy, int n, void *out, void *out_bi, int16_t *field, void *work, int steps)
{
    (void)steps; (void)n; (void)out_bi;
    (void)ctx; (void)S; (void)params; (void)mvp_rate; (void)src; (void)so; (void)ss; (void)ref; (void)ro; (void)rs; (void)phase; (void)pe; (void)po; (void)pus;
    (void)first; (void)cx; (void)cy; (void)out; (void)field; (void)work;
    return HAVOC_MI355X_EINVAL;
}

size_t havoc_mi355x_rdoq_workspace(int njobs) { (void)njobs; return 256; }
void havoc_mi355x_rdoq_lambda(double lambda, int inv_scale, int32_t *lq, int32_t *sf) { oracle_rdoq_lambda(lambda, inv_scale, lq, sf); }

int havoc_mi355x_rdoq(havoc_mi355x_ctx *ctx, int bitDepth, int log2, int16_t *dst, const int16_t *src, const uint8_t *states, const havoc_mi355x_rdoq_job *j, int n,
                      int32_t *cbf, void *work, size_t work_bytes)
{
    (void)ctx; (void)work; (void)work_bytes; ++g_launches;
    for (int i = 0; i < n; ++i)
        cbf[i] = oracle_rdoq(dst + j[i].dst_off, src + j[i].src_off, log2, j[i].c_idx, j[i].scan_idx, j[i].is_intra, j[i].sdh, j[i].quant_scale, j[i].quant_shift,
                             j[i].inv_scale, bitDepth, j[i].lambda_q16, j[i].sdh_factor, states + 128 * (long)j[i].ctx_index);
    return 0;
}

int havoc_mi355x_derive_bs(havoc_mi355x_ctx *ctx, const havoc_mi355x_cell *cells, intptr_t cs, int width, int height, int8_t *data, uint8_t *bs)
{
    (void)ctx; ++g_launches;
    oracle_derive_bs(cells, cs, width, height, data, bs);
    return 0;
}
