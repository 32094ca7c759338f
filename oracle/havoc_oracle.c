/*
 * TEST INFRASTRUCTURE ONLY -- see havoc_oracle.h.  Plain-C99 restatement of the havoc primitives, written
 * from the HEVC definitions and the behaviour of the reference C functions cited per function.  Scalar,
 * single-threaded, deliberately simple: it is the checker, never the thing shipped or optimised.
 * Parity status: pinned against oracle/_ref (the reference's own sources compiled here) and tests/golden/.
 */
#include "havoc_oracle.h"

#include <stdlib.h>
#include <string.h>

static inline int px(const void *p, intptr_t i, int S)
{
    return S == 1 ? (int)((const uint8_t *)p)[i] : (int)((const uint16_t *)p)[i];
}

static inline void put(void *p, intptr_t i, int v, int S)
{
    if (S == 1) ((uint8_t *)p)[i] = (uint8_t)v;
    else ((uint16_t *)p)[i] = (uint16_t)v;
}

static inline int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }

/* ------------------------------------------------------------------------------------------------ */
/* distortion metrics                                                                               */
/* ------------------------------------------------------------------------------------------------ */

/* havoc/sad.cpp:432-449 */
int oracle_sad(const void *src, intptr_t ss, const void *ref, intptr_t rs, int w, int h, int S)
{
    int sad = 0;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            sad += abs(px(src, x + y * ss, S) - px(ref, x + y * rs, S));
    return S == 2 ? sad >> 2 : sad;
}

/* havoc/sad.cpp:513-542 */
void oracle_sad4(const void *src, intptr_t ss, const void *const ref[4], intptr_t rs, int sad[4], int w, int h, int S)
{
    for (int k = 0; k < 4; ++k)
        sad[k] = oracle_sad(src, ss, ref[k], rs, w, h, S);
}

/* havoc/ssd.cpp:28-43: accumulates in uint32_t (wraps mod 2^32), 16-bit path >>4 */
uint32_t oracle_ssd(const void *a, intptr_t sa, const void *b, intptr_t sb, int w, int h, int S)
{
    uint32_t ssd = 0;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
        {
            const int d = px(a, x + y * sa, S) - px(b, x + y * sb, S);
            ssd += (uint32_t)(d * d);
        }
    return S == 2 ? ssd >> 4 : ssd;
}

/* havoc/diff.cpp:29-39 */
int oracle_ssd_linear(const uint8_t *a, const uint8_t *b, int n)
{
    int sum = 0;
    for (int i = 0; i < n; ++i)
    {
        const int d = (int)a[i] - (int)b[i];
        sum += d * d;
    }
    return sum;
}

/* in-place length-n Walsh-Hadamard butterfly network over a strided vector (unnormalised).  The output
 * ordering differs from the reference's recursion (havoc/hadamard.cpp:31-56) but SATD only sums |coeff|. */
static void wht(int *v, int n, int stride)
{
    for (int len = 1; len < n; len <<= 1)
        for (int i = 0; i < n; i += len << 1)
            for (int j = i; j < i + len; ++j)
            {
                const int a = v[j * stride], b = v[(j + len) * stride];
                v[j * stride] = a + b;
                v[(j + len) * stride] = a - b;
            }
}

/* havoc/hadamard.cpp:58-98 */
int oracle_satd(const void *a, intptr_t sa, const void *b, intptr_t sb, int n, int S)
{
    int m[8 * 8];
    for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x)
            m[y * n + x] = px(a, x + y * sa, S) - px(b, x + y * sb, S);
    for (int y = 0; y < n; ++y) wht(m + y * n, n, 1);
    for (int x = 0; x < n; ++x) wht(m + x, n, n);
    int sum = n / 4;
    for (int i = 0; i < n * n; ++i) sum += abs(m[i]);
    sum /= n / 2;
    return S == 2 ? sum >> 2 : sum;
}

/* turing/Measure.h:97-135 */
int oracle_pu_satd(const void *a, intptr_t sa, const void *b, intptr_t sb, int w, int h, int S)
{
    const int n = ((w | h) & 3) ? 2 : (((w | h) & 7) ? 4 : 8);
    int satd = 0;
    for (int y = 0; y < h; y += n)
        for (int x = 0; x < w; x += n)
            satd += oracle_satd((const char *)a + (x + y * sa) * S, sa, (const char *)b + (x + y * sb) * S, sb, n, S);
    return satd;
}

/* ------------------------------------------------------------------------------------------------ */
/* inter prediction                                                                                 */
/* ------------------------------------------------------------------------------------------------ */

/* HEVC interpolation filter taps, havoc/pred_inter.cpp:39-69 */
static const int lumaTaps[4][8] = {
    { 0, 0, 0, 64, 0, 0, 0, 0 },
    { -1, 4, -10, 58, 17, -5, 1, 0 },
    { -1, 4, -11, 40, 40, -11, 4, -1 },
    { 0, 1, -5, 17, 58, -10, 4, -1 },
};
static const int chromaTaps[8][4] = {
    { 0, 64, 0, 0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 },
    { -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 },
};

static inline int tap(int taps, int frac, int k) { return taps == 8 ? lumaTaps[frac][k] : chromaTaps[frac][k]; }

/* horizontal pass to an int plane of (h + taps - 1) rows, first row = taps/2-1 rows above the block:
 * havoc/pred_inter.cpp:146-163 (uni HV) and :1238-1248 (bi): t = sum(c*x) >> shift1, no rounding */
static void hpass(int *tmp, const void *ref, intptr_t rs, int w, int h, int frac, int taps, int shift1, int S)
{
    const int above = taps / 2 - 1;
    for (int y = 0; y < h + taps - 1; ++y)
        for (int x = 0; x < w; ++x)
        {
            int a = 0;
            for (int k = 0; k < taps; ++k)
                a += tap(taps, frac, k) * px(ref, (x + k - above) + (intptr_t)(y - above) * rs, S);
            tmp[y * 64 + x] = a >> shift1;
        }
}

/* havoc/pred_inter.cpp:113-202 */
void oracle_pred_uni(void *dst, intptr_t sd, const void *ref, intptr_t rs, int w, int h, int xFrac, int yFrac, int bitDepth, int taps, int S)
{
    const int max = (1 << bitDepth) - 1;
    const int above = taps / 2 - 1;
    if (!xFrac && !yFrac)
    {
        for (int y = 0; y < h; ++y)
            memcpy((char *)dst + y * sd * S, (const char *)ref + y * rs * S, (size_t)w * S);
        return;
    }
    if (!yFrac || !xFrac)
    {
        const intptr_t step = xFrac ? 1 : rs;
        const int frac = xFrac ? xFrac : yFrac;
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x)
            {
                int a = 32;
                for (int k = 0; k < taps; ++k)
                    a += tap(taps, frac, k) * px(ref, x + y * rs + (k - above) * step, S);
                put(dst, x + y * sd, clip3(0, max, a >> 6), S);
            }
        return;
    }
    int shift1 = bitDepth - 8;
    if (shift1 > 4) shift1 = 4;
    int shift3 = 14 - bitDepth;
    if (shift3 < 2) shift3 = 2;
    const int shift = 6 + shift3;
    int *tmp = (int *)malloc(sizeof(int) * 64 * (64 + 7));
    hpass(tmp, ref, rs, w, h, xFrac, taps, shift1, S);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
        {
            int a = 1 << (shift - 1);
            for (int k = 0; k < taps; ++k)
                a += tap(taps, yFrac, k) * tmp[(y + k) * 64 + x];
            put(dst, x + y * sd, clip3(0, max, a >> shift), S);
        }
    free(tmp);
}

/* havoc/pred_inter.cpp:1207-1252 */
void oracle_pred_bi(void *dst, intptr_t sd, const void *ref0, const void *ref1, intptr_t rs, int w, int h, int xFrac0, int yFrac0, int xFrac1, int yFrac1, int bitDepth, int taps, int S)
{
    const int max = (1 << bitDepth) - 1;
    int shift1 = bitDepth - 8;
    if (shift1 > 4) shift1 = 4;
    int shift3 = 14 - bitDepth;
    if (shift3 < 2) shift3 = 2;
    int *tmp = (int *)malloc(sizeof(int) * 64 * (64 + 7));
    int *inter = (int *)malloc(sizeof(int) * 2 * 64 * 64);
    const void *refs[2] = { ref0, ref1 };
    const int xf[2] = { xFrac0, xFrac1 }, yf[2] = { yFrac0, yFrac1 };
    for (int r = 0; r < 2; ++r)
    {
        hpass(tmp, refs[r], rs, w, h, xf[r], taps, shift1, S);
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x)
            {
                int a = 0;
                for (int k = 0; k < taps; ++k)
                    a += tap(taps, yf[r], k) * tmp[(y + k) * 64 + x];
                inter[r * 4096 + y * 64 + x] = a >> 6;
            }
    }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
        {
            const int v = (inter[y * 64 + x] + inter[4096 + y * 64 + x] + (1 << shift3)) >> (shift3 + 1);
            put(dst, x + y * sd, clip3(0, max, v), S);
        }
    free(inter);
    free(tmp);
}

/* havoc/pred_inter.cpp:2063-2080 */
void oracle_subtract_bi(void *dst, intptr_t sd, const void *pred, intptr_t sp, const void *src, intptr_t ss, int w, int h, int bitDepth, int S)
{
    const int max = (1 << bitDepth) - 1;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            put(dst, x + y * sd, clip3(0, max, 2 * px(src, x + y * ss, S) - px(pred, x + y * sp, S)), S);
}

/* ------------------------------------------------------------------------------------------------ */
/* intra prediction                                                                                 */
/* ------------------------------------------------------------------------------------------------ */

/* havoc/pred_intra.cpp:76-98 */
static const int intraPredAngle[35] = { 0, 0, 32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26, -32,
                                        -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32 };
static const int invAngle[26] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, -4096, -1638, -910, -630, -482, -390, -315, -256,
                                  -315, -390, -482, -630, -910, -1638, -4096 };

/* p(x, y) with the reference's neighbour layout, havoc/pred_intra.cpp:43-51 */
#define P(x, y) px(nb, (x) - (y) - 1, S)

/* havoc/pred_intra.cpp:20282-20401 */
void oracle_intra(void *dst, intptr_t sd, const void *nb, int log2, int mode, int edge, int bitDepth, int S)
{
    const int n = 1 << log2;
    const int max = (1 << bitDepth) - 1;
    if (mode == 0)
    {
        for (int y = 0; y < n; ++y)
            for (int x = 0; x < n; ++x)
                put(dst, x + y * sd, ((n - 1 - x) * P(-1, y) + (x + 1) * P(n, -1) + (n - 1 - y) * P(x, -1) + (y + 1) * P(-1, n) + n) >> (log2 + 1), S);
        return;
    }
    if (mode == 1)
    {
        int dc = n;
        for (int i = 0; i < n; ++i) dc += P(i, -1) + P(-1, i);
        dc >>= log2 + 1;
        for (int y = 0; y < n; ++y)
            for (int x = 0; x < n; ++x)
                put(dst, x + y * sd, dc, S);
        if (edge)
        {
            put(dst, 0, (P(-1, 0) + 2 * dc + P(0, -1) + 2) >> 2, S);
            for (int x = 1; x < n; ++x) put(dst, x, (P(x, -1) + 3 * dc + 2) >> 2, S);
            for (int y = 1; y < n; ++y) put(dst, y * sd, (P(-1, y) + 3 * dc + 2) >> 2, S);
        }
        return;
    }
    /* angular: build the 1-D reference array, index range [-n .. 2n] stored at offset 64 */
    int refbuf[64 + 2 * 32 + 1 + 64];
    int *ref = refbuf + 64;
    const int angle = intraPredAngle[mode];
    const int vertical = mode >= 18;
    for (int i = 0; i <= n; ++i) ref[i] = vertical ? P(-1 + i, -1) : P(-1, -1 + i);
    if (angle < 0)
    {
        const int last = (n * angle) >> 5;
        if (last < -1)
            for (int i = -1; i >= last; --i)
            {
                const int j = -1 + ((i * invAngle[mode] + 128) >> 8);
                ref[i] = vertical ? P(-1, j) : P(j, -1);
            }
    }
    else
        for (int i = n + 1; i <= 2 * n; ++i) ref[i] = vertical ? P(-1 + i, -1) : P(-1, -1 + i);

    for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x)
        {
            const int major = vertical ? y : x, minor = vertical ? x : y;
            const int idx = ((major + 1) * angle) >> 5;
            const int fact = ((major + 1) * angle) & 31;
            int v;
            if (fact == 0) v = ref[minor + idx + 1];
            else v = ((32 - fact) * ref[minor + idx + 1] + fact * ref[minor + idx + 2] + 16) >> 5;
            put(dst, x + y * sd, v, S);
        }
    if (edge && mode == 26)
        for (int y = 0; y < n; ++y)
            put(dst, y * sd, clip3(0, max, P(0, -1) + ((P(-1, y) - P(-1, -1)) >> 1)), S);
    if (edge && mode == 10)
        for (int x = 0; x < n; ++x)
            put(dst, x, clip3(0, max, P(-1, 0) + ((P(x, -1) - P(-1, -1)) >> 1)), S);
}
#undef P

/* ------------------------------------------------------------------------------------------------ */
/* transforms                                                                                       */
/* ------------------------------------------------------------------------------------------------ */

/* HEVC core transform basis.  Magnitudes by angle index j (units of pi/64); row k, column n of the 32-point
 * matrix is cos-like in k*(2n+1).  Values equal the tables at havoc/transform.cpp:85-91,119-129,170-188,
 * 243-277 (checked against the reference build by tests/test_oracle_vs_reference.py). */
static const int mag[33] = { 64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                             61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0 };

static int basis(int n, int k, int c) /* coefficient (k, c) of the n-point DCT matrix */
{
    const int kk = k * (32 / n);
    if (kk == 0) return 64;
    const int j = (kk * (2 * c + 1)) & 127;
    if (j <= 32) return mag[j];
    if (j <= 64) return -mag[64 - j];
    if (j <= 96) return -mag[j - 64];
    return mag[128 - j];
}

/* DST-VII 4x4, havoc/transform.cpp:59-67 (factorised there) */
static const int dst7[4][4] = { { 29, 55, 74, 84 }, { 74, 74, 0, -74 }, { 84, -29, -74, 55 }, { 55, -84, 74, -29 } };

static inline int coef(int n, int trType, int k, int c) { return trType ? dst7[k][c] : basis(n, k, c); }

/* forward: out[k][i] = (short)((sum_j M[k][j] * in[i][j] + add) >> shift)  -- WRAPS to int16
 * (havoc/transform.cpp:3071-3084 shiftRight truncates through `short` before its clamp) */
static void fwd_pass(int16_t *out, const int16_t *in, intptr_t stride, int n, int trType, int shift)
{
    const int add = 1 << (shift - 1);
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < n; ++k)
        {
            int a = add;
            for (int j = 0; j < n; ++j) a += coef(n, trType, k, j) * in[i * stride + j];
            out[k * n + i] = (int16_t)(uint16_t)((uint32_t)(a >> shift) & 0xffff);
        }
}

/* havoc/transform.cpp:3355-3397 */
void oracle_transform(int16_t *coeffs, const int16_t *src, intptr_t stride, int log2, int trType, int bitDepth)
{
    const int n = 1 << log2;
    int16_t tmp[32 * 32];
    fwd_pass(tmp, src, stride, n, trType, log2 - 1 + bitDepth - 8);
    fwd_pass(coeffs, tmp, n, n, trType, log2 + 6);
}

/* inverse: out[j][k] = clip16((sum_i M[i][k] * in[i][j] + add) >> shift)  (havoc/transform.cpp:50-336) */
static void inv_pass(int16_t *out, const int16_t *in, int n, int trType, int shift)
{
    const int add = 1 << (shift - 1);
    for (int j = 0; j < n; ++j)
        for (int k = 0; k < n; ++k)
        {
            int a = add;
            for (int i = 0; i < n; ++i) a += coef(n, trType, i, k) * in[i * n + j];
            out[j * n + k] = (int16_t)clip3(-32768, 32767, a >> shift);
        }
}

/* havoc/transform.cpp:339-355 */
void oracle_inverse_transform(int16_t *dst, const int16_t *coeffs, int log2, int trType, int bitDepth)
{
    const int n = 1 << log2;
    int16_t tmp[32 * 32];
    inv_pass(tmp, coeffs, n, trType, 7);
    inv_pass(dst, tmp, n, trType, 20 - bitDepth);
}

/* havoc/transform.cpp:358-401, transform.h:95-114 */
void oracle_inverse_transform_add(void *dst, intptr_t sd, const void *pred, intptr_t sp, const int16_t *coeffs, int log2, int trType, int bitDepth, int S)
{
    const int n = 1 << log2;
    const int max = (1 << bitDepth) - 1;
    int16_t res[32 * 32];
    oracle_inverse_transform(res, coeffs, log2, trType, bitDepth);
    for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x)
            put(dst, x + y * sd, clip3(0, max, px(pred, x + y * sp, S) + res[y * n + x]), S);
}

/* ------------------------------------------------------------------------------------------------ */
/* quantisation                                                                                     */
/* ------------------------------------------------------------------------------------------------ */

/* havoc/quantize.cpp:37-46 */
void oracle_quantize_inverse(int16_t *dst, const int16_t *src, int scale, int shift, int n)
{
    for (int i = 0; i < n; ++i)
        dst[i] = (int16_t)clip3(-32768, 32767, (src[i] * scale + (1 << (shift - 1))) >> shift);
}

/* havoc/quantize.cpp:278-304 */
int oracle_quantize(int16_t *dst, const int16_t *src, int scale, int shift, int offset, int n)
{
    int cbf = 0;
    offset <<= shift - 16;
    for (int i = 0; i < n; ++i)
    {
        int x = src[i];
        const int sign = x < 0 ? -1 : 1;
        x = ((abs(x) * scale + offset) >> shift) * sign;
        x = clip3(-32768, 32767, x);
        cbf |= x;
        dst[i] = (int16_t)x;
    }
    return cbf;
}

/* havoc/quantize.cpp:538-549 */
void oracle_quantize_reconstruct(uint8_t *rec, intptr_t sr, const uint8_t *pred, intptr_t sp, const int16_t *res, int n)
{
    for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x)
            rec[x + y * sr] = (uint8_t)clip3(0, 255, pred[x + y * sp] + res[x + y * n]);
}

/* turing/Reconstruct.cpp:258-260 */
void oracle_residual(int16_t *res, intptr_t sres, const void *src, intptr_t ss, const void *pred, intptr_t sp, int w, int h, int S)
{
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            res[x + y * sres] = (int16_t)(px(src, x + y * ss, S) - px(pred, x + y * sp, S));
}

/* turing/Padding.h:60-97 (Padding::padBlock; padImage :33-57 is the all-four-sides case): replicate the edge samples of a
 * w x h block into a border of `pad` samples on the requested sides.  The vertical pass copies the already
 * horizontally padded first / last row, so the corners get the corner sample. */
void oracle_pad_block(void *p, int w, int h, intptr_t stride, int pad, int top, int bottom, int left, int right, int S)
{
    unsigned char *b = (unsigned char *)p;
    intptr_t x0 = 0, wide = w;
    for (int y = 0; y < h; ++y)
    {
        unsigned char *row = b + (intptr_t)y * stride * S;
        if (left)
            for (int i = 1; i <= pad; ++i) memcpy(row - (intptr_t)i * S, row, (size_t)S);
        if (right)
            for (int i = 0; i < pad; ++i) memcpy(row + (intptr_t)(w + i) * S, row + (intptr_t)(w - 1) * S, (size_t)S);
    }
    if (left) { x0 -= pad; wide += pad; }
    if (right) wide += pad;
    if (top)
        for (int i = 1; i <= pad; ++i) memcpy(b + (x0 - (intptr_t)i * stride) * S, b + x0 * S, (size_t)(wide * S));
    if (bottom)
        for (int i = 1; i <= pad; ++i)
            memcpy(b + (x0 + (intptr_t)(h - 1 + i) * stride) * S, b + (x0 + (intptr_t)(h - 1) * stride) * S, (size_t)(wide * S));
}

/* ---------------------------------------------------------------------------------------------------------
 * Deblocking filter.  Follows turing/LoopFilter.h: Block (:52-91, packed QP / disable bit and the
 * four 2-bit boundary strengths of an 8x8 luma region), betaTable / tCTable (:217-227), LumaBlockEdge (:229-357),
 * ChromaBlockEdge (:359-400), Picture::deblock (:739-777).  The reference filters CTU by CTU (vertical edges of a region
 * shifted by 8, then horizontal edges: turing/TaskDeblock.cpp:105-127); no edge of one direction touches samples
 * another edge of that direction reads, so that order is equivalent to the two picture passes done here: every vertical
 * edge, then every horizontal edge.  blocks: grid of (W64/8 + 1) x (H64/8 + 1) entries, W64 / H64 = the picture size
 * rounded up to 64 (LoopFilter::Picture's constructor, :435-441).  4:2:0, one slice (offsets are per picture).
 * --------------------------------------------------------------------------------------------------------- */
static const int kBetaTable[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 22, 24, 26, 28,
                                   30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64};
static const int kTcTable[54] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5,
                                 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24};

static int dbk_clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
static int dbk_get(const void *p, long i, int S) { return S == 1 ? ((const uint8_t *)p)[i] : ((const uint16_t *)p)[i]; }
static void dbk_put(void *p, long i, int v, int S)
{
    if (S == 1) ((uint8_t *)p)[i] = (uint8_t)v;
    else ((uint16_t *)p)[i] = (uint16_t)v;
}
static int dbk_qpc(int qPi)   /* turing/Global.h:1417-1423 */
{
    static const int lookup[13] = {29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37};
    if (qPi < 30) return qPi;
    if (qPi > 42) return qPi - 6;
    return lookup[qPi - 30];
}
static int dbk_dsam(int p0, int p3, int q0, int q3, int dpq, int beta, int tC)
{
    return dpq < (beta >> 2) && abs(p3 - p0) + abs(q0 - q3) < (beta >> 3) && abs(p0 - q0) < ((5 * tC + 1) >> 1);
}

/* one 4-sample luma edge segment: s = sample q(0,0); across = step from p to q side, along = step to the next line */
static void dbk_luma_segment(void *pl, long s, long across, long along, int bS, int qpP, int qpQ, int enP, int enQ, int tc2, int beta2, int bd, int S)
{
    if (!bS) return;
#define P(i, k) dbk_get(pl, s - ((i) + 1) * across + (k) * along, S)
#define Q(i, k) dbk_get(pl, s + (i) * across + (k) * along, S)
    const int qPL = (qpQ + qpP + 1) >> 1;
    const int beta = kBetaTable[dbk_clip3(0, 51, qPL + (beta2 << 1))] * (1 << (bd - 8));
    const int tC = kTcTable[dbk_clip3(0, 53, qPL + 2 * (bS - 1) + (tc2 << 1))] * (1 << (bd - 8));
    const int dp0 = abs(P(2, 0) - 2 * P(1, 0) + P(0, 0)), dp3 = abs(P(2, 3) - 2 * P(1, 3) + P(0, 3));
    const int dq0 = abs(Q(2, 0) - 2 * Q(1, 0) + Q(0, 0)), dq3 = abs(Q(2, 3) - 2 * Q(1, 3) + Q(0, 3));
    const int dpq0 = dp0 + dq0, dpq3 = dp3 + dq3, dp = dp0 + dp3, dq = dq0 + dq3, d = dpq0 + dpq3;
    int dE = 0, dEp = 0, dEq = 0;
    if (d < beta)
    {
        const int s0 = dbk_dsam(P(0, 0), P(3, 0), Q(0, 0), Q(3, 0), 2 * dpq0, beta, tC);
        const int s3 = dbk_dsam(P(0, 3), P(3, 3), Q(0, 3), Q(3, 3), 2 * dpq3, beta, tC);
        dE = (s0 && s3) ? 2 : 1;
        if (dp < ((beta + (beta >> 1)) >> 3)) dEp = 1;
        if (dq < ((beta + (beta >> 1)) >> 3)) dEq = 1;
    }
    const int maxv = (1 << bd) - 1;
    for (int k = 0; k < 4 && dE; ++k)
    {
        const int p0 = P(0, k), p1 = P(1, k), p2 = P(2, k), p3 = P(3, k), q0 = Q(0, k), q1 = Q(1, k), q2 = Q(2, k), q3 = Q(3, k);
#define SETP(i, v) dbk_put(pl, s - ((i) + 1) * across + k * along, (v), S)
#define SETQ(i, v) dbk_put(pl, s + (i) * across + k * along, (v), S)
        if (dE == 2)
        {
            if (enP)
            {
                SETP(0, dbk_clip3(p0 - 2 * tC, p0 + 2 * tC, (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3));
                SETP(1, dbk_clip3(p1 - 2 * tC, p1 + 2 * tC, (p2 + p1 + p0 + q0 + 2) >> 2));
                SETP(2, dbk_clip3(p2 - 2 * tC, p2 + 2 * tC, (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3));
            }
            if (enQ)
            {
                SETQ(0, dbk_clip3(q0 - 2 * tC, q0 + 2 * tC, (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3));
                SETQ(1, dbk_clip3(q1 - 2 * tC, q1 + 2 * tC, (p0 + q0 + q1 + q2 + 2) >> 2));
                SETQ(2, dbk_clip3(q2 - 2 * tC, q2 + 2 * tC, (p0 + q0 + q1 + 3 * q2 + 2 * q3 + 4) >> 3));
            }
        }
        else
        {
            int delta = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
            if (abs(delta) < tC * 10)
            {
                delta = dbk_clip3(-tC, tC, delta);
                if (enP) SETP(0, dbk_clip3(0, maxv, p0 + delta));
                if (enQ) SETQ(0, dbk_clip3(0, maxv, q0 - delta));
                if (dEp && enP) SETP(1, dbk_clip3(0, maxv, p1 + dbk_clip3(-(tC >> 1), tC >> 1, (((p2 + p0 + 1) >> 1) - p1 + delta) >> 1)));
                if (dEq && enQ) SETQ(1, dbk_clip3(0, maxv, q1 + dbk_clip3(-(tC >> 1), tC >> 1, (((q2 + q0 + 1) >> 1) - q1 - delta) >> 1)));
            }
        }
    }
#undef P
#undef Q
#undef SETP
#undef SETQ
}

static void dbk_chroma_segment(void *pl, long s, long across, long along, int bS, int qpP, int qpQ, int enP, int enQ, int tc2, int offset, int bd, int S)
{
    if (bS != 2) return;
    const int qPi = ((qpQ + qpP + 1) >> 1) + offset;
    const int tC = kTcTable[dbk_clip3(0, 53, dbk_qpc(qPi) + 2 + (tc2 << 1))] * (1 << (bd - 8));
    const int maxv = (1 << bd) - 1;
    for (int k = 0; k < 4; ++k)
    {
        const int p0 = dbk_get(pl, s - across + k * along, S), p1 = dbk_get(pl, s - 2 * across + k * along, S);
        const int q0 = dbk_get(pl, s + k * along, S), q1 = dbk_get(pl, s + across + k * along, S);
        const int delta = dbk_clip3(-tC, tC, ((((q0 - p0) << 2) + p1 - q1 + 4) >> 3));
        if (enP) dbk_put(pl, s - across + k * along, dbk_clip3(0, maxv, p0 + delta), S);
        if (enQ) dbk_put(pl, s + k * along, dbk_clip3(0, maxv, q0 - delta), S);
    }
}

void oracle_deblock(void *luma, intptr_t stride_y, void *cb, void *cr, intptr_t stride_c, int width, int height, int bitDepth,
                    const int8_t *block_data, const uint8_t *block_bs, int tc_offset_div2, int beta_offset_div2, int cb_qp_offset,
                    int cr_qp_offset, int S)
{
    const int bstride = ((width + 63) / 64) * 8 + 1;
    for (int edge = 0; edge < 2; ++edge)   /* 0 = EDGE_VER, 1 = EDGE_HOR */
        for (int y = 0; y < height / 8; ++y)
            for (int x = 0; x < width / 8; ++x)
            {
                const long q = (long)bstride * y + x, p = edge ? q - bstride : q - 1;
                const int bsAll = block_bs[q];
                const int enQ = !(block_data[q] & 1), qpQ = block_data[q] >> 1;
                for (int pos = 0; pos < 2; ++pos)
                {
                    const int bS = 3 & (bsAll >> (4 * edge + 2 * pos));
                    if (!bS) continue;   /* block P is only looked at behind a non-zero strength (LoopFilter.h:242-247) */
                    const long s = edge ? (long)(8 * y) * stride_y + 8 * x + 4 * pos : (long)(8 * y + 4 * pos) * stride_y + 8 * x;
                    dbk_luma_segment(luma, s, edge ? stride_y : 1, edge ? 1 : stride_y, bS, block_data[p] >> 1, qpQ, !(block_data[p] & 1), enQ,
                                     tc_offset_div2, beta_offset_div2, bitDepth, S);
                }
                if (edge ? (y % 2 == 0) : (x % 2 == 0))
                {
                    const int bS = 3 & (bsAll >> (4 * edge));   /* position 0's strength for the whole chroma segment (:366, :385) */
                    if (bS == 2)
                    {
                        const long s = (long)(4 * y) * stride_c + 4 * x;
                        dbk_chroma_segment(cb, s, edge ? stride_c : 1, edge ? 1 : stride_c, bS, block_data[p] >> 1, qpQ, !(block_data[p] & 1), enQ,
                                           tc_offset_div2, cb_qp_offset, bitDepth, S);
                        dbk_chroma_segment(cr, s, edge ? stride_c : 1, edge ? 1 : stride_c, bS, block_data[p] >> 1, qpQ, !(block_data[p] & 1), enQ,
                                           tc_offset_div2, cr_qp_offset, bitDepth, S);
                    }
                }
            }
}


/* ---- boundary strengths of the deblocking filter from a picture's block structure ------------------------------------------------------
 * turing/LoopFilter.h: a transform unit of an intra coding unit raises the four edges of its luma block that lie on the 8-sample grid to
 * 2 (processTu, :645-690), a coded luma transform block of an inter unit raises them to 1 (processRc, :692-737), the left and top edge of a
 * prediction unit are raised to 1 over every 4-sample segment whose neighbour across the edge has different motion (processPu, :608-643,
 * sameMotion :402-422); strengths only ever go up (increaseBs, :85-89).  The byte of a region: (QpY << 1) | cu_transquant_bypass
 * (processCu :585-605, packData :57-64).  Restated per 4x4 cell: an edge segment looks at the cell on either side. */
typedef struct { int16_t mv[2][2]; int8_t dpb[2]; uint8_t flags; int8_t qp; uint8_t tu_log2; uint8_t pad[3]; } oracle_cell;

static int cell_same1(const oracle_cell *a, int la, const oracle_cell *b, int lb)
{
    if (a->dpb[la] != b->dpb[lb]) return 0;
    if (a->dpb[la] < 0) return 1;
    if (abs(a->mv[la][0] - b->mv[lb][0]) >= 4) return 0;
    if (abs(a->mv[la][1] - b->mv[lb][1]) >= 4) return 0;
    return 1;
}

static int cell_same(const oracle_cell *a, const oracle_cell *b)
{
    if (cell_same1(a, 0, b, 0) && cell_same1(a, 1, b, 1)) return 1;
    return cell_same1(a, 0, b, 1) && cell_same1(a, 1, b, 0);
}

static int cell_edge(const oracle_cell *a, const oracle_cell *b, int pos, int pu_edge_flag)
{
    static const oracle_cell none = {{{0, 0}, {0, 0}}, {-1, -1}, 0, 0, 0, {0, 0, 0}};
    int bs = 0, v;
    if (a && pos % (1 << a->tu_log2) == 0)
    {
        v = (a->flags & 1) ? 2 : ((a->flags & 2) ? 1 : 0);
        if (v > bs) bs = v;
    }
    if (b && pos % (1 << b->tu_log2) == 0)
    {
        v = (b->flags & 1) ? 2 : ((b->flags & 2) ? 1 : 0);
        if (v > bs) bs = v;
    }
    if (b && (b->flags & pu_edge_flag) && !(b->flags & 1) && !cell_same(a ? a : &none, b) && bs < 1) bs = 1;
    return bs;
}

void oracle_derive_bs(const void *cells_, intptr_t cs, int width, int height, int8_t *data, uint8_t *bs)
{
    const oracle_cell *cells = (const oracle_cell *)cells_;
    const int gw = (width + 63) / 64 * 8 + 1, gh = (height + 63) / 64 * 8 + 1, cw = width / 4, ch = height / 4;
    for (int ry = 0; ry < gh; ++ry)
        for (int rx = 0; rx < gw; ++rx)
        {
            int packed = 0;
            int8_t d = 0;
            for (int k = 0; k < 2; ++k)
            {
                /* vertical edge at x = 8 rx, rows 8 ry + 4 k */
                int cy = 2 * ry + k, cx = 2 * rx;
                if (cy < ch)
                {
                    const oracle_cell *a = cx - 1 >= 0 && cx - 1 < cw ? &cells[cy * cs + cx - 1] : 0, *b = cx < cw ? &cells[cy * cs + cx] : 0;
                    if (a || b) packed |= cell_edge(a, b, 8 * rx, 8) << (2 * k);
                }
                /* horizontal edge at y = 8 ry, columns 8 rx + 4 k */
                cx = 2 * rx + k;
                cy = 2 * ry;
                if (cx < cw)
                {
                    const oracle_cell *a = cy - 1 >= 0 && cy - 1 < ch ? &cells[(cy - 1) * cs + cx] : 0, *b = cy < ch ? &cells[cy * cs + cx] : 0;
                    if (a || b) packed |= cell_edge(a, b, 8 * ry, 16) << (4 + 2 * k);
                }
            }
            /* processCtu (LoopFilter.h:484-510) comes after the CTU's units (Decode.h:281-286) and clears the edges on the picture boundary */
            if (rx == 0) packed &= 0xF0;
            if (ry == 0) packed &= 0x0F;
            if (2 * rx < cw && 2 * ry < ch)
            {
                const oracle_cell *c = &cells[2 * ry * cs + 2 * rx];
                d = (int8_t)((c->qp << 1) | ((c->flags & 4) ? 1 : 0));
            }
            data[ry * gw + rx] = d;
            bs[ry * gw + rx] = (uint8_t)packed;
        }
}

/* turing/IntraReferenceSamples.h:373-421 */
void oracle_intra_filter_neighbours(const int32_t *p, int32_t *pF, int nTbS, int bitDepthY, int strong)
{
#define P(x, y) p[(x) - (y)-1]
#define PF(x, y) pF[(x) - (y)-1]
    const int last = nTbS * 2 - 1;
    const int thr = 1 << (bitDepthY - 5);
    const int biInt = strong == 1 && nTbS == 32 && abs(P(-1, -1) + P(last, -1) - 2 * P(nTbS - 1, -1)) < thr &&
                      abs(P(-1, -1) + P(-1, last) - 2 * P(-1, nTbS - 1)) < thr;
    if (biInt)
    {   /* :384-397 */
        PF(-1, -1) = P(-1, -1);
        for (int y = 0; y <= 62; ++y) PF(-1, y) = ((63 - y) * P(-1, -1) + (y + 1) * P(-1, 63) + 32) >> 6;
        PF(-1, 63) = P(-1, 63);
        for (int x = 0; x <= 62; ++x) PF(x, -1) = ((63 - x) * P(-1, -1) + (x + 1) * P(63, -1) + 32) >> 6;
        PF(63, -1) = P(63, -1);
    }
    else
    {   /* :400-412 */
        PF(-1, -1) = (P(-1, 0) + 2 * P(-1, -1) + P(0, -1) + 2) >> 2;
        for (int y = 0; y <= last - 1; ++y) PF(-1, y) = (P(-1, y + 1) + 2 * P(-1, y) + P(-1, y - 1) + 2) >> 2;
        PF(-1, last) = P(-1, last);
        for (int x = 0; x <= last - 1; ++x) PF(x, -1) = (P(x - 1, -1) + 2 * P(x, -1) + P(x + 1, -1) + 2) >> 2;
        PF(last, -1) = P(last, -1);
    }
#undef P
#undef PF
}

/* HEVC 8.4.4.2.2 / turing/IntraReferenceSamples.h:286-346: search from the bottom of the left column for the first available sample; everything before it
 * takes its value; every later unavailable sample takes its predecessor's; nothing available -> 1 << (bitDepth - 1) */
void oracle_intra_substitute(int32_t *val, const uint8_t *have, int nTbS, int bitDepthY)
{
    const int len = 4 * nTbS + 1;
    int first = 0;
    while (first < len && !have[first]) ++first;
    if (first == len)
    {
        for (int k = 0; k < len; ++k) val[k] = 1 << (bitDepthY - 1);
        return;
    }
    for (int k = 0; k < first; ++k) val[k] = val[first];
    for (int k = first + 1; k < len; ++k)
        if (!have[k]) val[k] = val[k - 1];
}
