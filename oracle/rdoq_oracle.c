/* TEST INFRASTRUCTURE ONLY (see oracle/README.md): plain-C restatement of the reference's rate-distortion optimised quantiser,
 * /root/reference/turing/Rdoq.cpp:35-1023 with Rdoq.h:163-187 (constructor), the scan tables of turing/ScanOrder.h:31-95 and the
 * bit-cost table of turing/Write.h:413-422.  Pinned against the reference's own Rdoq.cpp compiled into oracle/_ref
 * (oracle/ref_shim_rdoq.cpp) by tests/test_rdoq.py.
 *
 * Everything is integer: costs are Q16 int64 (turing/Cost.h:33 `Cost`), lambda and the distortion scale Q16 int32 (`Lambda`,
 * Rdoq.h:59), bit counts Q15.  The probability states are read, never updated (Rdoq.cpp:28-33 estimateBits), which is what
 * makes a frozen snapshot of 122 state bytes the whole of the entropy coder this path needs.
 *
 * Written as one sequential pass per transform block, coefficient groups and coefficients in reverse scan order, like the
 * reference; the device kernel (turingcodec_amd/csrc/kernels_rdoq.hip) evaluates the same recurrences in a different order and
 * must reproduce these results bit for bit. */
#include "havoc_oracle.h"
#include <limits.h>
#include <stdlib.h>
#include <string.h>

/* turing/Write.h:413-422: estimated bits (Q15) of coding the more / less probable symbol from each of the 64 CABAC states */
static const int32_t kBits[128] = {
    0x07b23, 0x085f9, 0x074a0, 0x08cbc, 0x06ee4, 0x09354, 0x067f4, 0x09c1b, 0x060b0, 0x0a62a, 0x05a9c, 0x0af5b, 0x0548d, 0x0b955, 0x04f56, 0x0c2a9,
    0x04a87, 0x0cbf7, 0x045d6, 0x0d5c3, 0x04144, 0x0e01b, 0x03d88, 0x0e937, 0x039e0, 0x0f2cd, 0x03663, 0x0fc9e, 0x03347, 0x10600, 0x03050, 0x10f95,
    0x02d4d, 0x11a02, 0x02ad3, 0x12333, 0x0286e, 0x12cad, 0x02604, 0x136df, 0x02425, 0x13f48, 0x021f4, 0x149c4, 0x0203e, 0x1527b, 0x01e4d, 0x15d00,
    0x01c99, 0x166de, 0x01b18, 0x17017, 0x019a5, 0x17988, 0x01841, 0x18327, 0x016df, 0x18d50, 0x015d9, 0x19547, 0x0147c, 0x1a083, 0x0138e, 0x1a8a3,
    0x01251, 0x1b418, 0x01166, 0x1bd27, 0x01068, 0x1c77b, 0x00f7f, 0x1d18e, 0x00eda, 0x1d91a, 0x00e19, 0x1e254, 0x00d4f, 0x1ec9a, 0x00c90, 0x1f6e0,
    0x00c01, 0x1fef8, 0x00b5f, 0x208b1, 0x00ab6, 0x21362, 0x00a15, 0x21e46, 0x00988, 0x2285d, 0x00934, 0x22ea8, 0x008a8, 0x239b2, 0x0081d, 0x24577,
    0x007c9, 0x24ce6, 0x00763, 0x25663, 0x00710, 0x25e8f, 0x006a0, 0x26a26, 0x00672, 0x26f23, 0x005e8, 0x27ef8, 0x005ba, 0x284b5, 0x0055e, 0x29057,
    0x0050c, 0x29bab, 0x004c1, 0x2a674, 0x004a7, 0x2aa5e, 0x0046f, 0x2b32f, 0x0041f, 0x2c0ad, 0x003e7, 0x2ca8d, 0x003ba, 0x2d323, 0x0010c, 0x3bfbb };

/* offsets into the flat state array (include/havoc_mi355x.h, HAVOC_RDOQ_CTX_*) */
enum { CTX_ROOT_CBF = 0, CTX_CBF_LUMA = 1, CTX_CBF_CHROMA = 3, CTX_LAST_X = 8, CTX_LAST_Y = 26, CTX_CSBF = 44, CTX_SIG = 48, CTX_G1 = 92, CTX_G2 = 116 };
enum { MAX_G1_BINS = 8, MAX_G2_BINS = 1 };   /* Rdoq.h:32-33 */

/* Rdoq.cpp:28-33 -- note the index is (state >> 1) ^ bin (ContextModel.h:59-62 getState) */
static int32_t bits_of(const uint8_t *states, int ctx, int bin) { return kBits[(states[ctx] >> 1) ^ bin]; }

/* ScanOrder.h:31-95: position `pos` of the up-right diagonal (0), horizontal (1) or vertical (2) scan of a size x size block */
static void scan_xy(int size, int scanIdx, int pos, int *x, int *y)
{
    if (scanIdx == 1) { *x = pos % size; *y = pos / size; return; }
    if (scanIdx == 2) { *x = pos / size; *y = pos % size; return; }
    for (int d = 0;; ++d)      /* anti-diagonal d holds the in-range points (x, d - x), x ascending */
    {
        int lo = d < size ? 0 : d - size + 1, hi = d < size ? d : size - 1, len = hi - lo + 1;
        if (pos < len) { *x = lo + pos; *y = d - *x; return; }
        pos -= len;
    }
}

int oracle_scan_order(int log2BlockSize, int scanIdx, int sPos, int sComp)   /* ScanOrder.h:212-223 */
{
    int x = 0, y = 0;
    if (log2BlockSize < 1 || log2BlockSize > 5) return 0;
    scan_xy(1 << log2BlockSize, scanIdx, sPos, &x, &y);
    return sComp ? y : x;
}

/* Rdoq.h:163-187: the two numbers the constructor derives from the floating-point lambda */
void oracle_rdoq_lambda(double lambda, int invQuantScale, int32_t *lambdaQ16, int32_t *sdhFactor)
{
    *lambdaQ16 = (int32_t)(lambda * 65536 + 0.5);                            /* FixedPoint.h:47-50 */
    *sdhFactor = (int)(invQuantScale * invQuantScale / lambda / 16 + 0.5);   /* Rdoq.h:166 */
}

typedef struct
{
    const uint8_t *states;
    int64_t lambda;          /* Q16 */
    int32_t distScale;       /* Q16 */
    int invScale, invShift, invOffset;
    int cIdx;
    int64_t costCoded[1024]; /* by scan position: best RD cost of the coefficient (Rdoq.h:62 m_rdCostCoeff) */
    int64_t costSig[1024];   /* lambda * bits of its significance flag (m_rateCostCoeffSig) */
    int64_t dist0[1024];     /* distortion if rounded to zero (m_distCoeff0) */
} Engine;

/* the five variables the entropy coder's level binarisation carries from coefficient to coefficient (Rdoq.cpp:44-49) */
typedef struct { int ctxSet, c1, nG1, nG2, rice; } LevelState;

static int clip16(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

static int base_level(const LevelState *s) { return s->nG1 < MAX_G1_BINS ? 2 + (s->nG2 < MAX_G2_BINS) : 1; }

/* Rdoq.cpp:611-668 getLevelRateCost: lambda * (sign bit + greater1 / greater2 flags + Golomb-Rice / exp-Golomb remainder) */
static int64_t level_cost(const Engine *e, int level, int g1, int g2, const LevelState *s)
{
    int32_t rate = 32768;
    const int base = base_level(s);
    if (level >= base)
    {
        int symbol = level - base, length;
        if (symbol < (3 << s->rice))
        {
            length = symbol >> s->rice;
            rate += (length + 1 + s->rice) << 15;
        }
        else
        {
            length = s->rice;
            symbol -= 3 << s->rice;
            while (symbol >= (1 << length)) symbol -= 1 << length++;
            rate += (3 + length + 1 - s->rice + length) << 15;
        }
        if (s->nG1 < MAX_G1_BINS)
        {
            rate += bits_of(e->states, CTX_G1 + g1, 1);
            if (s->nG2 < MAX_G2_BINS) rate += bits_of(e->states, CTX_G2 + g2, 1);
        }
    }
    else if (level == 1)
        rate += bits_of(e->states, CTX_G1 + g1, 0);
    else if (level == 2)
        rate += bits_of(e->states, CTX_G1 + g1, 1) + bits_of(e->states, CTX_G2 + g2, 0);
    return e->lambda * rate;
}

/* Rdoq.cpp:819-885 getLevelRate: the (differently binarised) rate the sign-data-hiding stage works with */
static int level_rate(const Engine *e, int level, int g1, int g2, const LevelState *s)
{
    static const int range[5] = { 7, 14, 26, 46, 78 }, prefixLen[5] = { 8, 7, 6, 5, 4 };
    int rate = 0;
    const int base = base_level(s);
    if (level >= base)
    {
        int symbol = level - base;
        const int maxVlc = range[s->rice];
        if (symbol > maxVlc)
        {
            int rest = symbol - maxVlc, egs = 1;
            for (int top = 2; rest >= top; top <<= 1) egs += 2;
            rate += egs << 15;
            symbol = maxVlc + 1;
        }
        int prefix = symbol >> (s->rice + 1);
        rate += ((prefix < prefixLen[s->rice] ? prefix : prefixLen[s->rice]) + s->rice) << 15;
        if (s->nG1 < MAX_G1_BINS)
        {
            rate += bits_of(e->states, CTX_G1 + g1, 1);
            if (s->nG2 < MAX_G2_BINS) rate += bits_of(e->states, CTX_G2 + g2, 1);
        }
    }
    else if (level == 1)
        rate += bits_of(e->states, CTX_G1 + g1, 0);
    else if (level == 2)
        rate += bits_of(e->states, CTX_G1 + g1, 1) + bits_of(e->states, CTX_G2 + g2, 0);
    return rate;
}

/* Rdoq.h:137-142 */
static int dequantised(const Engine *e, int level) { return clip16((clip16(level) * e->invScale + e->invOffset) >> e->invShift); }

/* Rdoq.cpp:456-515 getAdjustedQuantLevel: keep the rounded level, lower it by one, or (small levels) drop it */
static int choose_level(Engine *e, int sp, int absCoeff, int level, int sigCtx, int g1, int g2, const LevelState *s, int first)
{
    int64_t sigOne = 0;
    int best = 0;
    if (!first && level < 3)
    {
        e->costSig[sp] = e->lambda * bits_of(e->states, CTX_SIG + sigCtx, 0);
        e->costCoded[sp] = e->dist0[sp] + e->costSig[sp];
        if (level == 0) return 0;
    }
    else
        e->costCoded[sp] = INT64_MAX;
    if (!first) sigOne = e->lambda * bits_of(e->states, CTX_SIG + sigCtx, 1);
    for (int l = level, lowest = level > 1 ? level - 1 : 1; l >= lowest; --l)
    {
        const int32_t err = absCoeff - dequantised(e, l);
        const int64_t cost = (int64_t)(int32_t)((uint32_t)err * (uint32_t)err) * e->distScale + level_cost(e, l, g1, g2, s) + sigOne;
        if (cost < e->costCoded[sp])
        {
            best = l;
            e->costCoded[sp] = cost;
            e->costSig[sp] = sigOne;
        }
    }
    return best;
}

/* Rdoq.cpp:517-603 getCoeffSigCtxInc */
static int sig_ctx(int csbfNeighbours, int scanIdx, int x, int y, int log2, int cIdx)
{
    static const int map4x4[16] = { 0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8 };
    int inc;
    if (log2 == 2)
        inc = map4x4[(y << 2) + x];
    else if (x + y == 0)
        inc = 0;
    else
    {
        const int xp = x & 3, yp = y & 3;
        if (csbfNeighbours == 0) inc = xp + yp == 0 ? 2 : (xp + yp < 3 ? 1 : 0);
        else if (csbfNeighbours == 1) inc = yp == 0 ? 2 : (yp == 1 ? 1 : 0);
        else if (csbfNeighbours == 2) inc = xp == 0 ? 2 : (xp == 1 ? 1 : 0);
        else inc = 2;
        if (cIdx == 0)
        {
            if ((x >> 2) + (y >> 2) > 0) inc += 3;
            inc += log2 == 3 ? (scanIdx == 0 ? 9 : 15) : 21;
        }
        else
            inc += log2 == 3 ? 9 : 12;
    }
    return cIdx == 0 ? inc : 27 + inc;
}

/* Rdoq.cpp:606-624 getPrevCsbf (right + 2 * below) and :670-693 getCgSigCtxInc */
static int csbf_neighbours(const int *csbf, int xs, int ys, int log2)
{
    const int w = 1 << (log2 - 2);
    return (xs < w - 1 ? csbf[ys * w + xs + 1] : 0) + (ys < w - 1 ? csbf[(ys + 1) * w + xs] << 1 : 0);
}
static int csbf_ctx(const int *csbf, int xs, int ys, int log2, int cIdx)
{
    const int w = 1 << (log2 - 2);
    const int sum = (xs < w - 1 ? csbf[ys * w + xs + 1] : 0) + (ys < w - 1 ? csbf[(ys + 1) * w + xs] : 0);
    return (cIdx ? 2 : 0) + (sum < 1 ? sum : 1);
}

/* Rdoq.cpp:706-763 getLastSigCoeffPosRateCost with :765-771 */
static int64_t last_position_cost(const Engine *e, int xc, int yc, int log2)
{
    static const int32_t prefixBins[32] = { 0, 1, 2, 3, 4, 4, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 8, 8, 8, 8, 8, 8, 8, 8, 9, 9, 9, 9, 9, 9, 9, 9 };
    const int offset = e->cIdx ? 15 : 3 * (log2 - 2) + ((log2 - 1) >> 2), shift = e->cIdx ? log2 - 2 : (log2 + 1) >> 2;
    const int len[2] = { prefixBins[xc], prefixBins[yc] }, table[2] = { CTX_LAST_X, CTX_LAST_Y };
    int32_t rate = 0;
    for (int k = 0; k < 2; ++k)
    {
        for (int i = 0; i <= len[k] && i < 9; ++i)
        {
            int ctx = (i >> shift) + offset;
            ctx = ctx < 0 ? 0 : (ctx > 17 ? 17 : ctx);
            rate += bits_of(e->states, table[k] + ctx, i < len[k]);
        }
        if (len[k] > 3) rate += 32768 * ((len[k] - 2) >> 1);
    }
    return e->lambda * rate;
}

/* Rdoq.cpp:773-817 updateEntropyCodingEngine */
static void advance(LevelState *s, int level, int sp, int cIdx)
{
    if (level >= base_level(s) && level > 3 * (1 << s->rice)) s->rice = s->rice + 1 < 4 ? s->rice + 1 : 4;
    if (level >= 1) s->nG1++;
    if (level > 1)
    {
        s->c1 = 0;
        s->nG2++;
    }
    else if (s->c1 < 3 && s->c1 > 0 && level)
        s->c1++;
    if (sp % 16 == 0 && sp > 0)     /* the next coefficient opens a new group */
    {
        s->rice = s->nG1 = s->nG2 = 0;
        s->ctxSet = (sp == 16 || cIdx != 0) ? 0 : 2;
        if (s->c1 == 0) s->ctxSet++;
        s->c1 = 1;
    }
}

/* Rdoq.cpp:887-1023 signDataHiding */
static void hide_signs(int nGroups, int16_t *dst, const int16_t *src, const int *scan, const int *rateUp, const int *rateDown, const int *sigDelta,
                       const int *deltaU, int factor)
{
    int lastGroup = -1;
    for (int g = nGroups - 1; g >= 0; --g)
    {
        const int *pos = scan + (g << 4);
        int first = 16, last = -1, sum = 0;
        for (int i = 15; i >= 0; --i) if (dst[pos[i]]) { last = i; break; }
        for (int i = 0; i < 16; ++i) if (dst[pos[i]]) { first = i; break; }
        for (int i = first; i <= last; ++i) sum += dst[pos[i]];
        if (last >= 0 && lastGroup == -1) lastGroup = 1;
        if (last - first >= 4)
        {
            const int signbit = dst[pos[first]] > 0 ? 0 : 1;
            if (signbit != (sum & 1))
            {
                int minCost = INT_MAX, cost = INT_MAX, minPos = -1, finalChange = 0, change = 0;
                for (int i = lastGroup == 1 ? last : 15; i >= 0; --i)
                {
                    const int p = pos[i];
                    if (dst[p] != 0)
                    {
                        const int up = factor * -deltaU[p] + rateUp[p];
                        int down = factor * deltaU[p] + rateDown[p] - (abs(dst[p]) == 1 ? (1 << 15) + sigDelta[p] : 0);
                        if (lastGroup == 1 && last == i && abs(dst[p]) == 1) down -= 4 << 15;
                        if (up < down)
                        {
                            cost = up;
                            change = 1;
                        }
                        else
                        {
                            change = -1;
                            cost = (i == first && abs(dst[p]) == 1) ? INT_MAX : down;
                        }
                    }
                    else
                    {
                        cost = factor * -abs(deltaU[p]) + (1 << 15) + rateUp[p] + sigDelta[p];
                        change = 1;
                        if (i < first && (src[p] >= 0 ? 0 : 1) != signbit) cost = INT_MAX;
                    }
                    if (cost < minCost)
                    {
                        minCost = cost;
                        finalChange = change;
                        minPos = p;
                    }
                }
                if (dst[minPos] == 32767 || dst[minPos] == -32768) finalChange = -1;
                dst[minPos] = (int16_t)(src[minPos] >= 0 ? dst[minPos] + finalChange : dst[minPos] - finalChange);
            }
        }
        if (lastGroup == 1) lastGroup = 0;
    }
}

/* Rdoq.cpp:37-454 runQuantisation.  Returns the OR of the kept absolute levels (non-zero = coded block flag). */
int oracle_rdoq(int16_t *dst, const int16_t *src, int log2Size, int cIdx, int scanIdx, int isIntra, int sdh, int quantScale, int quantShift,
                int invScale, int bitDepth, int32_t lambdaQ16, int32_t sdhFactor, const uint8_t *states)
{
    const int n = 1 << 2 * log2Size, nGroups = n >> 4, log2Groups = log2Size - 2, groupsWide = 1 << log2Groups;
    Engine *e = (Engine *)calloc(1, sizeof(Engine));
    int *work = (int *)calloc(5 * 1024, sizeof(int));
    int *rateUp = work, *rateDown = work + 1024, *sigDelta = work + 2048, *deltaU = work + 3072, *scan = work + 4096;
    int csbf[64] = { 0 };
    int64_t groupSigCost[64] = { 0 };
    int64_t dist0Total = 0, costTu = 0;
    LevelState st = { 0, 1, 0, 0, 0 };
    int firstPos = -1, firstGroup = -1;    /* first non-zero level met in reverse scan = candidate last significant position */
    int cbf = 0;

    {   /* Rdoq.h:163-187 */
        const int transformShift = 15 - bitDepth - log2Size;
        e->states = states;
        e->lambda = lambdaQ16;
        e->distScale = (int32_t)(((double)(1 << (15 - 2 * transformShift - 2 * (bitDepth - 8)))) * 65536 + 0.5);
        e->invScale = invScale;
        e->invShift = 20 - 14 - transformShift;
        e->invOffset = 1 << (e->invShift - 1);
        e->cIdx = cIdx;
    }
    for (int g = 0; g < nGroups; ++g)      /* Rdoq.cpp:403-416: scan position -> raster position */
    {
        int gx = 0, gy = 0, x, y;
        if (log2Groups) scan_xy(groupsWide, scanIdx, g, &gx, &gy);
        for (int i = 0; i < 16; ++i)
        {
            scan_xy(4, scanIdx, i, &x, &y);
            scan[g * 16 + i] = (((gy << 2) + y) << log2Size) + (gx << 2) + x;
        }
    }

    /* step 1 + 2 (Rdoq.cpp:83-298): per group, choose levels, then weigh zeroing the whole group */
    for (int g = nGroups - 1; g >= 0; --g)
    {
        const int gx = (scan[g * 16] & ((1 << log2Size) - 1)) >> 2, gy = scan[g * 16] >> (log2Size + 2), gPos = gy * groupsWide + gx;
        const int neighbours = csbf_neighbours(csbf, gx, gy, log2Size);
        int nonZeroAbovePos0 = 0;
        int64_t gDist0 = 0, gSig = 0, gSigPos0 = 0, gCoded = 0;

        for (int i = 15; i >= 0; --i)
        {
            const int sp = g * 16 + i, p = scan[sp], x = p & ((1 << log2Size) - 1), y = p >> log2Size;
            const int a = abs(src[p]), scaled = a * quantScale;
            const int level = (scaled + (1 << (quantShift - 1))) >> quantShift;
            e->dist0[sp] = (int64_t)(a * a) * e->distScale;
            dist0Total += e->dist0[sp];
            dst[p] = (int16_t)level;
            if (level > 0 && firstPos < 0)
            {
                firstPos = sp;
                firstGroup = g;
                st.ctxSet = (sp < 16 || cIdx != 0) ? 0 : 2;
            }
            if (firstPos >= 0)
            {
                const int g1 = 4 * st.ctxSet + st.c1 + (cIdx > 0 ? 16 : 0), g2 = st.ctxSet + (cIdx > 0 ? 4 : 0);
                const int sc = sig_ctx(neighbours, scanIdx, x, y, log2Size, cIdx);
                const int kept = choose_level(e, sp, a, level, sc, g1, g2, &st, sp == firstPos);
                deltaU[p] = (scaled - (kept << quantShift)) >> (quantShift - 8);
                if (sp != firstPos) sigDelta[p] = bits_of(states, CTX_SIG + sc, 1) - bits_of(states, CTX_SIG + sc, 0);
                if (kept > 0)
                {
                    const int now = level_rate(e, kept, g1, g2, &st);
                    rateUp[p] = level_rate(e, kept + 1, g1, g2, &st) - now;
                    rateDown[p] = level_rate(e, kept - 1, g1, g2, &st) - now;
                }
                else
                    rateUp[p] = bits_of(states, CTX_G1 + g1, 0);
                dst[p] = (int16_t)kept;
                costTu += e->costCoded[sp];
                advance(&st, kept, sp, cIdx);
            }
            else
                costTu += e->dist0[sp];
            gSig += e->costSig[sp];
            if (i == 0) gSigPos0 = e->costSig[sp];
            if (dst[p])
            {
                csbf[gPos] = 1;
                gCoded += e->costCoded[sp] - e->costSig[sp];
                gDist0 += e->dist0[sp];
                if (i != 0) nonZeroAbovePos0++;
            }
        }

        if (firstGroup < 0) continue;
        if (g == 0)
        {
            csbf[gPos] = 1;      /* the DC group is always coded */
            continue;
        }
        if (csbf[gPos] == 0)
        {
            const int64_t zero = e->lambda * bits_of(states, CTX_CSBF + csbf_ctx(csbf, gx, gy, log2Size, cIdx), 0);
            costTu += zero - gSig;
            groupSigCost[g] = zero;
        }
        else if (g < firstGroup)
        {
            if (nonZeroAbovePos0 == 0)
            {
                costTu -= gSigPos0;
                gSig -= gSigPos0;
            }
            const int ctx = CTX_CSBF + csbf_ctx(csbf, gx, gy, log2Size, cIdx);
            const int64_t zero = e->lambda * bits_of(states, ctx, 0), one = e->lambda * bits_of(states, ctx, 1);
            const int64_t allZero = costTu + zero + gDist0 - gCoded - gSig;
            costTu += one;
            groupSigCost[g] = one;
            if (allZero < costTu)
            {
                csbf[gPos] = 0;
                costTu = allZero;
                groupSigCost[g] = zero;
                for (int i = 15; i >= 0; --i)
                {
                    const int sp = g * 16 + i;
                    if (dst[scan[sp]])
                    {
                        dst[scan[sp]] = 0;
                        e->costCoded[sp] = e->dist0[sp];
                        e->costSig[sp] = 0;
                    }
                }
            }
        }
    }

    if (firstPos >= 0)
    {
        /* step 3 (Rdoq.cpp:307-399): where to put the last significant coefficient; start from "code nothing" */
        int64_t best;
        int lastIdx = 0, stop = 0, absSum = 0;
        {
            const int ctx = (!isIntra && cIdx == 0) ? CTX_ROOT_CBF : (cIdx == 0 ? CTX_CBF_LUMA + 1 : CTX_CBF_CHROMA + 0);
            best = dist0Total + e->lambda * bits_of(states, ctx, 0);
            costTu += e->lambda * bits_of(states, ctx, 1);
        }
        for (int g = firstGroup; g >= 0 && !stop; --g)
        {
            const int gx = (scan[g * 16] & ((1 << log2Size) - 1)) >> 2, gy = scan[g * 16] >> (log2Size + 2);
            costTu -= groupSigCost[g];
            if (!csbf[gy * groupsWide + gx]) continue;
            for (int i = 15; i >= 0; --i)
            {
                const int sp = g * 16 + i, p = scan[sp];
                if (sp > firstPos) continue;
                if (dst[p])
                {
                    const int x = p & ((1 << log2Size) - 1), y = p >> log2Size;
                    const int64_t total = costTu + (scanIdx == 2 ? last_position_cost(e, y, x, log2Size) : last_position_cost(e, x, y, log2Size)) - e->costSig[sp];
                    if (total < best)
                    {
                        lastIdx = sp + 1;
                        best = total;
                    }
                    if (dst[p] > 1)
                    {
                        stop = 1;
                        break;
                    }
                    costTu += e->dist0[sp] - e->costCoded[sp];
                }
                else
                    costTu -= e->costSig[sp];
            }
        }
        /* Rdoq.cpp:418-441: signs back on the kept levels, zeros above the chosen last position, then sign-data hiding */
        for (int sp = 0; sp < lastIdx; ++sp)
        {
            const int p = scan[sp], level = dst[p];
            absSum += level;
            dst[p] = (int16_t)(src[p] < 0 ? -level : level);
            cbf |= level;
        }
        for (int sp = lastIdx; sp <= firstPos; ++sp) dst[scan[sp]] = 0;
        if (sdh && absSum >= 2) hide_signs(nGroups, dst, src, scan, rateUp, rateDown, sigDelta, deltaU, sdhFactor);
    }
    free(work);
    free(e);
    return cbf;
}
