// k_rdoq : rate-distortion optimised quantisation of transform blocks.
//
// Reference: turing/Rdoq.cpp:37-454 runQuantisation with its helpers (:456-885), sign-data hiding (:887-1023), the constructor
// turing/Rdoq.h:163-187, the scans of turing/ScanOrder.h:31-95 and the bit-cost table turing/Write.h:413-422; called between the
// forward transform and the reconstruction at turing/Reconstruct.cpp:289-312 (intra) and :794-812 (inter).
//
// The reference walks a block's coefficients one by one in reverse scan order, carrying the entropy coder's level state and
// three running costs.  What actually couples one 4x4 coefficient group to the rest of the block is small:
//   * whether the groups to its right and below ended up coded (2 bits: they pick the significance contexts, Rdoq.cpp:517-624,
//     and the group flag's context, :670-693),
//   * whether the previous group in scan order ended with a level > 1 (1 bit: it bumps the greater-than-one context set,
//     Rdoq.cpp:806-816),
//   * where the first non-zero rounded level sits (known after plain quantisation),
//   * sums of Q16 costs, which are integers and therefore associative.
// So a workgroup takes 64 coefficient groups (one 32x32 block, four 16x16, sixteen 8x8 or sixty-four 4x4 blocks) and
//   pass 1  evaluates every group under all 8 possible values of those 3 bits at once (512 lanes, 16 coefficients each, the
//           level choice of Rdoq.cpp:456-515 and the group zeroing of :196-297 included) and keeps 2 bits per case: "group
//           stays coded" and "ends with a level > 1";
//   resolve one lane per block follows the chain of those bits through the groups in reverse scan order (no arithmetic);
//   pass 2  64 lanes redo their group under the case that really applies and keep the per-coefficient costs in LDS;
//   last    the search for the last significant position (Rdoq.cpp:342-399) becomes a suffix sum of per-group cost deltas,
//           a per-lane walk of 16 coefficients and a (cost, position) minimum; the early exit at the first level > 1 becomes
//           a maximum; signs, truncation and sign-data hiding (one group per lane, groups are independent there) follow.
// Integer throughout: Q16 int64 costs, Q16 int32 lambda / distortion scale, Q15 bit counts -- bit-exact by construction as
// long as every sum adds the same terms.
#include "common.h"
#include <cstdlib>
#include <cstring>

namespace havoc_gpu {

namespace {

constexpr int kGroups = 64;              // coefficient groups per workgroup
constexpr int kCases = 8;                // right-coded | below-coded << 1 | carry << 2
constexpr int kRdoqThreads = kGroups * kCases;

// turing/Write.h:413-422: estimated bits (Q15) for the more / less probable symbol from each CABAC state
__device__ const int32_t kEntropyBits[128] = {
    0x07b23, 0x085f9, 0x074a0, 0x08cbc, 0x06ee4, 0x09354, 0x067f4, 0x09c1b, 0x060b0, 0x0a62a, 0x05a9c, 0x0af5b, 0x0548d, 0x0b955, 0x04f56, 0x0c2a9,
    0x04a87, 0x0cbf7, 0x045d6, 0x0d5c3, 0x04144, 0x0e01b, 0x03d88, 0x0e937, 0x039e0, 0x0f2cd, 0x03663, 0x0fc9e, 0x03347, 0x10600, 0x03050, 0x10f95,
    0x02d4d, 0x11a02, 0x02ad3, 0x12333, 0x0286e, 0x12cad, 0x02604, 0x136df, 0x02425, 0x13f48, 0x021f4, 0x149c4, 0x0203e, 0x1527b, 0x01e4d, 0x15d00,
    0x01c99, 0x166de, 0x01b18, 0x17017, 0x019a5, 0x17988, 0x01841, 0x18327, 0x016df, 0x18d50, 0x015d9, 0x19547, 0x0147c, 0x1a083, 0x0138e, 0x1a8a3,
    0x01251, 0x1b418, 0x01166, 0x1bd27, 0x01068, 0x1c77b, 0x00f7f, 0x1d18e, 0x00eda, 0x1d91a, 0x00e19, 0x1e254, 0x00d4f, 0x1ec9a, 0x00c90, 0x1f6e0,
    0x00c01, 0x1fef8, 0x00b5f, 0x208b1, 0x00ab6, 0x21362, 0x00a15, 0x21e46, 0x00988, 0x2285d, 0x00934, 0x22ea8, 0x008a8, 0x239b2, 0x0081d, 0x24577,
    0x007c9, 0x24ce6, 0x00763, 0x25663, 0x00710, 0x25e8f, 0x006a0, 0x26a26, 0x00672, 0x26f23, 0x005e8, 0x27ef8, 0x005ba, 0x284b5, 0x0055e, 0x29057,
    0x0050c, 0x29bab, 0x004c1, 0x2a674, 0x004a7, 0x2aa5e, 0x0046f, 0x2b32f, 0x0041f, 0x2c0ad, 0x003e7, 0x2ca8d, 0x003ba, 0x2d323, 0x0010c, 0x3bfbb };

struct RdoqJob   // == havoc_mi355x_rdoq_job
{
    int32_t dst_off, src_off, quant_scale, quant_shift, inv_scale, lambda_q16, sdh_factor, ctx_index;
    uint8_t c_idx, scan_idx, is_intra, sdh;
    int32_t reserved[3];
};
static_assert(sizeof(RdoqJob) == 48 && sizeof(havoc_mi355x_rdoq_job) == 48, "rdoq job layout");

// what a lane knows about its transform block
struct Block
{
    const uint8_t *states;    // LDS: this block's 128 state bytes, `stateStride` apart
    const int32_t *bits;      // LDS: kEntropyBits
    const int16_t *src;       // LDS: coefficient (x, y) at src[(y & coefMask) * coefRow + (x & coefMask) * coefCol]
    int stateStride, coefMask, coefRow, coefCol;
    int64_t lambda;
    int32_t distScale;
    int quantScale, quantShift, invScale, invShift, invOffset;
    int cIdx, scanIdx;
    uint64_t scan4;           // the 4x4 scan as 16 nibbles x | y << 2
};

struct LevelState { int ctxSet, c1, nG1, nG2, rice; };   // Rdoq.cpp:44-49

__device__ __forceinline__ int32_t bitsOf(const Block &b, int ctx, int bin) { return b.bits[(b.states[ctx * b.stateStride] >> 1) ^ bin]; }
__device__ __forceinline__ int coefAt(const Block &b, int x, int y) { return b.src[(y & b.coefMask) * b.coefRow + (x & b.coefMask) * b.coefCol]; }
__device__ __forceinline__ int baseLevel(const LevelState &s) { return s.nG1 < 8 ? 2 + (s.nG2 < 1) : 1; }
__device__ __forceinline__ int clip16(int v) { return min(max(v, -32768), 32767); }

// ScanOrder.h:31-53: position `pos` of the up-right diagonal scan of a size x size block
__device__ __forceinline__ void diagXy(int size, int pos, int &x, int &y)
{
    for (int d = 0;; ++d)
    {
        const int lo = d < size ? 0 : d - size + 1, hi = d < size ? d : size - 1, len = hi - lo + 1;
        if (pos < len) { x = lo + pos; y = d - x; return; }
        pos -= len;
    }
}
__device__ __forceinline__ void scanXy(int size, int scanIdx, int pos, int &x, int &y)
{
    if (scanIdx == 1) { x = pos & (size - 1); y = pos / size; }
    else if (scanIdx == 2) { x = pos / size; y = pos & (size - 1); }
    else diagXy(size, pos, x, y);
}
__host__ __device__ constexpr uint64_t scan4Nibbles(int scanIdx)
{
    uint64_t v = 0;
    int i = 0;
    if (scanIdx == 0)
    {
        for (int d = 0; d < 7; ++d)
            for (int x = 0; x <= d; ++x)
                if (x < 4 && d - x < 4) { v |= (uint64_t)(x | (d - x) << 2) << (4 * i); ++i; }
    }
    else
        for (; i < 16; ++i) v |= (uint64_t)(scanIdx == 1 ? i : (i >> 2) | (i & 3) << 2) << (4 * i);
    return v;
}

// Rdoq.cpp:611-668 getLevelRateCost (without the lambda)
__device__ __forceinline__ int32_t levelBits(const Block &b, int level, int g1, int g2, const LevelState &s)
{
    int32_t rate = 32768;
    const int base = baseLevel(s);
    if (level >= base)
    {
        int symbol = level - base, length;
        if (symbol < (3 << s.rice))
            rate += ((symbol >> s.rice) + 1 + s.rice) << 15;
        else
        {
            length = s.rice;
            symbol -= 3 << s.rice;
            while (symbol >= (1 << length)) symbol -= 1 << length++;
            rate += (3 + length + 1 - s.rice + length) << 15;
        }
        if (s.nG1 < 8)
        {
            rate += bitsOf(b, HAVOC_RDOQ_CTX_GREATER1 + g1, 1);
            if (s.nG2 < 1) rate += bitsOf(b, HAVOC_RDOQ_CTX_GREATER2 + g2, 1);
        }
    }
    else if (level == 1)
        rate += bitsOf(b, HAVOC_RDOQ_CTX_GREATER1 + g1, 0);
    else if (level == 2)
        rate += bitsOf(b, HAVOC_RDOQ_CTX_GREATER1 + g1, 1) + bitsOf(b, HAVOC_RDOQ_CTX_GREATER2 + g2, 0);
    return rate;
}

// Rdoq.cpp:819-885 getLevelRate
__device__ __forceinline__ int levelRate(const Block &b, int level, int g1, int g2, const LevelState &s)
{
    int rate = 0;
    const int base = baseLevel(s);
    if (level >= base)
    {
        int symbol = level - base;
        const int maxVlc = (0x4e2e1a0e07ull >> (8 * s.rice)) & 0xff;           // 7, 14, 26, 46, 78
        const int prefixMax = 8 - s.rice;                                      // 8, 7, 6, 5, 4
        if (symbol > maxVlc)
        {
            const int rest = symbol - maxVlc;
            int egs = 1;
            for (int top = 2; rest >= top; top <<= 1) egs += 2;
            rate += egs << 15;
            symbol = maxVlc + 1;
        }
        rate += (min(symbol >> (s.rice + 1), prefixMax) + s.rice) << 15;
        if (s.nG1 < 8)
        {
            rate += bitsOf(b, HAVOC_RDOQ_CTX_GREATER1 + g1, 1);
            if (s.nG2 < 1) rate += bitsOf(b, HAVOC_RDOQ_CTX_GREATER2 + g2, 1);
        }
    }
    else if (level == 1)
        rate += bitsOf(b, HAVOC_RDOQ_CTX_GREATER1 + g1, 0);
    else if (level == 2)
        rate += bitsOf(b, HAVOC_RDOQ_CTX_GREATER1 + g1, 1) + bitsOf(b, HAVOC_RDOQ_CTX_GREATER2 + g2, 0);
    return rate;
}

// Rdoq.cpp:517-603 getCoeffSigCtxInc
template <int LOG2>
__device__ __forceinline__ int sigCtx(int neighbours, int scanIdx, int x, int y, int cIdx)
{
    int inc;
    if (LOG2 == 2)
        inc = (0x8877886654325410ull >> (4 * ((y << 2) + x))) & 15;      // 0 1 4 5 / 2 3 4 5 / 6 6 8 8 / 7 7 8 8
    else if (x + y == 0)
        inc = 0;
    else
    {
        const int xp = x & 3, yp = y & 3;
        if (neighbours == 0) inc = xp + yp == 0 ? 2 : (xp + yp < 3 ? 1 : 0);
        else if (neighbours == 1) inc = yp == 0 ? 2 : (yp == 1 ? 1 : 0);
        else if (neighbours == 2) inc = xp == 0 ? 2 : (xp == 1 ? 1 : 0);
        else inc = 2;
        if (cIdx == 0)
        {
            if ((x >> 2) + (y >> 2) > 0) inc += 3;
            inc += LOG2 == 3 ? (scanIdx == 0 ? 9 : 15) : 21;
        }
        else
            inc += LOG2 == 3 ? 9 : 12;
    }
    return cIdx == 0 ? inc : 27 + inc;
}

// per-coefficient results of pass 2, [coefficient][group lane] so that a wavefront's accesses are conflict free
struct Records
{
    int64_t costCoded[16 * kGroups];   // m_rdCostCoeff
    int64_t costSig[16 * kGroups];     // m_rateCostCoeffSig
    int32_t rateUp[16 * kGroups], rateDown[16 * kGroups], sigDelta[16 * kGroups], deltaU[16 * kGroups];
    int16_t kept[16 * kGroups];
};

struct GroupResult
{
    int64_t cost;      // this group's contribution to the block's running RD cost after steps 1 and 2
    int64_t sigCost;   // lambda * bits of its coded_sub_block_flag as coded (m_rateCostCgSig)
    int64_t dist0;     // distortion of the group with every level zero
    int coded;         // coded_sub_block_flag after step 2
    int carry;         // the group ended with greater1CtxIdx == 0
};

// Steps 1 and 2 of runQuantisation for one coefficient group (Rdoq.cpp:83-298) under a given case.
//   g, gx, gy : the group's scan index and coordinates;  firstPos : scan position of the first non-zero rounded level (-1: none)
template <int LOG2, bool RECORD>
__device__ __forceinline__ GroupResult processGroup(const Block &b, int g, int gx, int gy, int firstPos, int caseBits, Records *rec, int lane)
{
    constexpr int size = 1 << LOG2;
    GroupResult r = {0, 0, 0, 0, 0};
    const int firstGroup = firstPos >> 4;                 // -1 when the block quantises to zero
    const int neighbours = caseBits & 3;
    const bool active = firstPos >= 0 && g <= firstGroup;
    LevelState st;
    st.c1 = 1;
    st.nG1 = st.nG2 = st.rice = 0;
    st.ctxSet = g == firstGroup ? ((firstPos < 16 || b.cIdx) ? 0 : 2) : ((g == 0 || b.cIdx) ? 0 : 2) + (caseBits >> 2);
    int nonZeroAbovePos0 = 0;
    int64_t gSig = 0, gSigPos0 = 0, gCoded = 0, gDist0 = 0;
    uint32_t keptMask = 0;

    for (int i = 15; i >= 0; --i)
    {
        const int nib = (int)(b.scan4 >> (4 * i)) & 15, x = (gx << 2) + (nib & 3), y = (gy << 2) + (nib >> 2);
        const int a = abs(coefAt(b, x, y));
        const int64_t dist0 = (int64_t)(a * a) * b.distScale;
        r.dist0 += dist0;
        const int sp = g * 16 + i;
        if (!active || sp > firstPos)
        {
            r.cost += dist0;
            if (RECORD)
            {
                rec->costCoded[i * kGroups + lane] = 0;
                rec->costSig[i * kGroups + lane] = 0;
                rec->rateUp[i * kGroups + lane] = rec->rateDown[i * kGroups + lane] = rec->sigDelta[i * kGroups + lane] = rec->deltaU[i * kGroups + lane] = 0;
                rec->kept[i * kGroups + lane] = 0;
            }
            continue;
        }
        const int scaled = a * b.quantScale;
        const int level = (scaled + (1 << (b.quantShift - 1))) >> b.quantShift;
        const bool first = sp == firstPos;
        const int g1 = 4 * st.ctxSet + st.c1 + (b.cIdx ? 16 : 0), g2 = st.ctxSet + (b.cIdx ? 4 : 0);
        const int sc = HAVOC_RDOQ_CTX_SIG + sigCtx<LOG2>(neighbours, b.scanIdx, x, y, b.cIdx);

        // Rdoq.cpp:456-515 getAdjustedQuantLevel
        int64_t costCoded, costSig = 0, sigOne = 0;
        int kept = 0;
        bool decide = true;
        if (!first && level < 3)
        {
            costSig = b.lambda * bitsOf(b, sc, 0);
            costCoded = dist0 + costSig;
            decide = level != 0;
        }
        else
            costCoded = INT64_MAX;
        if (decide)
        {
            if (!first) sigOne = b.lambda * bitsOf(b, sc, 1);
            for (int l = level, lowest = level > 1 ? level - 1 : 1; l >= lowest; --l)
            {
                const int rebuilt = clip16((clip16(l) * b.invScale + b.invOffset) >> b.invShift);   // Rdoq.h:137-142
                const int32_t err = a - rebuilt;
                const int64_t cost = (int64_t)(int32_t)((uint32_t)err * (uint32_t)err) * b.distScale + b.lambda * levelBits(b, l, g1, g2, st) + sigOne;
                if (cost < costCoded)
                {
                    kept = l;
                    costCoded = cost;
                    costSig = sigOne;
                }
            }
        }
        if (RECORD)
        {
            int up, down = 0;
            if (kept > 0)
            {
                const int now = levelRate(b, kept, g1, g2, st);
                up = levelRate(b, kept + 1, g1, g2, st) - now;
                down = levelRate(b, kept - 1, g1, g2, st) - now;
            }
            else
                up = bitsOf(b, HAVOC_RDOQ_CTX_GREATER1 + g1, 0);
            rec->costCoded[i * kGroups + lane] = costCoded;
            rec->costSig[i * kGroups + lane] = costSig;
            rec->rateUp[i * kGroups + lane] = up;
            rec->rateDown[i * kGroups + lane] = down;
            rec->sigDelta[i * kGroups + lane] = first ? 0 : bitsOf(b, sc, 1) - bitsOf(b, sc, 0);
            rec->deltaU[i * kGroups + lane] = (scaled - (kept << b.quantShift)) >> (b.quantShift - 8);
            rec->kept[i * kGroups + lane] = (int16_t)kept;
        }
        r.cost += costCoded;
        // Rdoq.cpp:773-800 updateEntropyCodingEngine (the per-group reset is the initialisation above)
        if (kept >= baseLevel(st) && kept > 3 * (1 << st.rice)) st.rice = min(st.rice + 1, 4);
        if (kept >= 1) st.nG1++;
        if (kept > 1)
        {
            st.c1 = 0;
            st.nG2++;
        }
        else if (st.c1 < 3 && st.c1 > 0 && kept)
            st.c1++;
        gSig += costSig;
        if (i == 0) gSigPos0 = costSig;
        if ((int16_t)kept)
        {
            keptMask |= 1u << i;
            gCoded += costCoded - costSig;
            gDist0 += dist0;
            if (i) nonZeroAbovePos0++;
        }
    }
    r.carry = st.c1 == 0;
    r.coded = keptMask != 0;
    if (!active) return r;
    if (g == 0)
    {
        r.coded = 1;
        return r;
    }
    // step 2 (Rdoq.cpp:196-297)
    const int flagCtx = HAVOC_RDOQ_CTX_CSBF + (b.cIdx ? 2 : 0) + (neighbours ? 1 : 0);
    const int64_t zero = b.lambda * bitsOf(b, flagCtx, 0);
    if (!r.coded)
    {
        r.cost += zero - gSig;
        r.sigCost = zero;
    }
    else if (g < firstGroup)
    {
        if (nonZeroAbovePos0 == 0)
        {
            r.cost -= gSigPos0;
            gSig -= gSigPos0;
        }
        const int64_t one = b.lambda * bitsOf(b, flagCtx, 1);
        if (zero + gDist0 - gCoded - gSig < one)
        {
            r.coded = 0;
            r.cost += zero + gDist0 - gCoded - gSig;
            r.sigCost = zero;
            if (RECORD)
                for (int i = 15; i >= 0; --i)
                    if (keptMask >> i & 1)
                    {
                        const int nib = (int)(b.scan4 >> (4 * i)) & 15, x = (gx << 2) + (nib & 3), y = (gy << 2) + (nib >> 2);
                        const int a = abs(coefAt(b, x, y));
                        rec->kept[i * kGroups + lane] = 0;
                        rec->costCoded[i * kGroups + lane] = (int64_t)(a * a) * b.distScale;
                        rec->costSig[i * kGroups + lane] = 0;
                    }
        }
        else
        {
            r.cost += one;
            r.sigCost = one;
        }
    }
    return r;
}

__device__ __forceinline__ int64_t shflXor64(int64_t v, int m)
{
    const int lo = __shfl_xor((int)(uint32_t)v, m), hi = __shfl_xor((int)(v >> 32), m);
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}
__device__ __forceinline__ int64_t shflDown64(int64_t v, int d)
{
    const int lo = __shfl_down((int)(uint32_t)v, d), hi = __shfl_down((int)(v >> 32), d);
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

// Rdoq.cpp:887-1023 signDataHiding for one group: the levels live in rec->kept[.][lane] (signed by now), scan order
__device__ __forceinline__ void hideSigns(Records *rec, int lane, const Block &b, int gx, int gy, int size, bool lastGroup, int factor)
{
    int first = 16, last = -1, sum = 0;
    for (int i = 0; i < 16; ++i)
    {
        const int v = rec->kept[i * kGroups + lane];
        sum += v;
        if (v)
        {
            last = i;
            if (first == 16) first = i;
        }
    }
    if (last - first < 4) return;
    const int signbit = rec->kept[first * kGroups + lane] > 0 ? 0 : 1;
    if (signbit == (sum & 1)) return;
    int minCost = INT32_MAX, cost = INT32_MAX, minIdx = -1, finalChange = 0, change = 0;
    for (int i = lastGroup ? last : 15; i >= 0; --i)
    {
        const int k = i * kGroups + lane, v = rec->kept[k], du = rec->deltaU[k];
        if (v != 0)
        {
            const int up = factor * -du + rec->rateUp[k];
            int down = factor * du + rec->rateDown[k] - (abs(v) == 1 ? (1 << 15) + rec->sigDelta[k] : 0);
            if (lastGroup && last == i && abs(v) == 1) down -= 4 << 15;
            if (up < down)
            {
                cost = up;
                change = 1;
            }
            else
            {
                change = -1;
                cost = (i == first && abs(v) == 1) ? INT32_MAX : down;
            }
        }
        else
        {
            cost = factor * -abs(du) + (1 << 15) + rec->rateUp[k] + rec->sigDelta[k];
            change = 1;
            if (i < first)
            {
                const int nib = (int)(b.scan4 >> (4 * i)) & 15;
                if ((coefAt(b, (gx << 2) + (nib & 3), (gy << 2) + (nib >> 2)) >= 0 ? 0 : 1) != signbit) cost = INT32_MAX;
            }
        }
        if (cost < minCost)
        {
            minCost = cost;
            finalChange = change;
            minIdx = i;
        }
    }
    const int k = minIdx * kGroups + lane, v = rec->kept[k];
    if (v == 32767 || v == -32768) finalChange = -1;
    const int nib = (int)(b.scan4 >> (4 * minIdx)) & 15;
    const bool positive = coefAt(b, (gx << 2) + (nib & 3), (gy << 2) + (nib >> 2)) >= 0;
    rec->kept[k] = (int16_t)(positive ? v + finalChange : v - finalChange);
}

struct RdoqShared
{
    Records rec;
    int32_t bits[128];
    int32_t lastRate[kRdoqThreads];       // [block][x | y][coordinate]: bits of last_sig_coeff_{x,y}_prefix + suffix
    int16_t src[16 * kGroups], dst[16 * kGroups];
    uint8_t states[kGroups * HAVOC_RDOQ_CTX_BYTES];
    uint8_t caseFlags[kCases][kGroups];   // pass 1: coded | carry << 1
    uint8_t chosen[kGroups];              // resolve: the case that applies to each group lane
    uint8_t coded[kGroups];               // resolve: coded_sub_block_flag by raster group position within the block
    int firstPos[kGroups];
};

// LOG2 = log2 of the transform size; a workgroup holds 64 >> (2 * LOG2 - 4) blocks
template <int LOG2>
__global__ __launch_bounds__(kRdoqThreads) void k_rdoq(int16_t *__restrict__ dstAll, const int16_t *__restrict__ srcAll, const uint8_t *__restrict__ statesAll,
                                                       const RdoqJob *__restrict__ jobs, int njobs, int32_t *__restrict__ cbfOut, int bitDepth, int stages)
{
    constexpr int size = 1 << LOG2, n = size * size, G = n >> 4, T = kGroups / G, log2G = 2 * LOG2 - 4, gw = size >> 2;
    __shared__ RdoqShared sh;
    const int tid = threadIdx.x, lane = tid & (kGroups - 1), caseBits = tid >> 6;
    const int tl = lane >> log2G, g = lane & (G - 1);        // block within the workgroup, group scan index within the block
    const int tu = blockIdx.x * T + tl;
    const bool valid = tu < njobs;

    // ---- stage in: bit table, states, coefficients ----
    if (tid < 128) sh.bits[tid] = kEntropyBits[tid];
    for (int k = tid; k < T * HAVOC_RDOQ_CTX_BYTES; k += kRdoqThreads)
    {
        const int t = blockIdx.x * T + (k >> 7);
        sh.states[k] = t < njobs ? statesAll[(long)jobs[t].ctx_index * HAVOC_RDOQ_CTX_BYTES + (k & 127)] : 0;
    }
    for (int k = tid; k < 16 * kGroups; k += kRdoqThreads)
    {
        const int t = blockIdx.x * T + (k >> (2 * LOG2));
        sh.src[k] = t < njobs ? srcAll[(long)jobs[t].src_off + (k & (n - 1))] : (int16_t)0;
    }
    if (tid < kGroups) sh.firstPos[tid] = -1;

    RdoqJob job = jobs[valid ? tu : 0];
    Block b;
    b.states = sh.states + tl * HAVOC_RDOQ_CTX_BYTES;
    b.bits = sh.bits;
    b.src = sh.src + tl * n;
    b.stateStride = 1;
    b.coefMask = ~0;
    b.coefRow = size;
    b.coefCol = 1;
    b.lambda = job.lambda_q16;
    {   // Rdoq.h:163-187
        const int transformShift = 15 - bitDepth - LOG2;
        b.distScale = 1 << (15 - 2 * transformShift - 2 * (bitDepth - 8) + 16);
        b.invShift = 6 - transformShift;
        b.invOffset = 1 << (b.invShift - 1);
    }
    b.quantScale = job.quant_scale;
    b.quantShift = job.quant_shift;
    b.invScale = job.inv_scale;
    b.cIdx = job.c_idx;
    b.scanIdx = job.scan_idx;
    b.scan4 = job.scan_idx == 0 ? scan4Nibbles(0) : (job.scan_idx == 1 ? scan4Nibbles(1) : scan4Nibbles(2));
    int gx = 0, gy = 0;
    if (G > 1) scanXy(gw, b.scanIdx, g, gx, gy);
    __syncthreads();

    // ---- first non-zero rounded level (Rdoq.cpp:118-126), and the last-position bit counts (Rdoq.cpp:706-771) ----
    if (caseBits == 0)
    {
        int top = -1;
        for (int i = 15; i >= 0 && top < 0; --i)
        {
            const int nib = (int)(b.scan4 >> (4 * i)) & 15;
            const int a = abs(coefAt(b, (gx << 2) + (nib & 3), (gy << 2) + (nib >> 2)));
            if (((a * b.quantScale + (1 << (b.quantShift - 1))) >> b.quantShift) > 0) top = g * 16 + i;
        }
        if (top >= 0 && valid) atomicMax(&sh.firstPos[tl], top);
    }
    if (tid < T * 2 * size)
    {
        const int t = tid / (2 * size), axis = (tid / size) & 1, c = tid & (size - 1);
        const RdoqJob &jt = jobs[min((int)blockIdx.x * T + t, njobs - 1)];
        const uint8_t *st = sh.states + t * HAVOC_RDOQ_CTX_BYTES + (axis ? HAVOC_RDOQ_CTX_LAST_Y : HAVOC_RDOQ_CTX_LAST_X);
        const int len = c < 4 ? c : (c < 8 ? 4 + ((c - 4) >> 1) : (c < 16 ? 6 + ((c - 8) >> 2) : 8 + ((c - 16) >> 3)));   // 0 1 2 3 4 4 5 5 6 6 6 6 7 ...
        const int offset = jt.c_idx ? 15 : 3 * (LOG2 - 2) + ((LOG2 - 1) >> 2), shift = jt.c_idx ? LOG2 - 2 : (LOG2 + 1) >> 2;
        int32_t rate = 0;
        for (int i = 0; i <= len && i < 9; ++i) rate += sh.bits[(st[min(max((i >> shift) + offset, 0), 17)] >> 1) ^ (i < len ? 1 : 0)];
        if (len > 3) rate += 32768 * ((len - 2) >> 1);
        sh.lastRate[tid] = rate;
    }
    __syncthreads();
    const int firstPos = sh.firstPos[tl], firstGroup = firstPos >> 4;

    // ---- pass 1: every group under every case ----
    {
        const GroupResult r = processGroup<LOG2, false>(b, g, gx, gy, firstPos, caseBits, nullptr, lane);
        sh.caseFlags[caseBits][lane] = (uint8_t)(r.coded | r.carry << 1);
    }
    __syncthreads();
    if (stages == 1) return;

    // ---- resolve: follow the three bits through the groups in reverse scan order ----
    if (tid < T)
    {
        const int fp = sh.firstPos[tid], fg = fp >> 4;
        const int scanIdx = jobs[min((int)blockIdx.x * T + tid, njobs - 1)].scan_idx;
        uint8_t *coded = sh.coded + tid * G;
        for (int k = 0; k < G; ++k) coded[k] = 0;
        int carry = 0;
        for (int k = fg; k >= 0; --k)
        {
            int x = 0, y = 0;
            if (G > 1) scanXy(gw, scanIdx, k, x, y);
            const int right = x < gw - 1 ? coded[y * gw + x + 1] : 0, below = y < gw - 1 ? coded[(y + 1) * gw + x] : 0;
            const int c = right | below << 1 | (k == fg ? 0 : carry) << 2;
            const int f = sh.caseFlags[c][tid * G + k];
            sh.chosen[tid * G + k] = (uint8_t)c;
            coded[y * gw + x] = f & 1;
            carry = f >> 1;
        }
        for (int k = G - 1; k > fg; --k) sh.chosen[tid * G + k] = 0;
    }
    __syncthreads();
    if (stages == 2) return;

    // ---- pass 2 and the tail, one wavefront: lanes = groups ----
    if (tid < kGroups)
    {
        const GroupResult r = processGroup<LOG2, true>(b, g, gx, gy, firstPos, sh.chosen[lane], &sh.rec, lane);
        const bool inScope = firstPos >= 0 && g <= firstGroup;
        if (stages == 3) return;

        // running-cost delta of this group in the last-position search (Rdoq.cpp:356-399 without the early exit) and the
        // position of its highest level > 1
        int64_t delta = 0;
        int big = -1;
        if (inScope)
        {
            delta = -r.sigCost;
            if (r.coded)
                for (int i = 15; i >= 0; --i)
                {
                    const int sp = g * 16 + i, k = i * kGroups + lane;
                    if (sp > firstPos) continue;
                    if (sh.rec.kept[k])
                    {
                        const int nib = (int)(b.scan4 >> (4 * i)) & 15;
                        const int a = abs(coefAt(b, (gx << 2) + (nib & 3), (gy << 2) + (nib >> 2)));
                        delta += (int64_t)(a * a) * b.distScale - sh.rec.costCoded[k];
                        if (sh.rec.kept[k] > 1 && big < 0) big = sp;
                    }
                    else
                        delta -= sh.rec.costSig[k];
                }
        }
        // block-wide sums over the G lanes of the block
        int64_t costTu = r.cost, dist0Total = r.dist0, after = delta;
        int stopPos = big;
#pragma unroll
        for (int m = 1; m < G; m <<= 1)
        {
            costTu += shflXor64(costTu, m);
            dist0Total += shflXor64(dist0Total, m);
            stopPos = max(stopPos, __shfl_xor(stopPos, m));
        }
#pragma unroll
        for (int d = 1; d < G; d <<= 1)      // inclusive suffix sum over the groups with a larger scan index
        {
            const int64_t o = shflDown64(after, d);
            if (g + d < G) after += o;
        }
        const int cbfCtx = (!job.is_intra && b.cIdx == 0) ? HAVOC_RDOQ_CTX_ROOT_CBF : (b.cIdx == 0 ? HAVOC_RDOQ_CTX_CBF_LUMA + 1 : HAVOC_RDOQ_CTX_CBF_CHROMA);
        const int64_t bestNone = dist0Total + b.lambda * bitsOf(b, cbfCtx, 0);
        int64_t running = costTu + b.lambda * bitsOf(b, cbfCtx, 1) + (after - delta) - r.sigCost;

        // candidates of this group
        int64_t best = INT64_MAX;
        int bestPos = -1;
        if (inScope && r.coded)
        {
            const int32_t *lr = sh.lastRate + tl * 2 * size;
            for (int i = 15; i >= 0; --i)
            {
                const int sp = g * 16 + i, k = i * kGroups + lane;
                if (sp > firstPos) continue;
                if (sh.rec.kept[k])
                {
                    const int nib = (int)(b.scan4 >> (4 * i)) & 15, x = (gx << 2) + (nib & 3), y = (gy << 2) + (nib >> 2);
                    const int32_t rate = b.scanIdx == 2 ? lr[y] + lr[size + x] : lr[x] + lr[size + y];
                    const int64_t total = running + b.lambda * rate - sh.rec.costSig[k];
                    if (sp >= stopPos && total < best)
                    {
                        best = total;
                        bestPos = sp;
                    }
                    const int a = abs(coefAt(b, x, y));
                    running += (int64_t)(a * a) * b.distScale - sh.rec.costCoded[k];
                }
                else
                    running -= sh.rec.costSig[k];
            }
        }
#pragma unroll
        for (int m = 1; m < G; m <<= 1)      // minimum cost; among equals the position met first, i.e. the highest
        {
            const int64_t ob = shflXor64(best, m);
            const int op = __shfl_xor(bestPos, m);
            if (ob < best || (ob == best && op > bestPos))
            {
                best = ob;
                bestPos = op;
            }
        }
        const int lastIdx = best < bestNone ? bestPos + 1 : 0;

        // signs, truncation (Rdoq.cpp:418-435)
        int absSum = 0, cbf = 0;
        for (int i = 0; i < 16; ++i)
        {
            const int sp = g * 16 + i, k = i * kGroups + lane;
            int level = sh.rec.kept[k];
            if (sp < lastIdx)
            {
                const int nib = (int)(b.scan4 >> (4 * i)) & 15;
                absSum += level;
                cbf |= level;
                if (coefAt(b, (gx << 2) + (nib & 3), (gy << 2) + (nib >> 2)) < 0) level = -level;
            }
            else
                level = 0;
            sh.rec.kept[k] = (int16_t)level;
        }
#pragma unroll
        for (int m = 1; m < G; m <<= 1)
        {
            absSum += __shfl_xor(absSum, m);
            cbf |= __shfl_xor(cbf, m);
        }
        if (job.sdh && absSum >= 2) hideSigns(&sh.rec, lane, b, gx, gy, size, g == ((lastIdx - 1) >> 4), job.sdh_factor);
        for (int i = 0; i < 16; ++i)
        {
            const int nib = (int)(b.scan4 >> (4 * i)) & 15;
            sh.dst[tl * n + ((gy << 2) + (nib >> 2)) * size + (gx << 2) + (nib & 3)] = sh.rec.kept[i * kGroups + lane];
        }
        if (g == 0 && valid) cbfOut[tu] = firstPos >= 0 ? cbf : 0;
    }
    __syncthreads();
    for (int k = tid; k < 16 * kGroups; k += kRdoqThreads)
    {
        const int t = blockIdx.x * T + (k >> (2 * LOG2));
        if (t < njobs) dstAll[(long)jobs[t].dst_off + (k & (n - 1))] = sh.dst[k];
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// k_rdoq_walk : the work-efficient form.  Transform coefficients after a QP-32 quantiser are sparse (3-5 % of the levels and
// ~15 % of the 4x4 groups of a 32x32 block are non-zero), and a group whose rounded levels are all zero does no more than add
// one flag cost: it keeps no level, leaves the level state alone and is never a candidate for the last position.  So:
//   * lane = transform block, 64 blocks per wavefront; a lane walks ITS block's groups in reverse scan order, hopping over the
//     all-zero ones (a handful of instructions each) and doing the full per-coefficient work (processGroup above, the same code
//     the speculative kernel runs) only on the others.  Nothing is speculated: the neighbours' flags and the carry are known.
//   * the search for the last significant position (Rdoq.cpp:342-399) is streamed: its running cost differs from the block's
//     final cost by a sum of per-coefficient deltas, so the best candidate relative to that final cost can be tracked group by
//     group, committed once the group's keep-or-zero decision is made, and compared with "code nothing" at the very end.
//   * sign-data hiding is applied to each group as it is finished, as if it were not the group holding the last significant
//     coefficient; that one group (known only at the end) is redone.
//   * the per-group record arrays live in LDS, [coefficient][lane]; the pre-pass that finds the non-zero groups (and zero-fills
//     the output) is cooperative: 64 lanes read one 32x32 block's 64 groups (or four 16x16, ...) per step, coalesced.
// ---------------------------------------------------------------------------------------------------------------------
struct WalkShared
{
    Records rec;
    int32_t bits[128];
    uint8_t states[HAVOC_RDOQ_CTX_BYTES][64];     // [context][lane]
    int32_t cumOne[2][10][64];                    // sum over i < k of bits(1, ctx(i)) for last_sig_coeff_{x,y}_prefix
    int32_t zeroBin[2][9][64];                    // bits(0, ctx(k))
    int16_t coef[16][64];                         // the current group's coefficients, raster order within the group
    uint8_t rasterOf[3][64];                      // scan index -> raster group position, per scan type
    uint64_t mask[64];                            // non-zero groups of each block (bit = raster group position)
    int64_t sumSq[64];                            // sum of squared coefficients of each block
    int32_t srcOff[64], dstOff[64], qScale[64], qShift[64];
};

template <int LOG2>
__global__ __launch_bounds__(64) void k_rdoq_walk(int16_t *__restrict__ dstAll, const int16_t *__restrict__ srcAll, const uint8_t *__restrict__ statesAll,
                                                  const RdoqJob *__restrict__ jobs, int njobs, int32_t *__restrict__ cbfOut, int bitDepth)
{
    constexpr int size = 1 << LOG2, G = (size * size) >> 4, log2G = 2 * LOG2 - 4, gw = size >> 2, perStep = 64 / G;
    __shared__ WalkShared sh;
    const int lane = threadIdx.x, blk = blockIdx.x * 64 + lane;
    const bool valid = blk < njobs;
    const RdoqJob job = jobs[valid ? blk : njobs - 1];

    // ---- stage in ----
    sh.bits[lane] = kEntropyBits[lane];
    sh.bits[64 + lane] = kEntropyBits[64 + lane];
    {
        const uint32_t *st = reinterpret_cast<const uint32_t *>(statesAll + (long)job.ctx_index * HAVOC_RDOQ_CTX_BYTES);
        for (int k = 0; k < HAVOC_RDOQ_CTX_BYTES / 4; ++k)
        {
            const uint32_t v = st[k];
            sh.states[4 * k][lane] = (uint8_t)v;
            sh.states[4 * k + 1][lane] = (uint8_t)(v >> 8);
            sh.states[4 * k + 2][lane] = (uint8_t)(v >> 16);
            sh.states[4 * k + 3][lane] = (uint8_t)(v >> 24);
        }
    }
    sh.srcOff[lane] = job.src_off;
    sh.dstOff[lane] = job.dst_off;
    sh.qScale[lane] = job.quant_scale;
    sh.qShift[lane] = job.quant_shift;
    if (lane < G)
        for (int t = 0; t < 3; ++t)
        {
            int x = 0, y = 0;
            if (G > 1) scanXy(gw, t, lane, x, y);
            sh.rasterOf[t][lane] = (uint8_t)(y * gw + x);
        }
    __syncthreads();

    // ---- pre-pass: which groups hold a non-zero rounded level, the blocks' energy, zeros into the output ----
    {
        const int sub = lane >> log2G, pos = lane & (G - 1), px = pos & (gw - 1), py = pos / gw;
        for (int step = 0; step < G; ++step)        // G steps of 64 / G blocks = 64 blocks
        {
            const int bl = step * perStep + sub;
            const bool have = blockIdx.x * 64 + bl < njobs;
            const int16_t *p = srcAll + (long)sh.srcOff[bl] + (py * 4) * size + px * 4;
            int16_t *q = dstAll + (long)sh.dstOff[bl] + (py * 4) * size + px * 4;
            const int qs = sh.qScale[bl], rnd = 1 << (sh.qShift[bl] - 1), qsh = sh.qShift[bl];
            bool nz = false;
            uint32_t lo = 0, hi = 0;
            if (have)
                for (int r = 0; r < 4; ++r)
                {
                    const u32x2 v = ld8(p + r * size);
                    st8(q + r * size, u32x2{0, 0});
                    const int c[4] = {(int16_t)v.x, (int16_t)(v.x >> 16), (int16_t)v.y, (int16_t)(v.y >> 16)};
                    for (int k = 0; k < 4; ++k)
                    {
                        const uint32_t a = (uint32_t)abs(c[k]);
                        nz |= (int)((a * qs + rnd) >> qsh) > 0;
                        lo += (a * a) & 0xffff;
                        hi += (a * a) >> 16;
                    }
                }
            const uint64_t m = __ballot(nz);
            const int slo = group_sum<G>((int)lo), shi = group_sum<G>((int)hi);
            if (pos == 0)
            {
                sh.mask[bl] = G == 64 ? m : (m >> (lane & ~(G - 1))) & ((1ull << (G & 63)) - 1);
                sh.sumSq[bl] = ((int64_t)shi << 16) + slo;
            }
        }
    }
    __syncthreads();

    // ---- per-lane set-up ----
    Block b;
    b.states = &sh.states[0][lane];
    b.stateStride = 64;
    b.bits = sh.bits;
    b.src = &sh.coef[0][lane];
    b.coefMask = 3;
    b.coefRow = 4 * 64;
    b.coefCol = 64;
    b.lambda = job.lambda_q16;
    const int transformShift = 15 - bitDepth - LOG2, distShift = 15 - 2 * transformShift - 2 * (bitDepth - 8) + 16;   // Rdoq.h:163-187
    b.distScale = 1 << distShift;
    b.invShift = 6 - transformShift;
    b.invOffset = 1 << (b.invShift - 1);
    b.quantScale = job.quant_scale;
    b.quantShift = job.quant_shift;
    b.invScale = job.inv_scale;
    b.cIdx = job.c_idx;
    b.scanIdx = job.scan_idx;
    b.scan4 = job.scan_idx == 0 ? scan4Nibbles(0) : (job.scan_idx == 1 ? scan4Nibbles(1) : scan4Nibbles(2));
    const uint8_t *rasterOf = sh.rasterOf[job.scan_idx < 3 ? job.scan_idx : 0];
    const int16_t *src = srcAll + job.src_off;
    int16_t *dst = dstAll + job.dst_off;
    for (int axis = 0; axis < 2; ++axis)      // Rdoq.cpp:706-771, as prefix sums over the bins
    {
        const int base = axis ? HAVOC_RDOQ_CTX_LAST_Y : HAVOC_RDOQ_CTX_LAST_X;
        const int offset = b.cIdx ? 15 : 3 * (LOG2 - 2) + ((LOG2 - 1) >> 2), shift = b.cIdx ? LOG2 - 2 : (LOG2 + 1) >> 2;
        int32_t run = 0;
        for (int i = 0; i < 10; ++i)
        {
            const int ctx = base + min(max((i >> shift) + offset, 0), 17);
            sh.cumOne[axis][i][lane] = run;
            if (i < 9)
            {
                sh.zeroBin[axis][i][lane] = bitsOf(b, ctx, 0);
                run += bitsOf(b, ctx, 1);
            }
        }
    }
    auto lastRate = [&](int axis, int c) -> int32_t {
        const int len = c < 4 ? c : (c < 8 ? 4 + ((c - 4) >> 1) : (c < 16 ? 6 + ((c - 8) >> 2) : 8 + ((c - 16) >> 3)));
        return sh.cumOne[axis][len][lane] + (len < 9 ? sh.zeroBin[axis][len][lane] : 0) + (len > 3 ? 32768 * ((len - 2) >> 1) : 0);
    };
    auto loadGroup = [&](int gx, int gy) {
        for (int r = 0; r < 4; ++r)
        {
            const u32x2 v = ld8(src + ((gy << 2) + r) * size + (gx << 2));
            sh.coef[4 * r][lane] = (int16_t)v.x;
            sh.coef[4 * r + 1][lane] = (int16_t)(v.x >> 16);
            sh.coef[4 * r + 2][lane] = (int16_t)v.y;
            sh.coef[4 * r + 3][lane] = (int16_t)(v.y >> 16);
        }
    };
    auto storeGroup = [&](int gx, int gy) {      // rec.kept (scan order, signed) -> the output block
        for (int i = 0; i < 16; ++i)
        {
            const int nib = (int)(b.scan4 >> (4 * i)) & 15;
            sh.coef[nib][lane] = sh.rec.kept[i * kGroups + lane];      // coef doubles as the raster staging area
        }
        for (int r = 0; r < 4; ++r)
        {
            u32x2 o;
            o.x = (uint16_t)sh.coef[4 * r][lane] | (uint32_t)(uint16_t)sh.coef[4 * r + 1][lane] << 16;
            o.y = (uint16_t)sh.coef[4 * r + 2][lane] | (uint32_t)(uint16_t)sh.coef[4 * r + 3][lane] << 16;
            st8(dst + ((gy << 2) + r) * size + (gx << 2), o);
        }
    };
    auto caseOf = [&](uint64_t coded, int gx, int gy, int carry) {
        const int p = gy * gw + gx;
        const int right = gx < gw - 1 ? (int)(coded >> (p + 1)) & 1 : 0, below = gy < gw - 1 ? (int)(coded >> (p + gw)) & 1 : 0;
        return right | below << 1 | carry << 2;
    };

    uint64_t nz = valid ? sh.mask[lane] : 0, coded = 0, carries = 0;
    int g = G - 1;
    while (g >= 0 && !((nz >> rasterOf[g]) & 1)) --g;
    const int firstGroup = g;
    int firstPos = -1;
    if (firstGroup >= 0)
    {
        const int p = rasterOf[firstGroup], gx = p & (gw - 1), gy = p / gw;
        loadGroup(gx, gy);
        for (int i = 15; i >= 0 && firstPos < 0; --i)
        {
            const int nib = (int)(b.scan4 >> (4 * i)) & 15;
            const int a = abs((int)sh.coef[nib][lane]);
            if (((a * b.quantScale + (1 << (b.quantShift - 1))) >> b.quantShift) > 0) firstPos = firstGroup * 16 + i;
        }
        nz |= 1;      // the DC group is always walked in full (it is coded whatever its levels, Rdoq.cpp:291-295)
    }

    // running state of the walk
    int64_t costTu = 0, heavyDist0 = 0;     // costTu: everything but the energy of the coefficients outside the walked groups
    int64_t rel = 0;                        // running cost of the last-position search relative to its start
    int64_t bestRel = INT64_MAX;
    int bestPos = -1, orSince = 0, carry = 0;
    bool stopped = false;

    while (true)
    {
        while (g >= 0 && !((nz >> rasterOf[g]) & 1))      // all-zero groups: one flag cost each (Rdoq.cpp:200-210)
        {
            const int p = rasterOf[g], gx = p & (gw - 1), gy = p / gw;
            const int c = caseOf(coded, gx, gy, 0);
            const int64_t zero = b.lambda * bitsOf(b, HAVOC_RDOQ_CTX_CSBF + (b.cIdx ? 2 : 0) + ((c & 3) ? 1 : 0), 0);
            costTu += zero;
            rel -= zero;
            carry = 0;
            --g;
        }
        if (__ballot(g >= 0) == 0) break;
        if (g >= 0)
        {
            const int p = rasterOf[g], gx = p & (gw - 1), gy = p / gw;
            const int c = caseOf(coded, gx, gy, carry);
            loadGroup(gx, gy);
            const GroupResult r = processGroup<LOG2, true>(b, g, gx, gy, firstPos, c, &sh.rec, lane);
            costTu += r.cost;
            heavyDist0 += r.dist0;
            coded |= (uint64_t)r.coded << p;
            carries |= (uint64_t)carry << g;
            carry = r.carry;
            rel -= r.sigCost;
            if (r.coded)
            {
                // candidates of this group (Rdoq.cpp:356-399), relative to `rel`
                int64_t q = 0, localBest = INT64_MAX;
                int localPos = -1, localOr = 0, groupOr = 0;
                bool localStop = false;
                for (int i = 15; i >= 0; --i)
                {
                    const int sp = g * 16 + i, k = i * kGroups + lane;
                    if (sp > firstPos) continue;
                    const int kept = sh.rec.kept[k];
                    if (kept)
                    {
                        const int nib = (int)(b.scan4 >> (4 * i)) & 15, x = (gx << 2) + (nib & 3), y = (gy << 2) + (nib >> 2);
                        const int32_t rate = b.scanIdx == 2 ? lastRate(0, y) + lastRate(1, x) : lastRate(0, x) + lastRate(1, y);
                        const int64_t total = q + b.lambda * rate - sh.rec.costSig[k];
                        groupOr |= kept;
                        if (!localStop && total < localBest)
                        {
                            localBest = total;
                            localPos = sp;
                            localOr = 0;
                        }
                        localOr |= kept;
                        if (kept > 1) localStop = true;
                        const int a = abs(coefAt(b, x, y));
                        q += (int64_t)(a * a) * b.distScale - sh.rec.costCoded[k];
                    }
                    else
                        q -= sh.rec.costSig[k];
                }
                if (!stopped && localPos >= 0 && rel + localBest < bestRel)
                {
                    bestRel = rel + localBest;
                    bestPos = localPos;
                    orSince = localOr;
                }
                else
                    orSince |= groupOr;
                stopped |= localStop;
                rel += q;

                // signs, sign-data hiding as for a group below the last one (Rdoq.cpp:418-441, :887-1023), out
                for (int i = 0; i < 16; ++i)
                {
                    const int nib = (int)(b.scan4 >> (4 * i)) & 15, k = i * kGroups + lane;
                    if (sh.coef[nib][lane] < 0) sh.rec.kept[k] = (int16_t)-sh.rec.kept[k];
                }
                if (job.sdh) hideSigns(&sh.rec, lane, b, gx, gy, size, false, job.sdh_factor);
                storeGroup(gx, gy);
            }
            --g;
        }
    }

    // ---- the block's verdict (Rdoq.cpp:307-341, :401-441) ----
    int cbf = 0;
    if (firstPos >= 0)
    {
        const int cbfCtx = (!job.is_intra && b.cIdx == 0) ? HAVOC_RDOQ_CTX_ROOT_CBF : (b.cIdx == 0 ? HAVOC_RDOQ_CTX_CBF_LUMA + 1 : HAVOC_RDOQ_CTX_CBF_CHROMA);
        const int64_t dist0Total = sh.sumSq[lane] << distShift;
        const int64_t bestNone = dist0Total + b.lambda * bitsOf(b, cbfCtx, 0);
        const int64_t start = (dist0Total - heavyDist0) + costTu + b.lambda * bitsOf(b, cbfCtx, 1);
        const int lastIdx = (bestPos >= 0 && start + bestRel < bestNone) ? bestPos + 1 : 0;
        cbf = lastIdx ? orSince : 0;
        const int lastGroup = (lastIdx - 1) >> 4;      // -1: nothing is coded
        // groups above the last one were written as if coded: clear them
        for (int k = firstGroup; k > lastGroup; --k)
        {
            const int p = rasterOf[k];
            if ((coded >> p) & 1)
                for (int r = 0; r < 4; ++r) st8(dst + (((p / gw) << 2) + r) * size + ((p & (gw - 1)) << 2), u32x2{0, 0});
        }
        // the group holding the last significant coefficient: levels again, truncated, hidden with the last-group rules
        if (lastGroup >= 0 && (lastIdx & 15 || job.sdh))
        {
            const int p = rasterOf[lastGroup], gx = p & (gw - 1), gy = p / gw;
            const int c = caseOf(coded, gx, gy, (int)(carries >> lastGroup) & 1);
            loadGroup(gx, gy);
            processGroup<LOG2, true>(b, lastGroup, gx, gy, firstPos, c, &sh.rec, lane);
            for (int i = 0; i < 16; ++i)
            {
                const int nib = (int)(b.scan4 >> (4 * i)) & 15, k = i * kGroups + lane;
                int v = lastGroup * 16 + i < lastIdx ? sh.rec.kept[k] : 0;
                if (sh.coef[nib][lane] < 0) v = -v;
                sh.rec.kept[k] = (int16_t)v;
            }
            if (job.sdh) hideSigns(&sh.rec, lane, b, gx, gy, size, true, job.sdh_factor);
            storeGroup(gx, gy);
        }
    }
    if (valid) cbfOut[blk] = cbf;
}
} // namespace

hipError_t launch_rdoq(hipStream_t st, int bitDepth, int log2, int16_t *dst, const int16_t *src, const uint8_t *states, const void *jobs, int njobs, int32_t *cbf)
{
    if (njobs <= 0) return hipSuccess;
    const int perGroup = kGroups >> (2 * log2 - 4), blocks = (njobs + perGroup - 1) / perGroup;
    const RdoqJob *j = static_cast<const RdoqJob *>(jobs);
    static const int stages = getenv("HAVOC_RDOQ_STAGES") ? atoi(getenv("HAVOC_RDOQ_STAGES")) : 0;   // diagnostic: stop after a stage
    static const bool speculative = getenv("HAVOC_RDOQ_KERNEL") && !strcmp(getenv("HAVOC_RDOQ_KERNEL"), "groups");
    if (!speculative)
    {
        const int wgs = (njobs + 63) / 64;
        switch (log2)
        {
        case 2: hipLaunchKernelGGL(k_rdoq_walk<2>, dim3(wgs), dim3(64), 0, st, dst, src, states, j, njobs, cbf, bitDepth); break;
        case 3: hipLaunchKernelGGL(k_rdoq_walk<3>, dim3(wgs), dim3(64), 0, st, dst, src, states, j, njobs, cbf, bitDepth); break;
        case 4: hipLaunchKernelGGL(k_rdoq_walk<4>, dim3(wgs), dim3(64), 0, st, dst, src, states, j, njobs, cbf, bitDepth); break;
        default: hipLaunchKernelGGL(k_rdoq_walk<5>, dim3(wgs), dim3(64), 0, st, dst, src, states, j, njobs, cbf, bitDepth); break;
        }
        return hipGetLastError();
    }
    switch (log2)
    {
    case 2: hipLaunchKernelGGL(k_rdoq<2>, dim3(blocks), dim3(kRdoqThreads), 0, st, dst, src, states, j, njobs, cbf, bitDepth, stages); break;
    case 3: hipLaunchKernelGGL(k_rdoq<3>, dim3(blocks), dim3(kRdoqThreads), 0, st, dst, src, states, j, njobs, cbf, bitDepth, stages); break;
    case 4: hipLaunchKernelGGL(k_rdoq<4>, dim3(blocks), dim3(kRdoqThreads), 0, st, dst, src, states, j, njobs, cbf, bitDepth, stages); break;
    default: hipLaunchKernelGGL(k_rdoq<5>, dim3(blocks), dim3(kRdoqThreads), 0, st, dst, src, states, j, njobs, cbf, bitDepth, stages); break;
    }
    return hipGetLastError();
}

} // namespace havoc_gpu
