// k_rdoq_walk : rate-distortion optimised quantisation of transform blocks.
//
// Reference: turing/Rdoq.cpp:37-454 runQuantisation with its helpers (:456-885), sign-data hiding (:887-1023), the constructor
// turing/Rdoq.h:163-187, the scans of turing/ScanOrder.h:31-95 and the bit-cost table turing/Write.h:413-422; called between the
// forward transform and the reconstruction at turing/Reconstruct.cpp:289-312 (intra) and :794-812 (inter).
//
// The reference walks a block's coefficients one by one in reverse scan order, carrying the entropy coder's level state and
// three running costs.  What couples one 4x4 coefficient group to the rest of the block is small -- whether the groups to its
// right and below ended up coded (they pick the significance contexts, Rdoq.cpp:517-624, and the group flag's context,
// :670-693), whether the previous group in scan order ended with a level > 1 (it bumps the greater-than-one context set,
// Rdoq.cpp:806-816), where the first non-zero rounded level sits, and sums of Q16 integer costs -- but it is a chain: a block is
// sequential group by group.  Integer throughout: Q16 int64 costs, Q16 int32 lambda / distortion scale, Q15 bit counts.
//
// (An earlier form evaluated all 64 groups of a 32x32 block at once under all 8 values of those three bits and resolved the
// chain afterwards: bit exact, but 8x the arithmetic on 85 % zero groups and an LDS footprint that left 3 workgroups per CU; it
// took 1.9 ms per 1080p picture against 0.93 ms for the form below.  profiles/r02_experiments.md has the numbers.)
#include "common.h"
#include "rdoq_work.h"

#include <cstdlib>

namespace havoc_gpu {

// Diagnostic build only (-DHAVOC_RDOQ_TIMING; profiles/micro/rdoq_timing.py): shader-clock cycles a wavefront of the walk kernels spends in each section, summed
// over the wavefronts of every launch since the last reset; read back by havoc_mi355x_debug_rdoq_timing.  The product build has none of this.
#ifdef HAVOC_RDOQ_TIMING
__device__ unsigned long long g_rdoqTiming[32];
struct RdoqTimer
{
    unsigned long long acc[16] = {}, last = __builtin_readcyclecounter();
    __device__ __forceinline__ void mark(int k)
    {
        const unsigned long long now = __builtin_readcyclecounter();
        acc[k] += now - last;
        last = now;
    }
    __device__ __forceinline__ void flush(int lane, int base)
    {
        if (lane != 0) return;
        for (int k = 0; k < 16; ++k)
            if (acc[k]) atomicAdd(&g_rdoqTiming[base + k], acc[k]);
    }
};
#define RT_DECL RdoqTimer rt
#define RT_MARK(k) rt.mark(k)
#define RT_PARAM , RdoqTimer &rt
#define RT_ARG , rt
#define RT_FLUSH(lane, base) rt.flush(lane, base)
#else
#define RT_DECL
#define RT_MARK(k)
#define RT_PARAM
#define RT_ARG
#define RT_FLUSH(lane, base)
#endif

namespace {

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

typedef short s16x2v __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2v __attribute__((ext_vector_type(2)));

static_assert(sizeof(havoc_mi355x_rdoq_job) == sizeof(RdoqJob), "rdoq job layout");

// what a lane knows about its transform block
struct Block
{
    const uint8_t *states;    // this block's 128 context states, `stateStride` apart: a column of the wavefront's LDS copy, or (stride 1) the CTU's snapshot in global memory
    const int32_t *bits;      // LDS: kEntropyBits
    const int32_t *lastBits;  // LDS: this block's [2][NLEN] bits of a last_sig_coeff_{x,y} coordinate whose prefix has k ones (Rdoq.cpp:706-763), `lastStride` apart
    int stateStride, lastStride, lastLen;
    int64_t lambda;
    int distShift;            // distortion scale = 1 << distShift (Q16)
    int quantScale, quantShift, invScale, invShift, invOffset;
    int cIdx, scanIdx;
    uint64_t scan4;           // the 4x4 scan as 16 nibbles x | y << 2
    int64_t csbfZero[2], csbfOne[2];   // lambda * bits of coded_sub_block_flag = 0 / 1 in its two contexts (neighbour coded or not): fixed for the block, asked for per group
    int64_t cbfZero, cbfOne;           // lambda * bits of the block's coded-block flag
    const uint32_t *clsTab;   // LDS: [right / below coded: 4 cases] scan positions of THIS block's scan whose significance context is base + 1 | those at base + 2, << 16
};

struct LevelState { int ctxSet, c1, nG1, nG2, rice; };   // Rdoq.cpp:44-49

__device__ __forceinline__ int32_t bitsOf(const Block &b, int ctx, int bin) { return b.bits[(b.states[ctx * b.stateStride] >> 1) ^ bin]; }
// c ? a : b with both arms already evaluated: the compiler then emits a select.  Written as a nested conditional expression with
// arithmetic in its arms, the same thing becomes a branch per arm -- and a wavefront of 64 blocks takes both sides of every branch,
// plus the exec-mask bookkeeping: the per-level loop of walkGroup lost a quarter of its instructions when its conditionals were
// rewritten this way (profiles/r02_experiments.md)
template <class T>
__device__ __forceinline__ T pick(bool c, T a, T b) { return c ? a : b; }

__device__ __forceinline__ int baseLevel(const LevelState &s) { return pick(s.nG1 < 8, 2 + (int)(s.nG2 < 1), 1); }
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

// The scan pass marks a block's 4x4 groups by RASTER position (a ballot of the lanes that looked at them); the walks go through the groups in SCAN order and ask "the
// next marked group below this one", "how many unmarked groups in between": bit g of a scan-order mask = group number g of the scan (ScanOrder.h:31-95 applied to the
// GW x GW groups).  Done once per block (k_rdoq_order for the sorted sizes, after the in-kernel scan for the others).
template <int GW>
__device__ __forceinline__ void toScanOrder(int scanIdx, uint64_t &m, uint64_t &m2, uint64_t &m3)
{
    if (GW == 1 || scanIdx == 1) return;      // horizontal: raster order
    uint64_t a = 0, b = 0, c = 0;
    int x = 0, y = 0;
    for (int g = 0; g < GW * GW; ++g)
    {
        const int p = y * GW + x;
        a |= ((m >> p) & 1) << g;
        b |= ((m2 >> p) & 1) << g;
        c |= ((m3 >> p) & 1) << g;
        if (scanIdx == 2)      // vertical: x = g / GW, y = g % GW
        {
            if (++y == GW) { y = 0; ++x; }
        }
        else                   // up-right diagonal: along an anti-diagonal x goes up and y down; the next one starts at its bottom-left end
        {
            ++x; --y;
            if (y < 0 || x >= GW)
            {
                const int d = x + y + 1;
                x = d < GW ? 0 : d - GW + 1;
                y = d - x;
            }
        }
    }
    m = a; m2 = b; m3 = c;
}

// Rdoq.cpp:710: number of ones in the prefix of a last-significant coordinate: 0 1 2 3 4 4 5 5 6 6 6 6 7 7 7 7 8 x 8, 9 x 8
__device__ __forceinline__ int lastPrefixLength(int c) { return pick(c < 4, c, pick(c < 8, 4 + ((c - 4) >> 1), pick(c < 16, 6 + ((c - 8) >> 2), 8 + ((c - 16) >> 3)))); }

// the four context-coded bin costs a level's binarisation can touch: coeff_abs_level_greater1_flag = 0 / 1 in its current
// context and coeff_abs_level_greater2_flag = 0 / 1; looked up when the contexts change, not per candidate level
struct FlagBits { int32_t g1zero, g1one, g2zero, g2one; };

// the context-coded part of a level >= base (greater1 = 1, then greater2 = 1 while those flags are still coded) ...
__device__ __forceinline__ int32_t flagsOfCoded(const LevelState &s, const FlagBits &f)
{
    const int32_t both = f.g1one + pick(s.nG2 < 1, f.g2one, 0);
    return pick(s.nG1 < 8, both, 0);
}
// ... and of a level below it: 1 is greater1 = 0; 2 is greater1 = 1, greater2 = 0
__device__ __forceinline__ int32_t flagsOfSmall(int level, const FlagBits &f)
{
    const int32_t two = f.g1one + f.g2zero;
    return pick(level == 1, f.g1zero, 0) + pick(level == 2, two, 0);
}

// Rdoq.cpp:611-668 getLevelRateCost (without the lambda), branch free.  The escape loop `while (symbol >= (1 << length))
// symbol -= 1 << length++` ends with length = floor(log2(symbol + (1 << rice))).
__device__ __forceinline__ int32_t levelBits(int level, const LevelState &s, const FlagBits &f)
{
    const int base = baseLevel(s), symbol = level - base, rest = symbol - (3 << s.rice);
    const int length = 31 - __clz(max(rest, 0) + (1 << s.rice));
    const int shortBins = (symbol >> s.rice) + 1 + s.rice, longBins = 3 + length + 1 - s.rice + length;
    const int32_t coded = (pick(rest < 0, shortBins, longBins) << 15) + flagsOfCoded(s, f);
    return 32768 + pick(symbol >= 0, coded, flagsOfSmall(level, f));
}

// Rdoq.cpp:819-885 getLevelRate, branch free.  `for (top = 2; rest >= top; top <<= 1) egs += 2` gives 1 + 2 floor(log2(rest)).
__device__ __forceinline__ int levelRate(int level, const LevelState &s, const FlagBits &f)
{
    const int base = baseLevel(s), symbol = level - base;
    const int maxVlc = (0x4e2e1a0e07ull >> (8 * s.rice)) & 0xff;           // 7, 14, 26, 46, 78
    const int prefixMax = 8 - s.rice;                                      // 8, 7, 6, 5, 4
    const int rest = symbol - maxVlc;
    const int escapeBits = (1 + 2 * (31 - __clz(max(rest, 1)))) << 15;
    const int escape = pick(rest > 0, escapeBits, 0);
    const int capped = pick(rest > 0, maxVlc + 1, symbol);
    const int coded = escape + ((min(capped >> (s.rice + 1), prefixMax) + s.rice) << 15) + flagsOfCoded(s, f);
    return pick(symbol >= 0, coded, flagsOfSmall(level, f));
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

// ---------------------------------------------------------------------------------------------------------------------
// Transform coefficients after a QP-32 quantiser are sparse (3-5 % of the levels and ~15 % of the 4x4 groups of a 32x32 block
// are non-zero), and a group whose rounded levels are all zero does no more than add one flag cost: it keeps no level, leaves
// the level state alone and is never a candidate for the last position.  So:
//   * lane = transform block, 64 blocks per wavefront; a lane walks ITS block's groups in reverse scan order, hopping over the
//     all-zero ones (a handful of instructions each) and doing the full per-coefficient work (walkGroup) only on the others.
//     Nothing is speculated: the neighbours' flags and the carry are known when a group is reached.
//   * the search for the last significant position (Rdoq.cpp:342-399) is streamed: its running cost differs from the block's
//     final cost by a sum of per-coefficient deltas, so the best candidate relative to that final cost can be tracked group by
//     group, committed once the group's keep-or-zero decision is made, and compared with "code nothing" at the very end.
//   * sign-data hiding is applied to each group as it is finished, as if it were not the group holding the last significant
//     coefficient; that one group (known only at the end) is redone.
//   * the per-group record arrays live in LDS, [coefficient][lane]; the scan that finds the non-zero groups (and zero-fills the
//     output) is cooperative: 64 lanes read one 32x32 block's 64 groups (or four 16x16, ...) per step, coalesced.
//   * a wavefront runs as long as its densest block (32x32 at QP 32: 14 groups to walk against a mean of 10; 16x16: 5.6 against
//     3.2), so the large sizes are walked densest-first: scan kernel -> counting sort by groups to walk -> walk kernel, through a
//     caller-provided workspace; 8x8 and 4x4 blocks keep the scan inside the walk kernel and job order.
// ---------------------------------------------------------------------------------------------------------------------
// what sign-data hiding needs of the current group, [coefficient in scan order][lane]
struct WalkRecords
{
    int16_t kept[16][64];       // magnitudes, by RASTER position x | y << 2 inside the group (the order they leave in)
    int32_t costUp[16][64];     // level != 0: factor * -deltaU + rateIncUp;  level == 0: factor * -|deltaU| + (1 << 15) + rateIncUp + sigRateDelta
    int32_t costDown[16][64];   // level != 0: factor * deltaU + rateIncDown - (level == 1 ? (1 << 15) + sigRateDelta : 0)
};

struct WalkShared
{
    WalkRecords rec;
    int32_t bits[128];
    int16_t coef[16][64];                         // the current group's coefficients, raster order within the group
    // (4x4 blocks keep, per position, significance context << 25 | flag bits of the zero levels above it IN rec.costDown: a position's record is read by loop B
    // before that iteration writes the position's costDown, position 0's only when loop B never visits it, and sign-data hiding reads costDown of kept levels only)
    uint8_t rasterOf[3][64];                      // scan index -> raster group position, per scan type
    uint8_t scanOf[3][64];                        // ... and back
    uint32_t clsScan[3][4];                       // per scan type and neighbour case: sigPattern() by SCAN position, as two 16-bit masks (value 1 | value 2 << 16)
};

// what is per transform block rather than per lane: SLOTS blocks per wavefront (64 when a lane is a block).  NLEN = 2 * log2 size: the prefix lengths a coordinate of the
// block can have.  LDS_STATES false: no copy of the context states -- the lanes read their CTU's snapshot where it lies (round 6: the copy was 8 of a wavefront's 30 KB of
// LDS, which is what kept the walk kernels at one wavefront per SIMD; blocks in job order share their CTU's 128 bytes, one or two cache lines per access).
template <int SLOTS, int NLEN, bool LDS_STATES>
struct BlockTables
{
    static constexpr int slots = SLOTS, nlen = NLEN;
    static constexpr bool ldsStates = LDS_STATES;
    uint8_t states[LDS_STATES ? HAVOC_RDOQ_CTX_BYTES : 1][SLOTS];  // [context][block]
    int32_t lastBits[2][NLEN][SLOTS];
};

struct WalkResult
{
    int64_t cost, sigCost, dist0;      // the group's RD cost; the cost of its coded_sub_block_flag; the energy of its coefficients
    int64_t q;                         // change of the last-position search's running cost across the group, if it stays coded
    int64_t localBest;                 // best candidate of the group relative to the running cost at its start
    int localPos, localOr, groupOr;    // its scan position; OR of the levels from it to the end of the group; OR of all levels
    bool localStop;                    // a level > 1 ends the search (Rdoq.cpp:385-389)
    int coded, carry;
};

// Rdoq.cpp:517-603 for LOG2 > 2: the position-dependent part of the significance context (0, 1 or 2) for every position of
// a 4x4 group, two bits per raster position x | y << 2, given which of the right / below groups are coded
__host__ __device__ constexpr uint32_t sigPattern(int neighbours)
{
    uint32_t v = 0;
    for (int yp = 0; yp < 4; ++yp)
        for (int xp = 0; xp < 4; ++xp)
        {
            const int inc = neighbours == 0 ? (xp + yp == 0 ? 2 : (xp + yp < 3 ? 1 : 0))
                          : neighbours == 1 ? (yp == 0 ? 2 : (yp == 1 ? 1 : 0))
                          : neighbours == 2 ? (xp == 0 ? 2 : (xp == 1 ? 1 : 0)) : 2;
            v |= (uint32_t)inc << (2 * (xp | yp << 2));
        }
    return v;
}

// what sign-data hiding needs beside the records: which positions were inside the coded range, the greater1 context each
// zero level would have been coded in (c1, two bits per position) and the cost of a greater1 flag = 0 in each of the four
// ... and where the kept levels are (scan-order bit masks: non-zero, odd) and which coefficients are negative (by scan position and
// by raster position): the records hold magnitudes, the signs go on when the group is written out
struct SdhAux { uint32_t active, c1At, keptMask, oddMask, negScan, negRaster; int32_t g1zero[4]; };
__device__ __forceinline__ int32_t pick4(const int32_t (&v)[4], int k) { return pick(k == 0, v[0], pick(k == 1, v[1], pick(k == 2, v[2], v[3]))); }      // registers, not scratch

// Steps 1 and 2 of runQuantisation for one coefficient group (Rdoq.cpp:83-298), with the group's share of step 3
// (Rdoq.cpp:356-399) and the cost terms of sign-data hiding (Rdoq.cpp:950-957, :980) folded in.  Two loops instead of the
// reference's one: a level that rounds to zero never changes the entropy coder's level state, so
//   loop Z  walks the 16 positions once doing only what a zero level needs (energy, significance context, its flag cost as a
//           running prefix), and notes which positions hold a non-zero rounded level;
//   loop B  visits only those, in reverse scan order, with the state machine, the two-candidate level decision, the three
//           level rates of sign-data hiding and the last-position candidate.
// A wavefront's trip count of loop B is the largest number of non-zero levels any of its 64 blocks has in the group at hand.
template <int LOG2>
__device__ __forceinline__ WalkResult walkGroup(const Block &b, WalkShared &sh, int lane, int g, int gx, int gy, int firstPos, int caseBits, int factor,
                                                SdhAux &aux RT_PARAM)
{
    RT_MARK(3);
    WalkResult r;
    r.cost = r.sigCost = r.dist0 = r.q = 0;
    r.localBest = INT64_MAX;
    r.localPos = -1;
    r.localOr = r.groupOr = 0;
    r.localStop = false;
    const int firstGroup = firstPos >> 4, neighbours = caseBits & 3;
    LevelState st;
    st.c1 = 1;
    st.nG1 = st.nG2 = st.rice = 0;
    st.ctxSet = g == firstGroup ? ((firstPos < 16 || b.cIdx) ? 0 : 2) : ((g == 0 || b.cIdx) ? 0 : 2) + (caseBits >> 2);
    const int g1base = HAVOC_RDOQ_CTX_GREATER1 + 4 * st.ctxSet + (b.cIdx ? 16 : 0), g2 = HAVOC_RDOQ_CTX_GREATER2 + st.ctxSet + (b.cIdx ? 4 : 0);
    int32_t g1one[4];      // the context set is fixed for the group; the greater1 context moves with c1
#pragma unroll
    for (int c = 0; c < 4; ++c)
    {
        aux.g1zero[c] = bitsOf(b, g1base + c, 0);
        g1one[c] = bitsOf(b, g1base + c, 1);
    }
    FlagBits fb;
    fb.g2zero = bitsOf(b, g2, 0);
    fb.g2one = bitsOf(b, g2, 1);
    const uint32_t active = g < firstGroup ? 0xffffu : (2u << (firstPos & 15)) - 1;      // positions at or below the first non-zero level
    aux.active = active;

    // significance contexts of the group (Rdoq.cpp:517-603)
    const int sigChroma = b.cIdx ? 27 : 0;
    const int sigGroup = sigChroma + (b.cIdx == 0 ? ((gx + gy > 0 ? 3 : 0) + (LOG2 == 3 ? (b.scanIdx == 0 ? 9 : 15) : 21)) : (LOG2 == 3 ? 9 : 12));
    // Blocks above 4x4 (round 6): a position's context is one of FOUR -- base + 0 / 1 / 2 by where it sits in the group (sigPattern) or, for the block's DC
    // position, a context of its own -- so the eight bit costs are looked up once per group and every position takes its pair with selects; which scan
    // positions have which is a table lookup (clsTab: made once per wavefront).  The flag bits of the zero levels above a position -- a running sum the
    // first form kept per position in LDS -- are then popcounts of (zero positions above) & (positions of a class) times the class's cost, made only for the
    // positions loop B visits.  Per position loop Z is left with: magnitude, sign, energy, "rounds to non-zero", and the up-cost record of sign-data hiding.
    const bool dcGroup = gx + gy == 0;
    uint32_t cls1 = 0, cls2 = 0, clsD = 0, cls0 = 0;
    int32_t zc0 = 0, zc1 = 0, zc2 = 0, zcD = 0, oc0 = 0, oc1 = 0, oc2 = 0, ocD = 0;
    if (LOG2 > 2)
    {
        const uint32_t w = b.clsTab[neighbours];
        clsD = dcGroup ? 1u : 0u;
        cls1 = (w & 0xffff) & ~clsD;
        cls2 = (w >> 16) & ~clsD;
        cls0 = 0xffffu & ~(cls1 | cls2 | clsD);
        const int base = HAVOC_RDOQ_CTX_SIG + sigGroup, dcCtx = HAVOC_RDOQ_CTX_SIG + sigChroma;
        zc0 = bitsOf(b, base, 0); oc0 = bitsOf(b, base, 1);
        zc1 = bitsOf(b, base + 1, 0); oc1 = bitsOf(b, base + 1, 1);
        zc2 = bitsOf(b, base + 2, 0); oc2 = bitsOf(b, base + 2, 1);
        zcD = bitsOf(b, dcCtx, 0); ocD = bitsOf(b, dcCtx, 1);
    }

    // ---- loop Z ----
    uint32_t nzMask = 0, sumAll = 0, sumAllHi = 0, negScan = 0, negRaster = 0;
    int32_t zeroBits = 0;      // bits of the significance flags (= 0) of the zero levels met so far
    const int rnd = 1 << (b.quantShift - 1);
#pragma unroll
    for (int i = 15; i >= 0; --i)
    {
        const int nib = (int)(b.scan4 >> (4 * i)) & 15;
        const int c = sh.coef[nib][lane];
        const uint32_t a = (uint32_t)abs(c), sq = a * a;
        negScan |= (uint32_t)(c < 0) << i;
        negRaster |= (uint32_t)(c < 0) << nib;
        sumAll += sq & 0xffff;
        sumAllHi += sq >> 16;
        // straight-line: every position writes its records (loop B overwrites those of the non-zero levels), what differs is selected
        const bool inside = (active >> i) & 1;
        const int scaled = (int)a * b.quantScale;
        const bool nonZero = inside & (((scaled + rnd) >> b.quantShift) > 0);
        nzMask |= (uint32_t)nonZero << i;
        int32_t z, one;
        if (LOG2 == 2)
        {
            const int ctx = HAVOC_RDOQ_CTX_SIG + sigChroma + (int)((0x8877886654325410ull >> (4 * nib)) & 15);
            reinterpret_cast<uint32_t (*)[64]>(sh.rec.costDown)[i][lane] = (uint32_t)ctx << 25 | (uint32_t)zeroBits;
            z = bitsOf(b, ctx, 0);
            one = bitsOf(b, ctx, 1);
            zeroBits += pick(inside & !nonZero, z, 0);
        }
        else
        {
            const bool is1 = (cls1 >> i) & 1, is2 = (cls2 >> i) & 1, isD = i == 0 && dcGroup;
            z = pick(isD, zcD, pick(is1, zc1, pick(is2, zc2, zc0)));
            one = pick(isD, ocD, pick(is1, oc1, pick(is2, oc2, oc0)));
        }
        sh.rec.kept[nib][lane] = 0;
        const int32_t upZero = (int32_t)((uint32_t)factor * (uint32_t)-(scaled >> (b.quantShift - 8))) + (1 << 15) + one - z;      // + g1zero[c1] when used
        sh.rec.costUp[i][lane] = pick(inside, upZero, 1 << 15);
    }
    const uint32_t zeroMask = active & ~nzMask;      // positions inside the coded range whose level rounds to zero
    if (LOG2 > 2)
        zeroBits = zc0 * __popc(zeroMask & cls0) + zc1 * __popc(zeroMask & cls1) + zc2 * __popc(zeroMask & cls2) + zcD * __popc(zeroMask & clsD);
    const int64_t sumSq = ((int64_t)sumAllHi << 16) + sumAll;
    r.dist0 = sumSq << b.distShift;

    // ---- loop B ----
    RT_MARK(4);
    int nonZeroAbovePos0 = 0;
    int64_t gSig = 0, gSigPos0 = 0, gCoded = 0, gDist0 = 0, costB = 0, distB = 0, qB = 0;
    uint32_t c1At = 0x55555555u;      // c1 = 1 everywhere
    uint32_t keptMask = 0, oddMask = 0;
    for (uint32_t m = nzMask; m;)      // written with selects, not branches: 64 blocks run this body together
    {
        const int i = 31 - __clz((int)m);
        m ^= 1u << i;
        const int nib = (int)(b.scan4 >> (4 * i)) & 15, x = (gx << 2) + (nib & 3), y = (gy << 2) + (nib >> 2);
        const int a = abs((int)sh.coef[nib][lane]);
        const int64_t dist0 = (int64_t)(a * a) << b.distShift;
        const int sp = g * 16 + i;
        const int scaled = a * b.quantScale;
        const int level = (scaled + rnd) >> b.quantShift;
        const bool first = sp == firstPos;
        int32_t z0, z1, aboveBits;
        if (LOG2 == 2)
        {
            const uint32_t pk = reinterpret_cast<const uint32_t (*)[64]>(sh.rec.costDown)[i][lane];
            const int sc = (int)(pk >> 25);
            aboveBits = (int32_t)(pk & 0x1ffffff);
            z0 = bitsOf(b, sc, 0);
            z1 = bitsOf(b, sc, 1);
        }
        else
        {
            const bool is1 = (cls1 >> i) & 1, is2 = (cls2 >> i) & 1, isD = (clsD >> i) & 1;
            z0 = pick(isD, zcD, pick(is1, zc1, pick(is2, zc2, zc0)));
            z1 = pick(isD, ocD, pick(is1, oc1, pick(is2, oc2, oc0)));
            const uint32_t above = zeroMask & ~((2u << i) - 1);      // the zero levels at higher scan positions (the block's DC position is never one of them)
            aboveBits = zc0 * __popc(above & cls0) + zc1 * __popc(above & cls1) + zc2 * __popc(above & cls2);
        }
        const int64_t zerosAbove = b.lambda * aboveBits;
        const int32_t sigZero = pick(first, 0, z0), sigOneBits = pick(first, 0, z1);
        fb.g1zero = pick4(aux.g1zero, st.c1);
        fb.g1one = pick4(g1one, st.c1);

        // Rdoq.cpp:456-515: drop (small levels only), keep, or lower by one
        const bool droppable = !first && level < 3;
        const int64_t sigOne = b.lambda * sigOneBits, dropSig = b.lambda * sigZero;
        const int lower = level - 1;
        const int32_t err1 = a - clip16((clip16(level) * b.invScale + b.invOffset) >> b.invShift);
        const int32_t err2 = a - clip16((clip16(lower) * b.invScale + b.invOffset) >> b.invShift);
        const int64_t cost1 = ((int64_t)(int32_t)((uint32_t)err1 * (uint32_t)err1) << b.distShift) + b.lambda * levelBits(level, st, fb) + sigOne;
        const int64_t cost2If = ((int64_t)(int32_t)((uint32_t)err2 * (uint32_t)err2) << b.distShift) + b.lambda * levelBits(lower, st, fb) + sigOne;
        const int64_t cost2 = pick(lower >= 1, cost2If, (int64_t)INT64_MAX);
        const int64_t dropCost = dist0 + dropSig;
        int64_t costCoded = pick(droppable, dropCost, (int64_t)INT64_MAX), costSig = pick(droppable, dropSig, (int64_t)0);
        int kept = 0;
        const bool take1 = cost1 < costCoded;
        kept = pick(take1, level, kept);
        costSig = pick(take1, sigOne, costSig);
        costCoded = pick(take1, cost1, costCoded);
        const bool take2 = cost2 < costCoded;
        kept = pick(take2, lower, kept);
        costSig = pick(take2, sigOne, costSig);
        costCoded = pick(take2, cost2, costCoded);

        costB += costCoded;
        distB += dist0;
        const int du = (scaled - (kept << b.quantShift)) >> (b.quantShift - 8), sigDelta = sigOneBits - sigZero;
        const int stored = (int16_t)kept;
        const int now = levelRate(kept, st, fb);
        const int upKept = factor * -du + levelRate(kept + 1, st, fb) - now;
        const int upZero = factor * -abs(du) + (1 << 15) + sigDelta;      // + g1zero[c1] when used, like the other zero levels
        sh.rec.kept[nib][lane] = (int16_t)kept;
        keptMask |= (uint32_t)(stored != 0) << i;
        oddMask |= (uint32_t)(stored & 1) << i;
        sh.rec.costUp[i][lane] = pick(kept > 0, upKept, upZero);
        const int lastOne = (1 << 15) + sigDelta;
        sh.rec.costDown[i][lane] = factor * du + levelRate(kept - 1, st, fb) - now - pick(kept == 1, lastOne, 0);

        // Rdoq.cpp:773-800
        const bool grow = (kept >= baseLevel(st)) & (kept > 3 * (1 << st.rice));
        st.rice = pick(grow, min(st.rice + 1, 4), st.rice);
        st.nG1 += kept >= 1;
        st.nG2 += kept > 1;
        const bool step = (st.c1 < 3) & (st.c1 > 0) & (kept != 0);
        st.c1 = pick(kept > 1, 0, pick(step, st.c1 + 1, st.c1));
        {
            const uint32_t below = (1u << (2 * i)) - 1;      // every position still to come sees the new c1
            c1At = (c1At & ~below) | ((0x55555555u * (uint32_t)st.c1) & below);
        }
        gSig += costSig;
        gSigPos0 = pick(i == 0, costSig, gSigPos0);
        const int64_t codedPart = costCoded - costSig;
        gCoded += pick(stored != 0, codedPart, (int64_t)0);
        gDist0 += pick(stored != 0, dist0, (int64_t)0);
        nonZeroAbovePos0 += (stored != 0) & (i != 0);
        // candidate for the last significant position (Rdoq.cpp:356-399)
        const int lx = lastPrefixLength(x), ly = lastPrefixLength(y);
        const int32_t rate = b.lastBits[(b.scanIdx == 2 ? ly : lx) * b.lastStride] + b.lastBits[(b.lastLen + (b.scanIdx == 2 ? lx : ly)) * b.lastStride];
        const int64_t total = qB - zerosAbove + b.lambda * rate - costSig;
        const bool better = (stored != 0) & !r.localStop & (total < r.localBest);
        r.localBest = pick(better, total, r.localBest);
        r.localPos = pick(better, sp, r.localPos);
        r.localOr = pick(better, 0, r.localOr) | stored;
        r.groupOr |= stored;
        r.localStop |= stored > 1;
        const int64_t keptGain = dist0 - costCoded, zeroGain = -costSig;
        qB += pick(stored != 0, keptGain, zeroGain);
    }
    RT_MARK(5);
    aux.c1At = c1At;
    aux.keptMask = keptMask;
    aux.oddMask = oddMask;
    aux.negScan = negScan;
    aux.negRaster = negRaster;
    const int64_t zeroCost = b.lambda * zeroBits;
    r.cost = (r.dist0 - distB) + zeroCost + costB;      // inactive and zero levels: their energy (+ the zero levels' flags); the others: their RD cost
    r.q = qB - zeroCost;
    gSig += zeroCost;
    // position 0 is always inside the coded range: when its level rounds to zero, its flag is what the running sum gained last
    if (!(nzMask & 1))
        gSigPos0 = LOG2 == 2 ? zeroCost - b.lambda * (int32_t)(reinterpret_cast<const uint32_t (*)[64]>(sh.rec.costDown)[0][lane] & 0x1ffffff) : b.lambda * pick(dcGroup, zcD, pick((bool)(cls1 & 1), zc1, pick((bool)(cls2 & 1), zc2, zc0)));
    r.carry = st.c1 == 0;
    // step 2 (Rdoq.cpp:196-297), as selects.  The DC group is coded whatever it holds; a group without a kept level pays the flag = 0 and
    // gets the significance flags of its zero levels back; a group below the first weighs zeroing all its levels against flag = 1 (a
    // lone level at position 0 implies its significance flag); the first group is coded and pays no flag.
    const bool dc = g == 0, any = keptMask != 0, inner = any & !dc & (g < firstGroup);
    const int64_t zero = pick(neighbours != 0, b.csbfZero[1], b.csbfZero[0]), one = pick(neighbours != 0, b.csbfOne[1], b.csbfOne[0]);
    const int64_t implied = pick(inner & (nonZeroAbovePos0 == 0), gSigPos0, (int64_t)0);
    const int64_t sigLeft = gSig - implied, allZero = zero + gDist0 - gCoded - sigLeft;
    const bool dropped = inner & (allZero < one), empty = !any & !dc;
    const int64_t innerCost = pick(dropped, allZero, one) - implied, emptyCost = zero - gSig;
    r.cost += pick(empty, emptyCost, pick(inner, innerCost, (int64_t)0));
    r.sigCost = pick(empty | dropped, zero, pick(inner, one, (int64_t)0));
    r.coded = dc | (any & !dropped);
    RT_MARK(6);
    return r;
}

// Rdoq.cpp:887-1023 for one group.  Works on the magnitudes in rec.kept: a level's sign is its coefficient's, so "dst += change
// for a non-negative coefficient, dst -= change otherwise" is magnitude += change; sum & 1 is the parity of the magnitudes.
__device__ __forceinline__ void hideSignsWalk(WalkShared &sh, int lane, const Block &b, bool lastGroup, const SdhAux &aux, uint32_t keptMask)
{
    // the group needs a change when it holds kept levels at least 4 scan positions apart and the parity of their sum is not the sign
    // of the first of them (Rdoq.cpp:905-930); the search below is straight-line so that the lanes which need it share one pass
    const int first = max(__ffs((int)keptMask) - 1, 0), last = keptMask ? 31 - __clz((int)keptMask) : 0;
    const int signbit = (int)(aux.negScan >> first) & 1;
    const bool needed = (keptMask != 0) & (last - first >= 4) & (signbit != (__popc(aux.oddMask & keptMask) & 1));
    if (__ballot(needed) == 0) return;
    const int top = lastGroup ? last : 15;      // positions above the last significant coefficient are not candidates
    int minCost = INT32_MAX, minIdx = 0, finalChange = 0;
#pragma unroll
    for (int i = 15; i >= 0; --i)
    {
        const bool isKept = (keptMask >> i) & 1;
        const int mag = (uint16_t)sh.rec.kept[(int)(b.scan4 >> (4 * i)) & 15][lane];
        const int up = sh.rec.costUp[i][lane];
        // a kept level: one up or one down, whichever is cheaper (down is no option for a first level of 1: it would vanish)
        const int downRaw = pick(isKept, sh.rec.costDown[i][lane], 0);      // only the kept positions have this record
        const int down = downRaw - pick(lastGroup & (last == i) & (mag == 1), 4 << 15, 0);
        const bool upBetter = up < down;
        const int costKept = pick(upBetter, up, pick((i == first) & (mag == 1), INT32_MAX, down));
        // a zero level: up to one, unless it comes before the first kept level with the other sign
        const int g1 = pick((aux.active >> i) & 1, pick4(aux.g1zero, (int)(aux.c1At >> (2 * i)) & 3), 0);
        const int costZero = pick((i < first) & (((int)(aux.negScan >> i) & 1) != signbit), INT32_MAX, up + g1);
        const int cost = pick(isKept, costKept, costZero), change = pick(isKept & !upBetter, -1, 1);
        const bool better = (i <= top) & (cost < minCost);
        minCost = pick(better, cost, minCost);
        finalChange = pick(better, change, finalChange);
        minIdx = pick(better, i, minIdx);
    }
    if (!needed) return;
    const int nib = (int)(b.scan4 >> (4 * minIdx)) & 15;
    const int mag = (uint16_t)sh.rec.kept[nib][lane], negative = (int)(aux.negScan >> minIdx) & 1;
    if ((!negative && mag == 32767) || (negative && mag == 32768)) finalChange = -1;
    sh.rec.kept[nib][lane] = (int16_t)(mag + finalChange);
}

// Workspace of one launch (caller-provided, havoc_mi355x_rdoq_workspace bytes): what the scan pass found per block, the
// histogram of blocks by their number of groups to walk, and the order the walk takes the blocks in.
// LDS of the cooperative scan of a workgroup's 64 blocks
struct ScanShared
{
    int32_t srcOff[64], dstOff[64], nzThreshold[64], gt1Threshold[64], gt2Threshold[64];
    uint64_t mask[64], mask2[64], mask3[64];      // groups of each block with a rounded level > 0 / > 1 / > 2 (bit = raster group position)
    int64_t sumSq[64];      // sum of squared coefficients of each block
};

// Cooperative scan: 64 lanes look at one 32x32 block's 64 groups (or four 16x16 blocks, ...) per step, coalesced: which groups hold
// a non-zero rounded level, the block's energy; zeros into the output.  Results for block k of the workgroup in sc.mask[k], sc.sumSq[k].
template <int LOG2>
__device__ __forceinline__ void scanBlocks(ScanShared &sc, int16_t *__restrict__ dstAll, const int16_t *__restrict__ srcAll, const RdoqJob *__restrict__ jobs, int njobs,
                                           int firstBlock, int steps)
{
    constexpr int size = 1 << LOG2, G = (size * size) >> 4, log2G = 2 * LOG2 - 4, gw = size >> 2, perStep = 64 / G;
    const int lane = threadIdx.x, blk = firstBlock + lane;
    {
        const RdoqJob &job = jobs[blk < njobs ? blk : njobs - 1];
        sc.srcOff[lane] = job.src_off;
        sc.dstOff[lane] = job.dst_off;
        // smallest |coefficient| whose rounded level is non-zero (Rdoq.cpp:108): |c| * scale + half >= 2 * half  <=>  |c| >= ceil(half / scale)
        const uint32_t half = 1u << (job.quant_shift - 1), scale = (uint32_t)max(job.quant_scale, 1);
        sc.nzThreshold[lane] = (int32_t)((half + scale - 1) / scale);
        sc.gt1Threshold[lane] = (int32_t)((3 * half + scale - 1) / scale);      // rounded level >= 2  <=>  |c| * scale >= 3 * half
        sc.gt2Threshold[lane] = (int32_t)((5 * half + scale - 1) / scale);      // rounded level >= 3  <=>  |c| * scale >= 5 * half
        sc.mask[lane] = sc.mask2[lane] = sc.mask3[lane] = 0;
        sc.sumSq[lane] = 0;
    }
    __syncthreads();
    const int sub = lane >> log2G, pos = lane & (G - 1), px = pos & (gw - 1), py = pos / gw;
    for (int step = 0; step < steps; ++step)    // 64 / G blocks per step
    {
        const int bl = step * perStep + sub;
        const bool have = firstBlock + bl < njobs;
        const int16_t *p = srcAll + (long)sc.srcOff[bl] + (py * 4) * size + px * 4;
        int16_t *q = dstAll + (long)sc.dstOff[bl] + (py * 4) * size + px * 4;
        const uint32_t thr = (uint32_t)sc.nzThreshold[bl], thr2 = (uint32_t)sc.gt1Threshold[bl], thr3 = (uint32_t)sc.gt2Threshold[bl];
        bool nz = false, nz2 = false, nz3 = false;
        uint32_t lo = 0, hi = 0;
        if (have)
        {
            u16x2v big = {0, 0};      // |c| and c^2 two at a time: packed max against the negation, v_dot2 of a pair with itself
            for (int r = 0; r < 4; ++r)
            {
                const u32x2 v = ld8(p + r * size);
                st8(q + r * size, u32x2{0, 0});
                const uint32_t w[2] = {v.x, v.y};
                for (int k = 0; k < 2; ++k)
                {
                    const s16x2v c = __builtin_bit_cast(s16x2v, w[k]);
                    const s16x2v a = __builtin_elementwise_max(c, (s16x2v){0, 0} - c);      // -32768 stays 0x8000 = 32768 unsigned
                    big = __builtin_elementwise_max(big, __builtin_bit_cast(u16x2v, a));
                    const uint32_t sq = (uint32_t)__builtin_amdgcn_sdot2(c, c, 0, false);   // c0^2 + c1^2 (mod 2^32: at most 2^31)
                    lo += sq & 0xffff;
                    hi += sq >> 16;
                }
            }
            nz = max((uint32_t)big.x, (uint32_t)big.y) >= thr;
            nz2 = max((uint32_t)big.x, (uint32_t)big.y) >= thr2;
            nz3 = max((uint32_t)big.x, (uint32_t)big.y) >= thr3;
        }
        const uint64_t m = __ballot(nz), m2 = __ballot(nz2), m3 = __ballot(nz3);
        const int slo = group_sum<G>((int)lo), shi = group_sum<G>((int)hi);
        if (pos == 0 && have)
        {
            sc.mask[bl] = G == 64 ? m : (m >> (lane & ~(G - 1))) & ((1ull << (G & 63)) - 1);
            sc.mask2[bl] = G == 64 ? m2 : (m2 >> (lane & ~(G - 1))) & ((1ull << (G & 63)) - 1);
            sc.mask3[bl] = G == 64 ? m3 : (m3 >> (lane & ~(G - 1))) & ((1ull << (G & 63)) - 1);
            sc.sumSq[bl] = ((int64_t)shi << 16) + slo;
        }
    }
    __syncthreads();
}

// Pass 1 of the sorted form (large blocks): the scan, kScanSteps steps per workgroup (eight 32x32 blocks, or thirty-two 16x16 blocks,
// per wavefront: enough wavefronts to fill the machine, few enough histogram atomics), its results to the workspace, histogram of
// the blocks by groups to walk.
constexpr int kScanSteps = 8;
template <int LOG2>
__global__ __launch_bounds__(64) void k_rdoq_scan(int16_t *__restrict__ dstAll, const int16_t *__restrict__ srcAll, const RdoqJob *__restrict__ jobs, int njobs,
                                                  RdoqWork *__restrict__ work, int withHist)
{
    constexpr int perWg = kScanSteps * (64 / (((1 << LOG2) * (1 << LOG2)) >> 4));      // <= 64
    __shared__ ScanShared sc;
    __shared__ uint32_t hist[kBins];
    const int lane = threadIdx.x, first = blockIdx.x * perWg, blk = first + lane;
    RdoqInfo *info = reinterpret_cast<RdoqInfo *>(reinterpret_cast<char *>(work) + rdoqInfoOffset());
    for (int k = lane; k < kBins; k += 64) hist[k] = 0;
    scanBlocks<LOG2>(sc, dstAll, srcAll, jobs, njobs, first, kScanSteps);
    if (lane < perWg && blk < njobs)
    {
        RdoqInfo r;
        r.mask = sc.mask[lane];
        r.mask2 = sc.mask2[lane];
        r.mask3 = sc.mask3[lane];
        r.sumSq = sc.sumSq[lane];
        info[blk] = r;
        atomicAdd(&hist[groupsToWalk(r.mask)], 1u);
        if (jobs[blk].scan_idx != 0) atomicAdd(&hist[kBins - 1], 1u);      // the spare bin (never a number of groups: at most 64) counts them
    }
    __syncthreads();
    if (!withHist) return;      // a launch walked in job order: nothing reads the histogram
    for (int k = lane; k < kBins - 1; k += 64)
        if (hist[k]) atomicAdd(&work->hist[k], hist[k]);
    if (lane == 0 && hist[kBins - 1]) atomicAdd(&work->otherScans, hist[kBins - 1]);
}

// The histogram alone, for blocks whose scan was done elsewhere (k_tu_forward<..., SCAN> left RdoqInfo per block in the workspace and zeroed the
// level blocks): reads 32 bytes per block instead of the block.
__global__ __launch_bounds__(256) void k_rdoq_hist(const RdoqJob *__restrict__ jobs, int njobs, RdoqWork *__restrict__ work)
{
    __shared__ uint32_t hist[kBins];
    const RdoqInfo *info = reinterpret_cast<const RdoqInfo *>(reinterpret_cast<const char *>(work) + rdoqInfoOffset());
    const int t = threadIdx.x, blk = blockIdx.x * 256 + t;
    if (t < kBins) hist[t] = 0;
    __syncthreads();
    if (blk < njobs)
    {
        atomicAdd(&hist[groupsToWalk(info[blk].mask)], 1u);
        if (jobs[blk].scan_idx != 0) atomicAdd(&hist[kBins - 1], 1u);
    }
    __syncthreads();
    if (t < kBins - 1 && hist[t]) atomicAdd(&work->hist[t], hist[t]);
    if (t == kBins - 1 && hist[t]) atomicAdd(&work->otherScans, hist[t]);
}

// Pass 2: blocks ordered by decreasing number of groups to walk (counting sort; the order inside a bin is whatever the atomics
// give -- it decides only which lane walks which block).  The 64 blocks of a wavefront then finish together.
template <int LOG2>
__global__ __launch_bounds__(256) void k_rdoq_order(const RdoqJob *__restrict__ jobs, int njobs, RdoqWork *__restrict__ work)
{
    __shared__ uint32_t count[kBins], base[kBins];
    RdoqInfo *info = reinterpret_cast<RdoqInfo *>(reinterpret_cast<char *>(work) + rdoqInfoOffset());
    uint32_t *order = reinterpret_cast<uint32_t *>(info + njobs);
    const int t = threadIdx.x, blk = blockIdx.x * 256 + t;
    if (t < kBins) count[t] = 0;
    __syncthreads();
    int bin = 0;
    uint32_t rank = 0;
    if (blk < njobs)
    {
        RdoqInfo v = info[blk];
        bin = groupsToWalk(v.mask);
        rank = atomicAdd(&count[bin], 1u);
        // what the walks read: the masks by scan position (the DC group is number 0 either way: the bin stays what the histogram counted)
        toScanOrder<(1 << LOG2) / 4>(jobs[blk].scan_idx, v.mask, v.mask2, v.mask3);
        info[blk] = v;
    }
    __syncthreads();
    __shared__ uint32_t histAll[kBins];
    if (t < kBins) histAll[t] = work->hist[t];      // (one load per lane, then the sums in LDS: 65 dependent global loads per lane made this launch 20-50 us)
    __syncthreads();
    if (t < kBins && count[t])
    {
        uint32_t before = 0;
        for (int k = t + 1; k < kBins - 1; ++k) before += histAll[k];      // bins with more groups come first
        base[t] = before + atomicAdd(&work->cursor[t], count[t]);
    }
    __syncthreads();
    if (blk < njobs) order[base[bin] + rank] = (uint32_t)blk;
}

// ---- what the two walk kernels share: a lane's view of its transform block ----
template <int LOG2>
struct LaneBlock
{
    static constexpr int size = 1 << LOG2, G = (size * size) >> 4, gw = size >> 2;
    Block b;
    const uint8_t *rasterOf;      // LDS: scan index of a group -> its raster position
    const uint8_t *scanOf;        // LDS: ... and back
    const int16_t *src;
    int16_t *dst;
    int distShift;

    // tables and the block's context states into LDS, then the per-lane constants (Rdoq.h:163-187, Rdoq.cpp:706-771); all 64 lanes call it.
    // `slot` = the block's place in the wavefront, shared by `per` lanes of which this one is number `sub`.
    template <class BT>
    __device__ __forceinline__ void stageIn(WalkShared &sh, BT &bt, int lane, int slot, int sub, int per, const RdoqJob &job,
                                            const uint8_t *__restrict__ statesAll, int bitDepth, const int16_t *__restrict__ srcAll, int16_t *__restrict__ dstAll)
    {
        constexpr int SLOTS = BT::slots;
        static_assert(BT::nlen == 2 * LOG2, "prefix lengths of a coordinate");
        sh.bits[lane] = kEntropyBits[lane];
        sh.bits[64 + lane] = kEntropyBits[64 + lane];
        if (BT::ldsStates)
        {
            const uint32_t *st = reinterpret_cast<const uint32_t *>(statesAll + (long)job.ctx_index * HAVOC_RDOQ_CTX_BYTES);
            for (int k = sub; k < HAVOC_RDOQ_CTX_BYTES / 4; k += per)
            {
                const uint32_t v = st[k];
                bt.states[4 * k][slot] = (uint8_t)v;
                bt.states[4 * k + 1][slot] = (uint8_t)(v >> 8);
                bt.states[4 * k + 2][slot] = (uint8_t)(v >> 16);
                bt.states[4 * k + 3][slot] = (uint8_t)(v >> 24);
            }
        }
        if (lane < G)
            for (int t = 0; t < 3; ++t)
            {
                int x = 0, y = 0;
                if (G > 1) scanXy(gw, t, lane, x, y);
                sh.rasterOf[t][lane] = (uint8_t)(y * gw + x);
                sh.scanOf[t][y * gw + x] = (uint8_t)lane;
            }
        if (lane < 12)
        {
            const int t = lane >> 2, nb = lane & 3;
            const uint64_t sc = t == 0 ? scan4Nibbles(0) : (t == 1 ? scan4Nibbles(1) : scan4Nibbles(2));
            const uint32_t pat = nb == 0 ? sigPattern(0) : nb == 1 ? sigPattern(1) : nb == 2 ? sigPattern(2) : sigPattern(3);
            uint32_t one = 0, two = 0;
            for (int i = 0; i < 16; ++i)
            {
                const uint32_t inc = (pat >> (2 * ((int)(sc >> (4 * i)) & 15))) & 3;
                one |= (uint32_t)(inc == 1) << i;
                two |= (uint32_t)(inc == 2) << i;
            }
            sh.clsScan[t][nb] = one | two << 16;
        }
        __syncthreads();

        b.states = BT::ldsStates ? &bt.states[0][slot] : statesAll + (long)job.ctx_index * HAVOC_RDOQ_CTX_BYTES;
        b.stateStride = BT::ldsStates ? SLOTS : 1;
        b.lastBits = &bt.lastBits[0][0][slot];
        b.lastStride = SLOTS;
        b.lastLen = BT::nlen;
        b.bits = sh.bits;
        b.lambda = job.lambda_q16;
        const int transformShift = 15 - bitDepth - LOG2;
        distShift = 15 - 2 * transformShift - 2 * (bitDepth - 8) + 16;
        b.distShift = distShift;
        b.invShift = 6 - transformShift;
        b.invOffset = 1 << (b.invShift - 1);
        b.quantScale = job.quant_scale;
        b.quantShift = job.quant_shift;
        b.invScale = job.inv_scale;
        b.cIdx = job.c_idx;
        b.scanIdx = job.scan_idx;
        b.scan4 = job.scan_idx == 0 ? scan4Nibbles(0) : (job.scan_idx == 1 ? scan4Nibbles(1) : scan4Nibbles(2));
        rasterOf = sh.rasterOf[job.scan_idx < 3 ? job.scan_idx : 0];
        scanOf = sh.scanOf[job.scan_idx < 3 ? job.scan_idx : 0];
        b.clsTab = sh.clsScan[job.scan_idx < 3 ? job.scan_idx : 0];
        src = srcAll + job.src_off;
        dst = dstAll + job.dst_off;
        if (G > 1)      // (a 4x4 block has no group flag)
        {
            const int csbf = HAVOC_RDOQ_CTX_CSBF + (b.cIdx ? 2 : 0);
            b.csbfZero[0] = b.lambda * bitsOf(b, csbf, 0);
            b.csbfOne[0] = b.lambda * bitsOf(b, csbf, 1);
            b.csbfZero[1] = b.lambda * bitsOf(b, csbf + 1, 0);
            b.csbfOne[1] = b.lambda * bitsOf(b, csbf + 1, 1);
        }
        else
            b.csbfZero[0] = b.csbfZero[1] = b.csbfOne[0] = b.csbfOne[1] = 0;
        {
            const int cbfCtx = (!job.is_intra && b.cIdx == 0) ? HAVOC_RDOQ_CTX_ROOT_CBF : (b.cIdx == 0 ? HAVOC_RDOQ_CTX_CBF_LUMA + 1 : HAVOC_RDOQ_CTX_CBF_CHROMA);
            b.cbfZero = b.lambda * bitsOf(b, cbfCtx, 0);
            b.cbfOne = b.lambda * bitsOf(b, cbfCtx, 1);
        }
        if (sub == 0)
            for (int axis = 0; axis < 2; ++axis)      // Rdoq.cpp:706-771 per prefix length
            {
                const int base = axis ? HAVOC_RDOQ_CTX_LAST_Y : HAVOC_RDOQ_CTX_LAST_X;
                const int offset = b.cIdx ? 15 : 3 * (LOG2 - 2) + ((LOG2 - 1) >> 2), shift = b.cIdx ? LOG2 - 2 : (LOG2 + 1) >> 2;
                int32_t ones = 0;
                for (int len = 0; len < BT::nlen; ++len)      // (the longest prefix of a 32x32 block, 9 ones, has no terminating zero)
                {
                    const int ctx = base + min(max((len >> shift) + offset, 0), 17);
                    bt.lastBits[axis][len][slot] = ones + (len < 9 ? bitsOf(b, ctx, 0) : 0) + (len > 3 ? 32768 * ((len - 2) >> 1) : 0);
                    if (len < 9) ones += bitsOf(b, ctx, 1);
                }
            }
        __syncthreads();
    }
    __device__ __forceinline__ void loadGroup(WalkShared &sh, int lane, int gx, int gy) const
    {
        for (int r = 0; r < 4; ++r)
        {
            const u32x2 v = ld8(src + ((gy << 2) + r) * size + (gx << 2));
            sh.coef[4 * r][lane] = (int16_t)v.x;
            sh.coef[4 * r + 1][lane] = (int16_t)(v.x >> 16);
            sh.coef[4 * r + 2][lane] = (int16_t)v.y;
            sh.coef[4 * r + 3][lane] = (int16_t)(v.y >> 16);
        }
    }
    // rec.kept (magnitudes, raster order) -> truncation at lastIdx, sign-data hiding, signs (Rdoq.cpp:418-441), the output block
    __device__ __forceinline__ void finishGroup(WalkShared &sh, int lane, bool sdh, const SdhAux &aux, int g, int lastIdx, bool lastGroup, int gx, int gy) const
    {
        uint32_t keptMask = aux.keptMask;
        if (lastGroup)
        {
            const int n = lastIdx - g * 16;      // scan positions of this group that stay
            const uint32_t stay = n >= 16 ? 0xffffu : (1u << n) - 1;
            for (int i = 0; i < 16; ++i)
                if (!((stay >> i) & 1)) sh.rec.kept[(int)(b.scan4 >> (4 * i)) & 15][lane] = 0;
            keptMask &= stay;
        }
        if (sdh) hideSignsWalk(sh, lane, b, lastGroup, aux, keptMask);
        for (int r = 0; r < 4; ++r)
        {
            int v[4];
            for (int k = 0; k < 4; ++k)
            {
                const int mag = (uint16_t)sh.rec.kept[4 * r + k][lane];
                v[k] = ((aux.negRaster >> (4 * r + k)) & 1) ? -mag : mag;
            }
            u32x2 o;
            o.x = (uint32_t)(uint16_t)v[0] | (uint32_t)(uint16_t)v[1] << 16;
            o.y = (uint32_t)(uint16_t)v[2] | (uint32_t)(uint16_t)v[3] << 16;
            st8(dst + ((gy << 2) + r) * size + (gx << 2), o);
        }
    }
    __device__ __forceinline__ void clearGroup(int p) const
    {
        for (int r = 0; r < 4; ++r) st8(dst + (((p / gw) << 2) + r) * size + ((p & (gw - 1)) << 2), u32x2{0, 0});
    }
    // right coded | below coded << 1 | carry << 2
    static __device__ __forceinline__ int caseOf(uint64_t coded, int gx, int gy, int carry)
    {
        const int p = gy * gw + gx;
        const int right = gx < gw - 1 ? (int)(coded >> (p + 1)) & 1 : 0, below = gy < gw - 1 ? (int)(coded >> (p + gw)) & 1 : 0;
        return right | below << 1 | carry << 2;
    }
    // position of the first non-zero rounded level of the group now in sh.coef (scan order within the group), or -1
    __device__ __forceinline__ int firstInGroup(const WalkShared &sh, int lane) const
    {
        uint32_t nz = 0;      // straight-line: sixteen independent reads, one wait (a loop that returns at the first hit runs as long as the unluckiest lane, a round trip per position)
#pragma unroll
        for (int i = 0; i < 16; ++i)
        {
            const int a = abs((int)sh.coef[(int)(b.scan4 >> (4 * i)) & 15][lane]);
            nz |= (uint32_t)(((a * b.quantScale + (1 << (b.quantShift - 1))) >> b.quantShift) > 0) << i;
        }
        return nz ? 31 - __clz((int)nz) : -1;
    }
    // Rdoq.cpp:307-341: the index one past the last significant position (0: nothing is coded)
    __device__ __forceinline__ int lastIndex(bool isIntra, int64_t sumSq, int64_t walkedDist0, int64_t costTu, int64_t bestRel, int bestPos) const
    {
        const int64_t dist0Total = sumSq << distShift;
        const int64_t bestNone = dist0Total + b.cbfZero;
        const int64_t start = (dist0Total - walkedDist0) + costTu + b.cbfOne;
        return (bestPos >= 0 && start + bestRel < bestNone) ? bestPos + 1 : 0;
    }
};

// the running state of a block's walk (identical in every lane that follows the block)
struct WalkState
{
    int64_t costTu = 0, walkedDist0 = 0;    // costTu: everything but the energy of the coefficients outside the walked groups
    int64_t rel = 0;                        // running cost of the last-position search relative to its start
    int64_t bestRel = INT64_MAX;
    int bestPos = -1, orSince = 0;
    bool stopped = false;
    uint64_t coded = 0, carries = 0;        // by raster position; carry INTO each walked group, by scan index
    uint64_t codedScan = 0;                 // `coded` by scan index: what the verdict clears above the last significant group
    uint64_t nbrScan = 0;                   // by scan index: groups whose right or below neighbour is coded (what picks a coded_sub_block_flag's context)

    // the all-zero groups `range` (by scan index) between two walked groups: each pays its flag = 0, in the context its neighbours give it (Rdoq.cpp:200-210) --
    // a sum of per-group terms, so counted instead of visited
    __device__ __forceinline__ void zeroGroups(uint64_t range, const Block &b)
    {
        const int n = __popcll(range), n1 = __popcll(range & nbrScan);
        const int64_t zero = (int64_t)(n - n1) * b.csbfZero[0] + (int64_t)n1 * b.csbfZero[1];
        costTu += zero;
        rel -= zero;
    }
    __device__ __forceinline__ void walkedGroup(const WalkResult &r, int g, int p, int carryIn, const uint8_t *scanOf, int gw)
    {
        costTu += r.cost;
        walkedDist0 += r.dist0;
        coded |= (uint64_t)r.coded << p;
        codedScan |= (uint64_t)r.coded << g;
        if (gw > 1)
        {
            const int gx = p & (gw - 1), gy = p / gw;
            const bool left = r.coded && gx > 0, above = r.coded && gy > 0;
            nbrScan |= (uint64_t)left << scanOf[left ? p - 1 : 0];
            nbrScan |= (uint64_t)above << scanOf[above ? p - gw : 0];
        }
        carries |= (uint64_t)carryIn << g;
        rel -= r.sigCost;
        if (r.coded)
        {
            if (!stopped && r.localPos >= 0 && rel + r.localBest < bestRel)
            {
                bestRel = rel + r.localBest;
                bestPos = r.localPos;
                orSince = r.localOr;
            }
            else
                orSince |= r.groupOr;
            stopped |= r.localStop;
            rel += r.q;
        }
    }
};

// The sequential walk, a lane per block.  Where the blocks and what the scan found of them come from:
//   kInKernel  8x8, 4x4: many short blocks -- the scan runs here, blocks in job order;
//   kSorted    32x32 / 16x16 of a launch with more wavefronts than the machine holds at once: blocks in the order of pass 2 (densest first, so that a wavefront's 64
//              blocks finish together), scan results from the workspace, masks already by scan position (k_rdoq_order);
//   kJobOrder  32x32 / 16x16 of a smaller launch (round 6): blocks in job order, scan results from the workspace as the scan left them.  Every wavefront of such a
//              launch is resident from the start and the launch lasts as long as its longest wavefront whichever blocks share it, so the histogram and the sort --
//              three dependent launches, 25-60 us of the picture's critical path at 1080p (profiles/r06/rdoq_side_by_side_timeline.txt) -- buy nothing.
// onlyOther: walk only the blocks that do NOT use the diagonal scan (k_rdoq_diag takes the others).
enum WalkSource { kInKernel, kSorted, kJobOrder };
// the sequential walk's LDS
template <int LOG2, bool LDS_STATES>
struct WalkLds
{
    // the cooperative scan's arrays (job-order form) are dead before the walk's first LDS write: they share its memory (a workgroup is ONE wavefront)
    __attribute__((aligned(16))) unsigned char walkMem[sizeof(WalkShared) > sizeof(ScanShared) ? sizeof(WalkShared) : sizeof(ScanShared)];
    BlockTables<64, 2 * LOG2, LDS_STATES> bt;
};
template <int LOG2, WalkSource SOURCE, bool LDS_STATES>
__device__ __forceinline__ void walkBody(WalkLds<LOG2, LDS_STATES> &lds, int wg, int16_t *__restrict__ dstAll, const int16_t *__restrict__ srcAll,
                                         const uint8_t *__restrict__ statesAll, const RdoqJob *__restrict__ jobs, int njobs, int32_t *__restrict__ cbfOut, int bitDepth,
                                         const RdoqWork *__restrict__ work, int onlyOther, int masksByScan = 0)
{
    constexpr bool SORTED = SOURCE == kSorted;
    typedef LaneBlock<LOG2> LB;
    constexpr int G = LB::G, gw = LB::gw;
    unsigned char *walkMem = lds.walkMem;
    WalkShared &sh = *reinterpret_cast<WalkShared *>(walkMem);
    BlockTables<64, 2 * LOG2, LDS_STATES> &bt = lds.bt;
    if (SORTED && onlyOther && work->otherScans == 0) return;
    const RdoqInfo *infoAll = reinterpret_cast<const RdoqInfo *>(reinterpret_cast<const char *>(work) + rdoqInfoOffset());
    const uint32_t *order = reinterpret_cast<const uint32_t *>(infoAll + njobs);
    const int lane = threadIdx.x, slot = wg * 64 + lane;
    bool valid = slot < njobs;
    const int blk = valid ? (SORTED ? (int)order[slot] : slot) : 0;
    const RdoqJob job = jobs[blk];
    if (onlyOther && job.scan_idx == 0) valid = false;
    if (SOURCE == kJobOrder && onlyOther && __ballot(valid) == 0) return;      // (the usual case: the reference's encoder scans these sizes diagonally)
    RdoqInfo info;
    if (SORTED)
        info = infoAll[blk];      // (masks by scan position: k_rdoq_order)
    else if (SOURCE == kJobOrder)
    {
        info = infoAll[blk];
        if (!masksByScan) toScanOrder<gw>(job.scan_idx, info.mask, info.mask2, info.mask3);      // (a sorted launch's k_rdoq_order has done it)
    }
    else
    {
        ScanShared &sc = *reinterpret_cast<ScanShared *>(walkMem);
        scanBlocks<LOG2>(sc, dstAll, srcAll, jobs, njobs, wg * 64, G);
        info.mask = sc.mask[lane];
        info.mask2 = info.mask3 = 0;
        info.sumSq = sc.sumSq[lane];
        toScanOrder<gw>(job.scan_idx, info.mask, info.mask2, info.mask3);
        __syncthreads();
    }
    RT_DECL;
    LB lb;
    lb.stageIn(sh, bt, lane, lane, 0, 1, job, statesAll, bitDepth, srcAll, dstAll);
    RT_MARK(0);
    const Block &b = lb.b;
    const uint8_t *rasterOf = lb.rasterOf;
    SdhAux aux;

    uint64_t nz = valid ? info.mask : 0;      // groups to walk, by scan position
    int g = nz ? 63 - __clzll((long long)nz) : -1;
    const int firstGroup = g;
    int firstPos = -1;
    if (firstGroup >= 0)
    {
        const int p = rasterOf[firstGroup];
        lb.loadGroup(sh, lane, p & (gw - 1), p / gw);
        firstPos = firstGroup * 16 + lb.firstInGroup(sh, lane);
        nz |= 1;      // the DC group is always walked in full (it is coded whatever its levels, Rdoq.cpp:291-295)
    }

    WalkState ws;
    int carry = 0;
    while (__ballot(g >= 0) != 0)
    {
        if (g >= 0)
        {
            const int p = rasterOf[g], gx = p & (gw - 1), gy = p / gw;
            RT_MARK(2);
            lb.loadGroup(sh, lane, gx, gy);
            const WalkResult r = walkGroup<LOG2>(b, sh, lane, g, gx, gy, firstPos, LB::caseOf(ws.coded, gx, gy, carry), job.sdh_factor, aux RT_ARG);
            ws.walkedGroup(r, g, p, carry, lb.scanOf, gw);
            // as a group below the last one (the last one is redone below) -- but for the DC group, the last to be walked: its records are
            // still there when the verdict is known, so it is finished then, once, as what it turns out to be
            RT_MARK(8);
            if (r.coded && g != 0) lb.finishGroup(sh, lane, job.sdh, aux, g, 1 << 30, false, gx, gy);
            RT_MARK(9);
            // the next group to walk, and the all-zero groups on the way to it: one flag cost each, no carry across them
            const uint64_t below = nz & ((1ull << g) - 1);
            const int next = below ? 63 - __clzll((long long)below) : -1;
            if (G > 1) ws.zeroGroups(((1ull << g) - 1) & ~((2ull << max(next, 0)) - 1) & ~below, b);
            carry = next == g - 1 ? r.carry : 0;
            g = next;
        }
    }
    RT_MARK(2);

    // ---- the block's verdict (Rdoq.cpp:307-341, :401-441) ----
    int cbf = 0;
    if (firstPos >= 0)
    {
        const int lastIdx = lb.lastIndex(job.is_intra, info.sumSq, ws.walkedDist0, ws.costTu, ws.bestRel, ws.bestPos);
        cbf = lastIdx ? ws.orSince : 0;
        const int lastGroup = (lastIdx - 1) >> 4;      // -1: nothing is coded
        for (uint64_t m = ws.codedScan & ~((2ull << max(lastGroup, 0)) - 1); m;)   // groups above the last one were written as if coded: clear them
        {
            const int k = 63 - __clzll((long long)m);
            m ^= 1ull << k;
            lb.clearGroup(rasterOf[k]);
        }
        if (lastGroup >= 0) lb.finishGroup(sh, lane, job.sdh, aux, 0, lastGroup == 0 ? lastIdx : 1 << 30, lastGroup == 0, 0, 0);      // the DC group, from the records of its walk
        // the group holding the last significant coefficient: levels again, truncated, hidden with the last-group rules
        if (lastGroup > 0 && (lastIdx & 15 || job.sdh))
        {
            const int p = rasterOf[lastGroup], gx = p & (gw - 1), gy = p / gw;
            lb.loadGroup(sh, lane, gx, gy);
            walkGroup<LOG2>(b, sh, lane, lastGroup, gx, gy, firstPos, LB::caseOf(ws.coded, gx, gy, (int)(ws.carries >> lastGroup) & 1), job.sdh_factor, aux RT_ARG);
            lb.finishGroup(sh, lane, job.sdh, aux, lastGroup, lastIdx, true, gx, gy);
        }
    }
    if (valid) cbfOut[blk] = cbf;
    RT_MARK(11);
    RT_FLUSH(lane, 0);
#ifdef HAVOC_RDOQ_TIMING
    if (lane == 0) atomicAdd(&g_rdoqTiming[15], 1ull);
#endif
}

template <int LOG2, WalkSource SOURCE, bool LDS_STATES>
__global__ __launch_bounds__(64, 2) void k_rdoq_walk(int16_t *__restrict__ dstAll, const int16_t *__restrict__ srcAll, const uint8_t *__restrict__ statesAll,
                                                  const RdoqJob *__restrict__ jobs, int njobs, int32_t *__restrict__ cbfOut, int bitDepth,
                                                  const RdoqWork *__restrict__ work, int onlyOther)
{
    __shared__ WalkLds<LOG2, LDS_STATES> lds;
    walkBody<LOG2, SOURCE, LDS_STATES>(lds, blockIdx.x, dstAll, srcAll, statesAll, jobs, njobs, cbfOut, bitDepth, work, onlyOther);
}

// ---------------------------------------------------------------------------------------------------------------------
// The diagonal walk (32x32 and 16x16 blocks with the up-right diagonal scan: all of them in the reference's encoder, which uses
// the other scans for 4x4 / 8x8 intra blocks only).  A group needs of the rest of the block: whether the groups to its RIGHT and
// BELOW are coded -- both on the next anti-diagonal of groups -- and the carry of the group before it in scan order, which with
// this scan is its neighbour on the SAME anti-diagonal (or the end of the next one).  So the groups of an anti-diagonal can be
// walked side by side once the next anti-diagonal is final, if the carry is known.  The carry a group leaves is "one of its
// kept levels is > 1", and a kept level is the rounded level, that minus one, or (rounded levels 1 and 2 only) zero: a group
// without a rounded level > 1 leaves 0, one with a rounded level > 2 leaves 1 -- the scan pass tells which -- and only a group
// whose largest rounded level is exactly 2 is undecided.  The group after such a one is walked TWICE, on two lanes, once for each
// carry, and the right lane is picked when the round is over; nothing is ever walked again.
// A block with 14 groups to walk on the anti-diagonals 0 .. 4 takes 5 rounds instead of 14 steps; the sequential chain is what
// bounds the launch (the machine is far from full).
//
// LPB lanes per block, 64 / LPB blocks per wavefront: in a round the lanes of a block take the next groups to walk of the current
// anti-diagonal, one or two lanes each.  Every lane of a block keeps the block's running state (it is small, and identical in
// all of them): after a round the lanes exchange their groups' results through LDS and each replays them in scan order.
// ---------------------------------------------------------------------------------------------------------------------
struct DiagExchange
{
    int64_t cost[64], sigCost[64], dist0[64], q[64], localBest[64];
    int32_t localPos[64], localOr[64], groupOr[64];
    int32_t flags[64];      // localStop | coded << 1 | carry out << 2 | carry in << 3
};

// sorted: blocks in the order of pass 2, masks by scan position (k_rdoq_order); otherwise (round 6: a launch whose wavefronts are all resident at once -- see WalkSource)
// blocks in job order, masks as the scan left them.  The workgroups past `diagWgs` walk the blocks that do not use the diagonal scan (a lane per block, job
// order: they look at their 64 jobs and leave when there is none -- the reference's encoder has none at these sizes; round 6: before, a launch of its own behind
// this one, 12 us of every picture's critical path for nothing).
template <int LOG2, int LPB>
__global__ __launch_bounds__(64, 2) void k_rdoq_diag(int16_t *__restrict__ dstAll, const int16_t *__restrict__ srcAll, const uint8_t *__restrict__ statesAll,
                                                  const RdoqJob *__restrict__ jobs, int njobs, int32_t *__restrict__ cbfOut, int bitDepth,
                                                  const RdoqWork *__restrict__ work, int sorted, int diagWgs)
{
    struct DiagLds
    {
        WalkShared sh;
        BlockTables<64 / LPB, 2 * LOG2, true> bt;
        DiagExchange ex;
    };
    __shared__ union { DiagLds d; WalkLds<LOG2, false> w; } mem;      // (a workgroup is one or the other)
    if ((int)blockIdx.x >= diagWgs)
    {
        walkBody<LOG2, kJobOrder, false>(mem.w, blockIdx.x - diagWgs, dstAll, srcAll, statesAll, jobs, njobs, cbfOut, bitDepth, work, 1, sorted);
        return;
    }
    typedef LaneBlock<LOG2> LB;
    constexpr int G = LB::G, gw = LB::gw, ND = 2 * gw - 1;
    WalkShared &sh = mem.d.sh;
    BlockTables<64 / LPB, 2 * LOG2, true> &bt = mem.d.bt;
    DiagExchange &ex = mem.d.ex;
    const RdoqInfo *infoAll = reinterpret_cast<const RdoqInfo *>(reinterpret_cast<const char *>(work) + rdoqInfoOffset());
    const uint32_t *order = reinterpret_cast<const uint32_t *>(infoAll + njobs);
    const int lane = threadIdx.x, k = lane & (LPB - 1), lane0 = lane & ~(LPB - 1), slot = blockIdx.x * (64 / LPB) + lane / LPB;
    bool valid = slot < njobs;
    const int blk = valid ? (sorted ? (int)order[slot] : slot) : 0;
    const RdoqJob job = jobs[blk];
    valid = valid && job.scan_idx == 0;
    RdoqInfo info = infoAll[blk];
    RT_DECL;
    LB lb;
    lb.stageIn(sh, bt, lane, lane / LPB, k, LPB, job, statesAll, bitDepth, srcAll, dstAll);
    RT_MARK(0);
    const Block &b = lb.b;
    const uint8_t *rasterOf = lb.rasterOf;
    SdhAux aux;
    if (!sorted)      // masks by raster position -> by scan position: the block's LPB lanes take every LPB-th group and put their parts together
    {
        uint64_t m[3] = {0, 0, 0};
        for (int g = k; g < G; g += LPB)
        {
            const int p = rasterOf[g];
            m[0] |= ((info.mask >> p) & 1) << g;
            m[1] |= ((info.mask2 >> p) & 1) << g;
            m[2] |= ((info.mask3 >> p) & 1) << g;
        }
#pragma unroll
        for (int q = 0; q < 3; ++q)
        {
            uint32_t lo = (uint32_t)m[q], hi = (uint32_t)(m[q] >> 32);
#pragma unroll
            for (int d = 1; d < LPB; d <<= 1)
            {
                lo |= (uint32_t)__shfl_xor((int)lo, d, kWave);
                hi |= (uint32_t)__shfl_xor((int)hi, d, kWave);
            }
            m[q] = (uint64_t)hi << 32 | lo;
        }
        info.mask = m[0];
        info.mask2 = m[1];
        info.mask3 = m[2];
    }

    // by SCAN index: groups to walk; groups that leave carry 1 for sure; groups whose carry is not known before they are walked
    uint64_t walkScan = valid ? info.mask : 0;      // (the workspace holds the masks by scan position: k_rdoq_order)
    const uint64_t sureScan = valid ? info.mask3 : 0, openScan = valid ? info.mask2 & ~info.mask3 : 0;
    const int firstGroup = walkScan ? 63 - __clzll((long long)walkScan) : -1;
    int firstPos = -1;
    if (firstGroup >= 0)
    {
        const int p = rasterOf[firstGroup];
        lb.loadGroup(sh, lane, p & (gw - 1), p / gw);
        firstPos = firstGroup * 16 + lb.firstInGroup(sh, lane);
        walkScan |= 1;      // the DC group is always walked in full
    }

    WalkState ws;
    uint64_t carryOut = 0;      // carry each walked group left, by scan index
    int gAcc = firstGroup;      // next group (scan index, going down) the running state has not seen yet
    auto hopZeros = [&](int lowest) {      // the groups lowest .. gAcc are all-zero ones
        if (gAcc < lowest) return;
        ws.zeroGroups(((2ull << gAcc) - 1) & ~((1ull << lowest) - 1), b);
        gAcc = lowest - 1;
    };

    // The wavefront goes down the anti-diagonals together.  (Letting each block go down ITS anti-diagonals saves the rounds a block spends
    // idle on an anti-diagonal only others use, and measured SLOWER, 0.24 against 0.21 ms: a round costs what its longest group costs,
    // groups of one anti-diagonal are alike -- dense near DC, one or two levels far from it -- and mixing them makes every round a long one.)
    RT_MARK(1);
    for (int d = ND - 1; d >= 0; --d)
    {
        const int start = d < gw ? d * (d + 1) / 2 : G - (ND - d) * (ND - d + 1) / 2, len = d < gw ? d + 1 : ND - d;
        if (__ballot(firstGroup >= start) == 0) continue;      // no block of the wavefront reaches this anti-diagonal
        uint64_t m = walkScan & (((1ull << len) - 1) << start);      // groups of the anti-diagonal still to walk
        while (__ballot(m != 0))      // rounds
        {
            // the round's picks, highest scan index first: group, first lane, lanes (2 = both carries)
            int pickG[LPB], pickAt[LPB], pickN[LPB];
#pragma unroll
            for (int i = 0; i < LPB; ++i)
            {
                pickG[i] = -1;
                pickAt[i] = pickN[i] = 0;
            }
            int used = 0, previous = -2;
            int myG = -1, carryIn = 0;
            bool full = false;
#pragma unroll
            for (int i = 0; i < LPB; ++i)
            {
                if (!m || full) continue;
                const int g = 63 - __clzll((long long)m);
                int need = 1, carry = 0;
                if (g < firstGroup && ((walkScan >> (g + 1)) & 1))
                {
                    if (g + 1 != previous) carry = (int)(carryOut >> (g + 1)) & 1;          // walked in an earlier round
                    else if ((openScan >> (g + 1)) & 1) need = 2;                           // walked in this round and undecided
                    else carry = (int)(sureScan >> (g + 1)) & 1;
                }
                if (used + need > LPB)      // the rest waits for the next round
                {
                    full = true;
                    continue;
                }
                pickG[i] = g;
                pickAt[i] = used;
                pickN[i] = need;
                if (k >= used && k < used + need)
                {
                    myG = g;
                    carryIn = need == 2 ? k - used : carry;
                }
                used += need;
                previous = g;
                m ^= 1ull << g;
            }
            const bool mine = myG >= 0;
            const int p = mine ? rasterOf[myG] : 0, gx = p & (gw - 1), gy = p / gw;
            WalkResult r;
            r.coded = 0;
            RT_MARK(2);
            if (mine)
            {
                lb.loadGroup(sh, lane, gx, gy);
                r = walkGroup<LOG2>(b, sh, lane, myG, gx, gy, firstPos, LB::caseOf(ws.coded, gx, gy, carryIn), job.sdh_factor, aux RT_ARG);
                ex.flags[lane] = (int)r.localStop | r.coded << 1 | r.carry << 2 | carryIn << 3;
                ex.cost[lane] = r.cost;
                ex.sigCost[lane] = r.sigCost;
                ex.dist0[lane] = r.dist0;
                ex.q[lane] = r.q;
                ex.localBest[lane] = r.localBest;
                ex.localPos[lane] = r.localPos;
                ex.localOr[lane] = r.localOr;
                ex.groupOr[lane] = r.groupOr;
            }
            __syncthreads();
            RT_MARK(7);
            // every lane of the block replays the round in scan order, taking of a group walked twice the lane whose carry was right
            int carry = 0;
            bool chosen = false;
#pragma unroll
            for (int i = 0; i < LPB; ++i)
            {
                if (pickG[i] < 0) continue;
                const int from = lane0 + pickAt[i] + (pickN[i] == 2 ? carry : 0), f = ex.flags[from];
                chosen |= from == lane;
                hopZeros(pickG[i] + 1);
                WalkResult o;
                o.cost = ex.cost[from];
                o.sigCost = ex.sigCost[from];
                o.dist0 = ex.dist0[from];
                o.q = ex.q[from];
                o.localBest = ex.localBest[from];
                o.localPos = ex.localPos[from];
                o.localOr = ex.localOr[from];
                o.groupOr = ex.groupOr[from];
                o.localStop = f & 1;
                o.coded = (f >> 1) & 1;
                o.carry = carry = (f >> 2) & 1;
                ws.walkedGroup(o, pickG[i], rasterOf[pickG[i]], (f >> 3) & 1, lb.scanOf, gw);
                carryOut |= (uint64_t)o.carry << pickG[i];
                gAcc = pickG[i] - 1;
            }
            RT_MARK(8);
            if (chosen && r.coded && myG != 0) lb.finishGroup(sh, lane, job.sdh, aux, myG, 1 << 30, false, gx, gy);      // as a group below the last one; the DC group waits for the verdict
            RT_MARK(9);
    __syncthreads();      // the exchange arrays are free again
            RT_MARK(10);
        }
        if (firstGroup >= start) hopZeros(start);
    }

    // ---- the block's verdict, by the block's first lane ----
    if (valid && k == 0)
    {
        int cbf = 0;
        if (firstPos >= 0)
        {
            const int lastIdx = lb.lastIndex(job.is_intra, info.sumSq, ws.walkedDist0, ws.costTu, ws.bestRel, ws.bestPos);
            cbf = lastIdx ? ws.orSince : 0;
            const int lastGroup = (lastIdx - 1) >> 4;
            for (uint64_t m = ws.codedScan & ~((2ull << max(lastGroup, 0)) - 1); m;)
            {
                const int g = 63 - __clzll((long long)m);
                m ^= 1ull << g;
                lb.clearGroup(rasterOf[g]);
            }
            if (lastGroup >= 0) lb.finishGroup(sh, lane, job.sdh, aux, 0, lastGroup == 0 ? lastIdx : 1 << 30, lastGroup == 0, 0, 0);      // the DC group: this lane walked it last
            if (lastGroup > 0 && (lastIdx & 15 || job.sdh))
            {
                const int p = rasterOf[lastGroup], gx = p & (gw - 1), gy = p / gw;
                lb.loadGroup(sh, lane, gx, gy);
                walkGroup<LOG2>(b, sh, lane, lastGroup, gx, gy, firstPos, LB::caseOf(ws.coded, gx, gy, (int)(ws.carries >> lastGroup) & 1), job.sdh_factor, aux RT_ARG);
                lb.finishGroup(sh, lane, job.sdh, aux, lastGroup, lastIdx, true, gx, gy);
            }
        }
        cbfOut[blk] = cbf;
    }
    RT_MARK(11);
    RT_FLUSH(lane, 16);
#ifdef HAVOC_RDOQ_TIMING
    if (lane == 0) atomicAdd(&g_rdoqTiming[31], 1ull);
#endif
}
} // namespace

#ifdef HAVOC_RDOQ_TIMING
extern "C" __attribute__((visibility("default"))) int havoc_mi355x_debug_rdoq_timing(unsigned long long *out, int reset)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rdoqTiming), sizeof(g_rdoqTiming)) != hipSuccess) return -2;
    if (reset)
    {
        unsigned long long z[32] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_rdoqTiming), z, sizeof(z)) != hipSuccess) return -3;
    }
    return 0;
}
#endif

size_t rdoq_workspace_bytes(int njobs) { return rdoqInfoOffset() + (size_t)max(njobs, 0) * (sizeof(RdoqInfo) + sizeof(uint32_t)) + 64; }

// diagnostic A/B switches (profiles/): HAVOC_RDOQ_DIAG = lanes per block of the anti-diagonal walk of 32x32 blocks (0: off, 4, 8), HAVOC_RDOQ_DIAG16 the same for 16x16
// blocks, HAVOC_RDOQ_SORT = 1: histogram + counting sort whatever the launch's size (tests/test_rdoq.py runs its cases both ways)
static int envInt(const char *name, int otherwise) { const char *v = getenv(name); return v ? atoi(v) : otherwise; }

// The diagonal walk of 32x32 blocks shortens a block's chain, not the work: 16 blocks to a wavefront instead of 64, most lanes idle -- 28 M vector instructions for a
// 1080p picture's 14.7 k blocks where the sequential sorted walk issues ~8 M.  Alone the launch is shorter (0.110 against 0.135 ms) and rounds 2-5 chose it for that;
// but the STEP is bound by vector-instruction issue (the kernels' VALU-busy time adds up to ~0.47 of its 0.69 ms), and there the sequential walk wins: same box,
// alternating, three runs each (gpu call r06s): step 0.697 / 0.699 / 0.696 ms with 4 lanes per block, 0.688 / 0.676 / 0.685 sequential, 0.730-0.741 with 8 lanes; one picture
// in flight 0.719 -> 0.690 ms.  (At 4K it always lost: 2.43 -> 2.63 ms forced.)  So since round 6 the default is the sequential walk; HAVOC_RDOQ_DIAG = 4 / 8 selects the
// diagonal walk for a caller that wants one launch's latency (<= 16 k blocks).
// 16x16 blocks (round 3): the anti-diagonal walk is instantiated for them too (parity: tests/test_rdoq.py) and MEASURED SLOWER than a lane per
// block -- 1080p QP32, 39 k blocks: 0.098 ms sequential, 0.157 ms with 4 lanes, 0.228 ms with 8; 4K QP27, 156 k blocks: 0.33 -> 0.82 ms (profiles/r03/rdoq_*_diag16_*.json):
// a 16x16 block walks 3.2 groups on average and 7 at most, there is no chain to shorten.  Default: 0.
static int diagLanes(int log2, int njobs)
{
    static const int diagEnv = envInt("HAVOC_RDOQ_DIAG", 0), diag16Env = envInt("HAVOC_RDOQ_DIAG16", 0);
    return log2 == 5 ? (diagEnv == 0 || njobs > 16 * 1024 ? 0 : (diagEnv == 8 ? 8 : 4)) : (diag16Env == 0 ? 0 : (diag16Env == 8 ? 8 : 4));
}
// Round 6: is the launch small enough that all its walk wavefronts are resident at once (one per SIMD, 1 024 SIMDs)?  Then it lasts as long as its longest wavefront
// whichever blocks share one, and the histogram + counting sort (memset, k_rdoq_hist / the scan's histogram, k_rdoq_order: three dependent launches on the picture's
// critical path) are left out: blocks in job order.
static bool inJobOrder(int log2, int njobs)
{
    // MEASURED (gpu call r06n, 1080p QP32): alone, the 32x32 launch 0.115 -> 0.110 ms and the five sizes side by side 0.248 -> 0.200 ms; but a 16x16 launch in job order
    // takes 0.078 instead of 0.068 ms (every wavefront now walks as many groups as its densest block), and the STEP -- two pictures in flight, bound by instruction
    // issue and wavefront slots, not by one chain's latency -- goes 0.687 -> 0.711 ms.  So: sorted by default (HAVOC_RDOQ_SORT: 0 = job order where it applies,
    // 2 = job order for 32x32 only, 3 = for 16x16 only; tests/test_rdoq.py runs both orders).
    static const int sortEnv = envInt("HAVOC_RDOQ_SORT", 1);
    if (sortEnv == 1 || (sortEnv == 2 && log2 == 4) || (sortEnv == 3 && log2 == 5)) return false;
    const int diag = diagLanes(log2, njobs);
    const long waves = diag ? ((long)njobs * diag + 63) / 64 : ((long)njobs + 63) / 64;
    return waves <= 1024;
}

template <int LOG2>
static void launchDiag(hipStream_t st, int lpb, int16_t *dst, const int16_t *src, const uint8_t *states, const RdoqJob *j, int njobs, int32_t *cbf, int bitDepth,
                       RdoqWork *work, int sorted)
{
    const int wgd = (njobs + 64 / lpb - 1) / (64 / lpb), tail = (njobs + 63) / 64;
    if (lpb == 4) hipLaunchKernelGGL((k_rdoq_diag<LOG2, 4>), dim3(wgd + tail), dim3(64), 0, st, dst, src, states, j, njobs, cbf, bitDepth, work, sorted, wgd);
    else hipLaunchKernelGGL((k_rdoq_diag<LOG2, 8>), dim3(wgd + tail), dim3(64), 0, st, dst, src, states, j, njobs, cbf, bitDepth, work, sorted, wgd);
}

// the passes after the scan.  Sorted: order, then the walk(s).  In job order: the walk alone
static hipError_t rdoq_order_and_walk(hipStream_t st, int bitDepth, int log2, int16_t *dst, const int16_t *src, const uint8_t *states, const RdoqJob *j, int njobs,
                                      int32_t *cbf, RdoqWork *work, bool jobOrder)
{
    const int wgs = (njobs + 63) / 64;
    const int diag = diagLanes(log2, njobs);
    if (jobOrder)
    {
        if (diag)
        {
            if (log2 == 5) launchDiag<5>(st, diag, dst, src, states, j, njobs, cbf, bitDepth, work, 0);
            else launchDiag<4>(st, diag, dst, src, states, j, njobs, cbf, bitDepth, work, 0);
        }
        else if (log2 == 4) hipLaunchKernelGGL((k_rdoq_walk<4, kJobOrder, false>), dim3(wgs), dim3(64), 0, st, dst, src, states, j, njobs, cbf, bitDepth, work, 0);
        else hipLaunchKernelGGL((k_rdoq_walk<5, kJobOrder, false>), dim3(wgs), dim3(64), 0, st, dst, src, states, j, njobs, cbf, bitDepth, work, 0);
        return hipGetLastError();
    }
    if (log2 == 4) hipLaunchKernelGGL(k_rdoq_order<4>, dim3((njobs + 255) / 256), dim3(256), 0, st, j, njobs, work);
    else hipLaunchKernelGGL(k_rdoq_order<5>, dim3((njobs + 255) / 256), dim3(256), 0, st, j, njobs, work);
    if (diag)      // (its last workgroups take the blocks of the other scans)
    {
        if (log2 == 5) launchDiag<5>(st, diag, dst, src, states, j, njobs, cbf, bitDepth, work, 1);
        else launchDiag<4>(st, diag, dst, src, states, j, njobs, cbf, bitDepth, work, 1);
    }
    // sorted blocks come from all over the picture: their context states stay a per-wavefront LDS copy (64 different cache lines per access otherwise)
    else if (log2 == 4) hipLaunchKernelGGL((k_rdoq_walk<4, kSorted, true>), dim3(wgs), dim3(64), 0, st, dst, src, states, j, njobs, cbf, bitDepth, work, 0);
    else hipLaunchKernelGGL((k_rdoq_walk<5, kSorted, true>), dim3(wgs), dim3(64), 0, st, dst, src, states, j, njobs, cbf, bitDepth, work, 0);
    return hipGetLastError();
}

hipError_t launch_rdoq(hipStream_t st, int bitDepth, int log2, int16_t *dst, const int16_t *src, const uint8_t *states, const void *jobs, int njobs, int32_t *cbf,
                       void *workspace)
{
    if (njobs <= 0) return hipSuccess;
    const RdoqJob *j = static_cast<const RdoqJob *>(jobs);
    RdoqWork *work = static_cast<RdoqWork *>(workspace);
    const int wgs = (njobs + 63) / 64;
    if (log2 <= 3)
    {
        // diagnostic A/B switch (profiles/): HAVOC_RDOQ_LDS_STATES=1 gives the job-order walks the per-wavefront LDS copy of the context states back
        static const bool ldsStates = getenv("HAVOC_RDOQ_LDS_STATES") && atoi(getenv("HAVOC_RDOQ_LDS_STATES")) != 0;
        if (ldsStates)
        {
            if (log2 == 2) hipLaunchKernelGGL((k_rdoq_walk<2, kInKernel, true>), dim3(wgs), dim3(64), 0, st, dst, src, states, j, njobs, cbf, bitDepth, work, 0);
            else hipLaunchKernelGGL((k_rdoq_walk<3, kInKernel, true>), dim3(wgs), dim3(64), 0, st, dst, src, states, j, njobs, cbf, bitDepth, work, 0);
        }
        else if (log2 == 2) hipLaunchKernelGGL((k_rdoq_walk<2, kInKernel, false>), dim3(wgs), dim3(64), 0, st, dst, src, states, j, njobs, cbf, bitDepth, work, 0);
        else hipLaunchKernelGGL((k_rdoq_walk<3, kInKernel, false>), dim3(wgs), dim3(64), 0, st, dst, src, states, j, njobs, cbf, bitDepth, work, 0);
        return hipGetLastError();
    }
    // Round 6: a launch of a few wavefronts (an intra picture's level, a decision step's candidates of one size: tens to hundreds of blocks) is ONE kernel, the scan
    // inside it, blocks in job order -- memset + scan + order + walk were four dependent launches for a few microseconds of work
    static const int tiny = envInt("HAVOC_RDOQ_TINY", 1024);
    if (njobs <= tiny)
    {
        // ... but for 32x32 blocks, whose walk is up to 64 groups one after the other (~11 us per group for a wavefront alone: 91 us per launch in an intra picture's
        // levels, profiles/r06/intra_chain_kernel_stats.csv): here one launch's LATENCY is what the caller waits for and the machine is empty, which is what the
        // diagonal walk is for -- scan + diagonal walk with 8 lanes per block, blocks in job order: an intra picture 0.323 -> 0.305 s (gpu call r06ad)
        static const int tinyDiag = envInt("HAVOC_RDOQ_TINY_DIAG", 8);
        if (log2 == 5 && (tinyDiag == 4 || tinyDiag == 8))
        {
            hipLaunchKernelGGL(k_rdoq_scan<5>, dim3((njobs + kScanSteps - 1) / kScanSteps), dim3(64), 0, st, dst, src, j, njobs, work, 0);
            launchDiag<5>(st, tinyDiag, dst, src, states, j, njobs, cbf, bitDepth, work, 0);
        }
        else if (log2 == 4) hipLaunchKernelGGL((k_rdoq_walk<4, kInKernel, false>), dim3(wgs), dim3(64), 0, st, dst, src, states, j, njobs, cbf, bitDepth, work, 0);
        else hipLaunchKernelGGL((k_rdoq_walk<5, kInKernel, false>), dim3(wgs), dim3(64), 0, st, dst, src, states, j, njobs, cbf, bitDepth, work, 0);
        return hipGetLastError();
    }
    const bool jobOrder = inJobOrder(log2, njobs);
    if (!jobOrder)
    {
        hipError_t e = hipMemsetAsync(work, 0, sizeof(RdoqWork), st);
        if (e != hipSuccess) return e;
    }
    if (log2 == 4) hipLaunchKernelGGL(k_rdoq_scan<4>, dim3((njobs + 4 * kScanSteps - 1) / (4 * kScanSteps)), dim3(64), 0, st, dst, src, j, njobs, work, jobOrder ? 0 : 1);
    else hipLaunchKernelGGL(k_rdoq_scan<5>, dim3((njobs + kScanSteps - 1) / kScanSteps), dim3(64), 0, st, dst, src, j, njobs, work, jobOrder ? 0 : 1);
    return rdoq_order_and_walk(st, bitDepth, log2, dst, src, states, j, njobs, cbf, work, jobOrder);
}

// Rdoq::runQuantisation for 16x16 / 32x32 blocks whose scan was done by havoc_mi355x_tu_forward_scan (RdoqInfo per block in the workspace, level
// blocks zeroed): histogram from the 32-byte records, order, walk
hipError_t launch_rdoq_prescanned(hipStream_t st, int bitDepth, int log2, int16_t *dst, const int16_t *src, const uint8_t *states, const void *jobs, int njobs,
                                  int32_t *cbf, void *workspace)
{
    if (njobs <= 0) return hipSuccess;
    if (log2 != 4 && log2 != 5) return hipErrorInvalidValue;
    const RdoqJob *j = static_cast<const RdoqJob *>(jobs);
    RdoqWork *work = static_cast<RdoqWork *>(workspace);
    if (inJobOrder(log2, njobs)) return rdoq_order_and_walk(st, bitDepth, log2, dst, src, states, j, njobs, cbf, work, true);
    hipError_t e = hipMemsetAsync(work, 0, sizeof(RdoqWork), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_rdoq_hist, dim3((njobs + 255) / 256), dim3(256), 0, st, j, njobs, work);
    return rdoq_order_and_walk(st, bitDepth, log2, dst, src, states, j, njobs, cbf, work, false);
}

} // namespace havoc_gpu
