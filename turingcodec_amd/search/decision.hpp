// decision.hpp -- the encoder's decision loops that CALL the havoc hot path, restated once over a small per-call
// interface so that the same code can be driven by
//   (a) a per-block implementation of the primitives (the reference's own table API; the tests also plug in their CPU checker) and
//   (b) the MI355X batch API, which serves the calls from super-set batches (SAD surfaces, phase planes) and replays
//       the loop on the host (batch_search.cpp).
// Identical decisions (motion vectors, costs, mode lists) from (a) and (b) is what makes the batched hot path a
// drop-in for these callers: SURVEY.md 8(f)-1.
//
// Restated from the reference (file:line under /root/reference/turing):
//   Cost / Lambda fixed point            Cost.h:33-34, FixedPoint.h:32-81
//   rateOf(mvd)                          Measure.h:177-220
//   MvCandidate (predictor choice, <)    Search.hpp:1252-1314
//   LimitFullPelMv                       Search.hpp:1366-1407
//   StateMeFullPel::considerPattern      Search.hpp:1447-1482
//   fullPelMotionEstimation              Search.hpp:2060-2336
//   costDistortionMv / costMv            Search.hpp:1963-2006
//   patternSearch / subPelRefinement     Search.hpp:2010-2061, 2340-2358
//   searchMotionUni                      Search.hpp:1317-1355
//   searchMotionBi                       Search.hpp:1498-1657
//   searchIntraPartition (SATD stage and candidate order)  Search.hpp:40-190
// Everything outside the primitives is integer arithmetic on 16-bit vector components and Q16 costs; the order of
// evaluation and the strict `<` comparisons are kept, because ties are decided by them.
//
// Since round 3 the same text is ALSO compiled for the device (csrc/kernels_search.hip: a workgroup per chain of searches runs these loops
// with the primitives computed by its own lanes), hence the HAVOC_HD marks and the constexpr tables.
#pragma once

#if defined(__HIPCC__)
#define HAVOC_HD __host__ __device__ __attribute__((always_inline))
#define HAVOC_UNROLL _Pragma("unroll")      // device: the four candidates of a pattern step live in registers, not in an indexed array
#else
#define HAVOC_HD
#define HAVOC_UNROLL
#endif

#include <cstdint>
#include <cstdlib>
#include <limits>

namespace havoc_search {

typedef int64_t Cost;                       // FixedPoint<int64_t, 16>
constexpr Cost kCostMax = 0x7fffffffffffffffll;

// a < b for the costs of the MOTION search, which are never negative (rates, lambda * distortion, kCostMax): on the device the 64-bit signed
// comparison has no scalar instruction and would drag the whole (wave-uniform) decision state into vector registers; the sign of the wrapped
// difference is the same answer for non-negative operands and stays scalar
HAVOC_HD inline bool costLess(Cost a, Cost b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t sign;      // through an opaque scalar shift: the compiler would fold any C++ spelling of "top bit set" back into the 64-bit comparison
    asm("s_lshr_b32 %0, %1, 31" : "=s"(sign) : "s"(__builtin_amdgcn_readfirstlane(uint32_t((uint64_t(a) - uint64_t(b)) >> 32))) : "scc");
    return sign != 0;
#else
    return a < b;
#endif
}

struct Lambda                               // FixedPoint<int32_t, 16>
{
    int32_t value = 0;
    HAVOC_HD void set(double d) { value = static_cast<int32_t>(d * (1 << 16) + 0.5); }   // FixedPoint::set(double)
    HAVOC_HD Cost operator*(int32_t y) const { return int64_t(value) * int64_t(y); }
};

struct Mv                                   // MotionVector: 16-bit components, quarter-sample units unless said otherwise
{
    int16_t x = 0, y = 0;
    HAVOC_HD constexpr Mv() {}
    HAVOC_HD constexpr Mv(int x_, int y_) : x(int16_t(x_)), y(int16_t(y_)) {}
    HAVOC_HD bool operator==(const Mv &o) const { return x == o.x && y == o.y; }
    HAVOC_HD bool operator!=(const Mv &o) const { return !(*this == o); }
};
HAVOC_HD inline Mv operator+(Mv a, Mv b) { return Mv(int16_t(a.x + b.x), int16_t(a.y + b.y)); }
HAVOC_HD inline Mv operator-(Mv a, Mv b) { return Mv(int16_t(a.x - b.x), int16_t(a.y - b.y)); }
HAVOC_HD inline Mv shr2(Mv a) { return Mv(int16_t(a.x >> 2), int16_t(a.y >> 2)); }
HAVOC_HD inline Mv shl2(Mv a) { return Mv(int16_t(a.x << 2), int16_t(a.y << 2)); }

// Measure.h:177-212: position of the highest set bit of |d|, 0 for 0 (both of the reference's branches -- the 32-step loop and
// 32 - lzcnt -- compute this; a count-leading-zeros with the zero case spelt out is the cheap form: this runs four times per candidate)
HAVOC_HD inline unsigned rateOfMvdComponent(int d)
{
    const unsigned u = unsigned(d < 0 ? -d : d);
    return u ? 32u - unsigned(__builtin_clz(u)) : 0u;
}
// Measure.h:214-220: Cost::make(r0 + r1 + 1, -1) = (r0 + r1 + 1) << 17
HAVOC_HD inline Cost rateOf(Mv mvd) { return Cost(rateOfMvdComponent(mvd.x) + rateOfMvdComponent(mvd.y) + 1) << 17; }

struct MvCandidate                          // Search.hpp:1252-1314
{
    Mv mv, mvd;
    Cost cost = kCostMax;
    int mvpFlag = 0;
    HAVOC_HD MvCandidate() {}
    // best of the two predictors for this vector; mvpRate[k] = rate of mvp_lX_flag == k in the current CABAC state
    HAVOC_HD MvCandidate(Mv mv_, const Mv predictors[2], const Cost mvpRate[2])
    {
        mvpFlag = 0;
        mvd = mv_ - predictors[0];
        cost = rateOf(mvd) + mvpRate[0];
        MvCandidate temp;
        temp.mvpFlag = 1;
        temp.mvd = mv_ - predictors[1];
        temp.cost = rateOf(temp.mvd) + mvpRate[1];
        consider(temp);
        mv = mv_;
    }
    HAVOC_HD bool consider(const MvCandidate &other)
    {
        const bool better = costLess(other.cost, cost);
        if (better) *this = other;
        return better;
    }
};

// what the loops read from encoder state
struct SearchParams
{
    int picWidth = 0, picHeight = 0;        // pic_width / height_in_luma_samples
    int ctbSize = 64;                       // CtbSizeY
    int concurrentFrames = 4;               // StateEncode::concurrentFrames (--concurrent-frames default, encode.cpp:151)
    bool met = true;                        // Speed::useMet(): medium and faster
    bool smallSearchWindow = false;         // Speed::useSmallSearchWindow(): fast and faster
    bool biSmallSearchWindow = false;       // Speed::useBiSmallSearchWindow()
    bool halfPel = true, quarterPel = true; // Speed::doHalfPelRefinement / doQuarterPelRefinement (medium: both)
    double reciprocalSqrtLambda = 0.0;      // StateEncodePicture::reciprocalSqrtLambda
    int bitDepth = 8;
};

// one prediction unit and the per-PU state the search reads
struct PuContext
{
    int x0 = 0, y0 = 0, w = 0, h = 0;       // prediction_unit
    int cuLog2Size = 0;                     // log2CbSize of the coding unit
    int cqtDepth = 0;
    bool part2Nx2N = true;
    int xCtb = 0, yCtb = 0;                 // CTU origin (LimitFullPelMv with concurrent frames)
    Mv mvp[2];                              // predictors->mvp[0][refList][0..1]
    Cost mvpRate[2] = {0, 0};               // EstimateRateBin<mvp_lX_flag>::rate(0 / 1)
    Mv mvPrevious2Nx2N;                     // stateEncodeSubstream->mvPreviousInteger2Nx2N[refList] (multiple of 4)
};

struct LimitFullPelMv                       // Search.hpp:1366-1407; full-sample units
{
    Mv lo, hi;
    HAVOC_HD LimitFullPelMv(const PuContext &pu, const SearchParams &sp)
    {
        lo = Mv(-sp.ctbSize - pu.x0, -sp.ctbSize - pu.y0);
        hi = Mv(sp.picWidth + sp.ctbSize - pu.x0 - pu.w, sp.picHeight + sp.ctbSize - pu.y0 - pu.h);
        if (sp.concurrentFrames > 1)
        {
            const int howCloseDoYouDare = 15;
            const int16_t wx = int16_t(pu.xCtb + 3 * sp.ctbSize - pu.x0 - pu.w - howCloseDoYouDare);
            const int16_t wy = int16_t(pu.yCtb + 2 * sp.ctbSize - pu.y0 - pu.h - howCloseDoYouDare);
            if (wx < hi.x) hi.x = wx;
            if (wy < hi.y) hi.y = wy;
        }
    }
    HAVOC_HD void operator()(Mv &mv) const
    {
        if (mv.x < lo.x) mv.x = lo.x;
        if (mv.y < lo.y) mv.y = lo.y;
        if (mv.x > hi.x) mv.x = hi.x;
        if (mv.y > hi.y) mv.y = hi.y;
    }
};

struct UniResult
{
    Mv mv, mvd;                             // after sub-sample refinement
    int mvpFlag = 0;
    Mv mvInteger;                           // best vector of the integer search (quarter units, multiple of 4)
    Cost costInteger = kCostMax;            // MvCandidate::cost of that vector
    Cost costSubPel = kCostMax;             // patternSearch's bestCost
    Cost costMvdZero[2] = {kCostMax, kCostMax};
    bool wrote2Nx2N = false;                // mvPreviousInteger2Nx2N was updated (to mvInteger)
    int calls = 0;                          // primitive calls the loop made (SAD, SAD4, interpolate + SATD)
};

// View = the per-call interface of ONE (PU, reference list) pair:
//     int  sad(int dx, int dy)                         havoc_sad of the source block against ref(x0 + dx, y0 + dy)
//     void sad4(const Mv d[4], int32_t out[4])         havoc_sad_multiref, four full-sample displacements
//     int  satdQpel(Mv mv)                             HavocPredUni at the quarter-sample vector + measureSatd
// (b)-type views may throw to ask for a replay once the missing data has been computed; the loops hold no state
// outside their arguments, so a replay is just a second call.
// a view that can evaluate several sub-sample positions at once (the device view: one wavefront per position) is told which ones the next
// costMv calls will ask for; other views ignore it
template <class View>
HAVOC_HD inline auto hintSatd(View &v, const Mv *positions, int n, int) -> decltype(v.hintSatd(positions, n), void()) { v.hintSatd(positions, n); }
template <class View>
HAVOC_HD inline void hintSatd(View &, const Mv *, int, long) {}

// the same for the full-sample positions of a rectangle of displacements (searchMotionBi's exhaustive grid)
template <class View>
HAVOC_HD inline auto hintSadRect(View &v, int x0, int y0, int x1, int y1, int) -> decltype(v.hintSadRect(x0, y0, x1, y1), void()) { v.hintSadRect(x0, y0, x1, y1); }
template <class View>
HAVOC_HD inline void hintSadRect(View &, int, int, int, int, long) {}

// A view whose lanes can do more than answer calls takes over whole steps of the loops (the device view, csrc/kernels_search.hip: the four candidates of a
// pattern step / the eight or nine of a sub-sample step are costed one per LANE with the arithmetic below, where this text costs them one after the other in
// scalar registers -- measured: 28 k scalar instructions per search, the kernel's bound).  Same candidates, same order of preference (the first of equal
// costs), same results: the trace-pin tests hold both forms against the reference encoder's own decisions.
//   bool patternStep(Mv, Mv, Mv, Mv, const PuContext &, Lambda, MvCandidate &best)             = sad4 + the four best.consider() of considerPattern
//   int  subpelStep(Mv mv, Mv mvd, int scale, bool tryOrigin, Lambda, Cost &bestCost)           = patternSearchOnce's costMv calls; returns bestI or -1
template <class View, class Search>
HAVOC_HD inline auto foldPatternStep(View &v, const Mv (&mv)[4], Search &s, int) -> decltype(v.patternStep(mv[0], mv[1], mv[2], mv[3], s.pu, s.lambda, s.best), bool())
{
    ++s.calls;
    return v.patternStep(mv[0], mv[1], mv[2], mv[3], s.pu, s.lambda, s.best);      // by value: an array handed down by reference ends up in (per-lane) memory
}
template <class View, class Search>
HAVOC_HD inline bool foldPatternStep(View &v, const Mv (&mv)[4], Search &s, long)
{
    int32_t sads[4];
    v.sad4(mv, sads);
    ++s.calls;
    bool improved = false;
    HAVOC_UNROLL
    for (int i = 0; i < 4; ++i)
    {
        MvCandidate candidate(shl2(mv[i]), s.pu.mvp, s.pu.mvpRate);
        candidate.cost += s.lambda * sads[i];
        improved |= s.best.consider(candidate);
    }
    return improved;
}
//   int  patternRing(Mv origin, const Mv *pattern, int n, int step, int dist, limit, pu, lambda, best)  = a whole considerPattern (4, 8 or 16 candidates): 0 / 1
//   void rasterSweep(int rasterSearch, limit, pu, lambda, best)                                         = the raster refinement's 175 (52) considerPattern calls
template <class View, class Search>
HAVOC_HD inline auto foldPatternRing(View &v, Mv origin, const Mv *pattern, int n, int step, int dist, Search &s, int)
    -> decltype(v.patternRing(origin, pattern, n, step, dist, s.limit, s.pu, s.lambda, s.best), int())
{
    s.calls += n / (4 * step);
    return v.patternRing(origin, pattern, n, step, dist, s.limit, s.pu, s.lambda, s.best);
}
template <class View, class Search>
HAVOC_HD inline int foldPatternRing(View &, Mv, const Mv *, int, int, int, Search &, long) { return -1; }
template <class View, class Search>
HAVOC_HD inline auto foldRasterSweep(View &v, int rasterSearch, Search &s, int) -> decltype(v.rasterSweep(rasterSearch, s.limit, s.pu, s.lambda, s.best), bool())
{
    const int rows = 2 * rasterSearch / 20 + 1, groups = 2 * rasterSearch / 80 + 1;
    s.calls += rows * groups;
    v.rasterSweep(rasterSearch, s.limit, s.pu, s.lambda, s.best);
    return true;
}
template <class View, class Search>
HAVOC_HD inline bool foldRasterSweep(View &, int, Search &, long) { return false; }

//   int  startProbe(Mv mvQuarter, int forcedFlag, bool met, bool hexagon, limit, pu, lambda, best, Cost *costOut, int &calls)
//                                                   = a start candidate of fullPel AND the early-termination probe around it (its SAD, the diamond's four and the
//                                                     hexagon's eight in one exchange): 1 = the search ends here
template <class View, class Search>
HAVOC_HD inline auto foldStartProbe(View &v, Mv mv, int forcedFlag, Cost *costOut, Search &s, int)
    -> decltype(v.startProbe(mv, forcedFlag, s.sp.met, s.pu.cuLog2Size >= 5, s.limit, s.pu, s.lambda, s.best, costOut, s.calls), int())
{
    return v.startProbe(mv, forcedFlag, s.sp.met, s.pu.cuLog2Size >= 5, s.limit, s.pu, s.lambda, s.best, costOut, s.calls);
}
template <class View, class Search>
HAVOC_HD inline int foldStartProbe(View &, Mv, int, Cost *, Search &, long) { return -1; }

//   bool biGrid(Mv originQuarter, int range, limit, pu, lambda, best)
//                                                   = the whole exhaustive grid of searchMotionBi ((2 range + 1)^2 candidates, their SADs announced with hintSadRect): a
//                                                     candidate per lane, the first of the cheapest folded into `best`; false = not taken over (the loops below run)
template <class View>
HAVOC_HD inline auto foldBiGrid(View &v, Mv origin, int range, const LimitFullPelMv &limit, const PuContext &pu, Lambda lambda, MvCandidate &best, int &calls, int)
    -> decltype(v.biGrid(origin, range, limit, pu, lambda, best), bool())
{
    if (!v.biGrid(origin, range, limit, pu, lambda, best)) return false;
    calls += (2 * range + 1) * ((2 * range + 1 + 3) / 4);
    return true;
}
template <class View>
HAVOC_HD inline bool foldBiGrid(View &, Mv, int, const LimitFullPelMv &, const PuContext &, Lambda, MvCandidate &, int &, long) { return false; }

template <class View, class Search>
HAVOC_HD inline auto foldSubpelStep(View &v, Search &s, int scale, bool tryOrigin, Mv mv, Mv mvd, Cost &bestCost, int) -> decltype(v.subpelStep(mv, mvd, scale, tryOrigin, s.lambda, bestCost), int())
{
    s.calls += tryOrigin ? 9 : 8;
    return v.subpelStep(mv, mvd, scale, tryOrigin, s.lambda, bestCost);
}

template <class View>
struct MotionSearch
{
    const SearchParams &sp;
    const PuContext &pu;
    View &view;
    LimitFullPelMv limit;
    Lambda lambda;
    MvCandidate best;
    int calls = 0;

    HAVOC_HD MotionSearch(const SearchParams &sp_, const PuContext &pu_, View &view_) : sp(sp_), pu(pu_), view(view_), limit(pu_, sp_)
    {
        lambda.set(sp.reciprocalSqrtLambda);
    }

    // Search.hpp:1447-1482.  origin in quarter units; pattern entries are multiplied by dist and divided by 4
    HAVOC_HD bool considerPattern(Mv origin, const Mv *pattern, int n, int step, int dist)
    {
        const int whole = foldPatternRing(view, origin, pattern, n, step, dist, *this, 0);
        if (whole >= 0) return whole != 0;
        bool improved = false;
        for (int j = 0; j < n; j += 4 * step)
        {
            Mv mv[4];
            HAVOC_UNROLL
            for (int i = 0; i < 4; ++i, pattern += step)
            {
                mv[i].x = int16_t((origin.x + dist * pattern->x) / 4);
                mv[i].y = int16_t((origin.y + dist * pattern->y) / 4);
                limit(mv[i]);
            }
            improved |= foldPatternStep(view, mv, *this, 0);
        }
        return improved;
    }

    // the early-termination probe after an improving start point (Search.hpp:2112-2124 and twice more)
    HAVOC_HD bool metTriggered()
    {
        static constexpr Mv diamond[4] = {{-4, 0}, {0, 4}, {4, 0}, {0, -4}};
        bool triggerMet = !considerPattern(best.mv, diamond, 4, 1, 1);
        if (triggerMet && pu.cuLog2Size >= 5)
        {
            static constexpr Mv hexagon[8] = {{0, -8}, {8, -4}, {8, 4}, {0, 8}, {-8, 4}, {-8, -4}, {-8, 4}, {-8, -4}};
            triggerMet = !considerPattern(best.mv, hexagon, 8, 1, 1);
        }
        return triggerMet;
    }

    HAVOC_HD int sadAt(Mv full)
    {
        ++calls;
        return view.sad(full.x, full.y);
    }

    // one start candidate of fullPel (Search.hpp:2100-2196: the zero vector, the two predictors, the previous 2Nx2N vector): its cost -- with predictor `forcedFlag`,
    // or the cheaper of the two when that is -1 --, and if it improves `best` and early termination is on, the probe around it.  true = the search ends here
    HAVOC_HD bool startCandidate(Mv mv, int forcedFlag, Cost *costOut)
    {
        const int whole = foldStartProbe(view, mv, forcedFlag, costOut, *this, 0);
        if (whole >= 0) return whole != 0;
        MvCandidate candidate;
        if (forcedFlag < 0)
            candidate = MvCandidate(mv, pu.mvp, pu.mvpRate);
        else
        {
            candidate.mvpFlag = forcedFlag;
            candidate.mv = mv;
            candidate.mvd = mv - pu.mvp[forcedFlag];
            candidate.cost = rateOf(candidate.mvd);
            candidate.cost += pu.mvpRate[forcedFlag];
        }
        candidate.cost += lambda * sadAt(shr2(mv));
        if (costOut) *costOut = candidate.cost;
        const bool better = best.consider(candidate);
        return better && sp.met && metTriggered();
    }

    // Search.hpp:2060-2336.  Returns true when mvPreviousInteger2Nx2N is to be updated with best.mv
    HAVOC_HD bool fullPel(Cost costMvdZero[2])
    {
        const int searchWindow = sp.smallSearchWindow ? 32 : 64;
        const int maxCounter = sp.smallSearchWindow ? 2 : 3;
        const int rasterSearch = sp.smallSearchWindow ? 120 : 240;
        // zero vector as a starting point (the position is NOT limited)
        if (startCandidate(Mv(0, 0), -1, nullptr)) return false;
        HAVOC_UNROLL
        for (int flag = 0; flag < 2; ++flag)
        {   // the two predictors, rounded to full samples; costed with THEIR predictor (not the cheaper of the two)
            Mv mv = shr2(Mv(int16_t(pu.mvp[flag].x + 1), int16_t(pu.mvp[flag].y + 1)));
            limit(mv);
            if (startCandidate(shl2(mv), flag, &costMvdZero[flag])) return false;
        }
        if (!pu.part2Nx2N || pu.cqtDepth != 0)
        {
            Mv mv = shr2(pu.mvPrevious2Nx2N);
            limit(mv);
            if (startCandidate(shl2(mv), -1, nullptr)) return false;
        }

        // HM style "star" search
        Mv mvStart = best.mv;
        int distBest = 0, counter = 0, step = 4;
        static constexpr Mv diamond[16] = {{0, -4}, {1, -3}, {2, -2}, {3, -1}, {4, 0}, {3, 1}, {2, 2}, {1, 3},
                                       {0, 4}, {-1, 3}, {-2, 2}, {-3, 1}, {-4, 0}, {-3, -1}, {-2, -2}, {-1, -3}};
        static constexpr Mv square4[4] = {{-4, -4}, {-4, 4}, {4, 4}, {4, -4}};
        for (int dist = 1; dist <= searchWindow && counter < maxCounter; dist <<= 1)
        {
            if (dist == 2 || dist == 8) step >>= 1;
            if (considerPattern(mvStart, diamond, 16, step, dist))
            {
                distBest = dist;
                counter = 0;
            }
            else
                ++counter;
        }
        if (distBest == 1)
        {
            distBest = 0;
            considerPattern(best.mv, square4, 4, 1, 1);
        }
        if (distBest > 5)
        {   // raster refinement: absolute positions, every 5th full sample
            static constexpr Mv line[4] = {{0, 0}, {1, 0}, {2, 0}, {3, 0}};
            if (!foldRasterSweep(view, rasterSearch, *this, 0))
                for (int my = -rasterSearch; my <= rasterSearch; my += 20)
                    for (int mx = -rasterSearch; mx <= rasterSearch; mx += 80) considerPattern(Mv(mx, my), line, 4, 1, 20);
            distBest = 5;
        }
        while (distBest > 0)
        {   // star refinement
            mvStart = best.mv;
            distBest = 0;
            step = 4;
            for (int dist = 1; dist <= searchWindow; dist <<= 1)
            {
                if (dist == 2 || dist == 8) step >>= 1;
                if (considerPattern(mvStart, diamond, 16, step, dist)) distBest = dist;
            }
            if (distBest == 1)
            {
                considerPattern(mvStart, square4, 4, 1, 1);
                distBest = 0;
            }
        }
        if (!sp.smallSearchWindow)
        {
            int j;
            do
            {
                static constexpr Mv diamond4[4] = {{0, -1}, {-1, 0}, {0, 1}, {1, 0}};
                Mv mv[4];
                HAVOC_UNROLL
                for (int i = 0; i < 4; ++i)
                {
                    mv[i] = Mv(int16_t(best.mv.x / 4), int16_t(best.mv.y / 4)) + diamond4[i];
                    limit(mv[i]);
                }
                j = foldPatternStep(view, mv, *this, 0) ? 0 : -1;      // the reference keeps the index of the last improving position; only "any" is used
            } while (j >= 0);
        }
        return pu.part2Nx2N;
    }

    // costMv, Search.hpp:2001-2006 (no mvp-flag rate here)
    HAVOC_HD Cost costMv(Mv mv, Mv mvd)
    {
        ++calls;
        return rateOf(mvd) + lambda * view.satdQpel(mv);
    }

    // Search.hpp:2010-2061 with maxIterations = 1, the only way subPelRefinement calls it.  pattern[i] = scale * the i-th of the eight neighbours in raster order
    HAVOC_HD void patternSearchOnce(const Mv (&pattern)[8], int scale, bool tryOrigin, Mv &mv, Mv &mvd, Cost &bestCost)
    {
        const int bestLanes = subpelDispatch(pattern, scale, tryOrigin, mv, mvd, bestCost, 0);
        if (bestLanes >= -1)
        {
            if (bestLanes >= 0)
            {
                mvd = mvd + pattern[bestLanes];
                mv = mv + pattern[bestLanes];
            }
            return;
        }
        {
            Mv ask[9];
            HAVOC_UNROLL
            for (int i = 0; i < 8; ++i) ask[i] = mv + pattern[i];
            ask[8] = mv;
            hintSatd(view, ask, tryOrigin ? 9 : 8, 0);
        }
        if (tryOrigin) bestCost = costMv(mv, mvd);
        int bestI = -1;
        for (int i = 0; i < 8; ++i)
        {
            const Cost cost = costMv(mv + pattern[i], mvd + pattern[i]);
            if (costLess(cost, bestCost))
            {
                bestI = i;
                bestCost = cost;
            }
        }
        if (bestI >= 0)
        {
            mvd = mvd + pattern[bestI];
            mv = mv + pattern[bestI];
        }
    }

    // -2: the view has no lane form (the loop above runs); otherwise the index of the winning neighbour or -1
    template <class V = View>
    HAVOC_HD auto subpelDispatch(const Mv (&)[8], int scale, bool tryOrigin, Mv mv, Mv mvd, Cost &bestCost, int) -> decltype(foldSubpelStep(*(V *)nullptr, *this, scale, tryOrigin, mv, mvd, bestCost, 0))
    {
        return foldSubpelStep(view, *this, scale, tryOrigin, mv, mvd, bestCost, 0);
    }
    HAVOC_HD int subpelDispatch(const Mv (&)[8], int, bool, Mv, Mv, Cost &, long) { return -2; }

    // searchMotionUni, Search.hpp:1317-1355
    // what the integer stage leaves behind: a caller that may have to run the sub-sample stage again (a batch client whose
    // sub-sample data was not there yet) keeps it and does not repeat the integer search
    struct IntegerStage
    {
        bool valid = false, wrote2Nx2N = false;
        MvCandidate best;
        int calls = 0;
        Cost costMvdZero[2] = {kCostMax, kCostMax};
    };

    HAVOC_HD UniResult run(IntegerStage *keep = nullptr)
    {
        UniResult r;
        if (keep && keep->valid)
        {
            best = keep->best;
            calls = keep->calls;
            r.wrote2Nx2N = keep->wrote2Nx2N;
            r.costMvdZero[0] = keep->costMvdZero[0];
            r.costMvdZero[1] = keep->costMvdZero[1];
        }
        else
        {
            r.wrote2Nx2N = fullPel(r.costMvdZero);
            if (keep)
            {
                keep->valid = true;
                keep->wrote2Nx2N = r.wrote2Nx2N;
                keep->best = best;
                keep->calls = calls;
                keep->costMvdZero[0] = r.costMvdZero[0];
                keep->costMvdZero[1] = r.costMvdZero[1];
            }
        }
        r.mvInteger = best.mv;
        r.costInteger = best.cost;
        r.mvpFlag = best.mvpFlag;
        Mv mv = best.mv, mvd = best.mvd;
        if (sp.halfPel)
        {
            static constexpr Mv half[8] = {{-2, -2}, {0, -2}, {2, -2}, {-2, 0}, {2, 0}, {-2, 2}, {0, 2}, {2, 2}};
            patternSearchOnce(half, 2, true, mv, mvd, r.costSubPel);
            if (sp.quarterPel)
            {
                static constexpr Mv quarter[8] = {{-1, -1}, {0, -1}, {1, -1}, {-1, 0}, {1, 0}, {-1, 1}, {0, 1}, {1, 1}};
                patternSearchOnce(quarter, 1, false, mv, mvd, r.costSubPel);
            }
        }
        r.mv = mv;
        r.mvd = mvd;
        r.calls = calls;
        return r;
    }
};

// ---- bi-directional refinement of one list against the other list's prediction (searchMotionBi, Search.hpp:1498-1657)
// BiView = the view of (PU, list being refined) whose source block is the "ideal" second predictor
// clip(2 * source - prediction from the other list) (havoc::SubtractBi); it is built by the caller from the other
// list's vector (limited like Search.hpp:1527-1530) and offers the same sad4 / satdQpel calls.
struct BiResult
{
    Mv mv, mvd;
    int mvpFlag = 0;
    Cost cost = kCostMax;
    int calls = 0;
};

template <class BiView>
HAVOC_HD BiResult searchMotionBi(const SearchParams &sp, const PuContext &pu, BiView &view, Mv startingMv)
{
    BiResult r;
    LimitFullPelMv limit(pu, sp);
    MvCandidate best;
    best.mv = shr2(Mv(int16_t(startingMv.x + 1), int16_t(startingMv.y + 1)));
    limit(best.mv);
    Mv mv1 = best.mv, mv2 = best.mv, mv3 = best.mv;
    best.mv = shl2(best.mv);
    best.cost = kCostMax;
    const Mv origin = best.mv;
    Lambda lambda;
    lambda.set(sp.reciprocalSqrtLambda * 0.5);
    const int range = sp.biSmallSearchWindow ? 1 : 5;
    {   // every full-sample position the grid below can ask for: a view that can evaluate them all at once is told (others ignore it)
        const Mv o = shr2(origin);
        hintSadRect(view, o.x - range, o.y - range, o.x + range + 3, o.y + range, 0);
    }
    if (!foldBiGrid(view, origin, range, limit, pu, lambda, best, r.calls, 0))
    for (int y = -range; y <= range; ++y)
        for (int xb = -range; xb <= range; xb += 4)      // the reference's x loop, four columns at a time: (x + range) % 4 == 0 exactly at x = xb
        {
            Mv mv = shr2(Mv(int16_t(origin.x + 4 * xb), int16_t(origin.y + 4 * y)));
            limit(mv);
            mv1 = Mv(int16_t(mv.x + 1), mv.y);
            limit(mv1);
            mv2 = Mv(int16_t(mv.x + 2), mv.y);
            limit(mv2);
            mv3 = Mv(int16_t(mv.x + 3), mv.y);
            limit(mv3);
            const Mv four[4] = {mv, mv1, mv2, mv3};
            int32_t sads[4];
            view.sad4(four, sads);
            ++r.calls;
            HAVOC_UNROLL
            for (int i = 0; i < 4; ++i)
            {
                const int x = xb + i;
                if (x > range) continue;
                if (i)
                {   // the candidate is the limited position of ITS column, the SAD the one taken i samples right of the group's first (the
                    // reference pairs them like this: they differ only where the limit moved the group's first position)
                    mv = shr2(Mv(int16_t(origin.x + 4 * x), int16_t(origin.y + 4 * y)));
                    limit(mv);
                }
                MvCandidate candidate(shl2(mv), pu.mvp, pu.mvpRate);
                candidate.cost += lambda * sads[i];
                best.consider(candidate);
            }
        }
    if (sp.halfPel)
    {
        const int refinement = sp.quarterPel ? 1 : 2;
        for (int step = 2; step; step -= refinement)
        {
            const Mv org = best.mv;
            best.cost = kCostMax;
            {
                Mv ask[9];
                HAVOC_UNROLL
                for (int k = 0; k < 9; ++k) ask[k] = Mv(int16_t(org.x + (k % 3 - 1) * step), int16_t(org.y + (k / 3 - 1) * step));
                hintSatd(view, ask, 9, 0);
            }
            for (int y = -step; y <= step; y += step)
                for (int x = -step; x <= step; x += step)
                {
                    const Mv mv(int16_t(org.x + x), int16_t(org.y + y));
                    MvCandidate candidate(mv, pu.mvp, pu.mvpRate);
                    candidate.cost += lambda * view.satdQpel(mv);   // costDistortionMv(..., k = 0.5)
                    ++r.calls;
                    best.consider(candidate);
                }
        }
    }
    r.mv = best.mv;
    r.mvd = best.mvd;
    r.mvpFlag = best.mvpFlag;
    r.cost = best.cost;
    return r;
}

// ---- intra: the 35-mode SATD stage and the order in which modes go forward to RD refinement (Search.hpp:40-190)
struct IntraContext
{
    int candModeList[3] = {0, 1, 26};       // CandModeList (most probable modes)
    int neighbourModes = 3;                 // candModeList.neighbourModes
    Cost rateBminusC = 0;                   // rateB - rateC of the current CABAC state (rateA - rateC = -rateC)
    Cost rateAminusC = 0;
    int maxRefine = 3;                      // Speed::nCandidatesIntraRefinement(log2 partition size)
};

struct IntraResult
{
    Cost costs[35];                         // after the SATD stage
    int order[35];                          // modes in the order they would be RD-refined
    int count = 0;
};

// satd35[m] = the distortion predictIntraLuma returns for mode m (prediction + Hadamard SATD against the source)
HAVOC_HD inline IntraResult intraModeOrder(const IntraContext &ic, double reciprocalSqrtLambda, const int32_t satd35[35])
{
    IntraResult r;
    Lambda lambda;
    lambda.set(reciprocalSqrtLambda);
    Cost costs[35];
    for (int n = 0; n < 35; ++n) costs[n] = 0;
    costs[ic.candModeList[0]] = ic.rateAminusC;
    costs[ic.candModeList[1]] = ic.rateBminusC;
    costs[ic.candModeList[2]] = ic.rateBminusC;
    for (int n = 0; n < 35; ++n) costs[n] += lambda * satd35[n];
    for (int n = 0; n < 35; ++n) r.costs[n] = costs[n];
    int nMpm = 0;
    for (int j = 0; j < ic.maxRefine + nMpm; ++j)
    {
        int mode = 0;
        Cost costBest = costs[0];
        for (int i = 1; i < 35; ++i)
            if (costs[i] < costBest)
            {
                costBest = costs[i];
                mode = i;
            }
        costs[mode] = kCostMax;
        if (j == ic.maxRefine - 1)
            for (int i = 0; i < ic.neighbourModes; ++i)
                if (costs[ic.candModeList[i]] != kCostMax)
                {
                    costs[ic.candModeList[i]] = 0;
                    ++nMpm;
                }
        r.order[r.count++] = mode;
    }
    return r;
}

} // namespace havoc_search
