// tu_decision.hpp -- the residual-quadtree decision of an inter coding unit, restated over a per-block interface the way decision.hpp
// restates the motion search.  Reference: turing/Reconstruct.cpp
//   :740-860    ReconstructInterBlock: residual -> forward transform -> Rdoq::runQuantisation -> de-quantise -> inverse transform + add
//               -> SSD(source, reconstruction)                                   [one transform block candidate = View::evaluate]
//   :1296-1428  reconstructInter: with one level of RQT allowed the split tree (rqtdepth 1: four blocks in z-order) is evaluated FIRST;
//               if none of its blocks is coded the unit stays unsplit with no residual and depth 0 is never tried; otherwise depth 0 is
//               evaluated too and the cheaper of  rate + (ssd0 + 4 ssd1 + 4 ssd2) * reciprocalLambda  wins, depth 0 on `<`.
// Luma only here (the chroma blocks of the unit follow the luma tree and add 4 x their SSD: the havoc calls are the same).
//
// The RATE term is the entropy coder's estimate of the coded tree (EstimateRate<residual_coding>, turing/EstimateRate.h), which is CABAC and
// out of this repository's scope: `tuRate` below is a STAND-IN with the same inputs' summary (coded flag, number and magnitude of the
// levels), the same in every arm of the tests.  What is restated -- and checked against the reference's tables + Rdoq.cpp -- is the
// order of evaluation, the uncoded short-cut, the cost arithmetic (Q16) and the strict comparison.
#pragma once

#include "decision.hpp"
#include "search_abi.h"

namespace havoc_search {

// stand-in for EstimateRate<residual_coding>: Q16 bits of one transform block
inline Cost tuRate(const havoc_tu_outcome &t) { return Cost(1 + (t.cbf ? 2 * t.nonzero + t.sum_abs : 0)) << 16; }

// the rate of a candidate tree (Q16 bits) from its blocks' outcomes.  The product's is the stand-in; the trace-pin test (tests/test_trace_pin.py) puts the
// reference encoder's OWN recorded rates here, so that the order, the short-cut, the Q16 arithmetic and the comparison below are exercised on real numbers
struct StandInTreeRate
{
    Cost operator()(int /*depth*/, const havoc_tu_outcome *blocks, int n) const
    {
        Cost r = 0;
        for (int i = 0; i < n; ++i) r += tuRate(blocks[i]);
        return r;
    }
};

// View: havoc_tu_outcome evaluate(int x0, int y0, int log2, int depth) = the chain of Reconstruct.cpp:740-860 for the block at (x0, y0)
template <class View, class Rate = StandInTreeRate>
havoc_rqt_result decideRqt(View &view, const havoc_rqt_cu &cu, Lambda reciprocalLambda, Rate rate = Rate())
{
    havoc_rqt_result r;
    r = havoc_rqt_result();
    const int half = 1 << (cu.log2_size - 1);
    int32_t ssdOne = 0;
    bool coded = false;
    for (int k = 0; k < 4; ++k)      // rqtdepth = 1 first (Reconstruct.cpp:1325-1326), blocks in z-order
    {
        r.one[k] = view.evaluate(cu.x0 + (k & 1) * half, cu.y0 + (k >> 1) * half, cu.log2_size - 1, 1);
        ssdOne += int32_t(r.one[k].ssd);
        coded |= r.one[k].cbf != 0;
    }
    r.cost_one = rate(1, r.one, 4) + reciprocalLambda * ssdOne;
    if (!coded)                      // cbfZero: the unit is left unsplit and without residual
    {
        r.depth = 0;
        r.tried_zero = 0;
        return r;
    }
    r.tried_zero = 1;
    r.zero = view.evaluate(cu.x0, cu.y0, cu.log2_size, 0);
    r.cost_zero = rate(0, &r.zero, 1) + reciprocalLambda * int32_t(r.zero.ssd);
    r.depth = (r.cost_zero < r.cost_one) ? 0 : 1;     // Reconstruct.cpp:1389
    return r;
}

// ---- intra: the RD refinement of a partition's candidate modes (searchIntraPartition's second stage, turing/Search.hpp:143-255) ----
// The first stage (35 predictions + SATD, decision.hpp: intraModeOrder) leaves the modes in the order they are to be refined; each is then
// RECONSTRUCTED -- predict, residual, DST (4x4 luma) / DCT, Rdoq::runQuantisation with the mode's scan (Global.h:1212-1227), de-quantise,
// inverse transform + add, SSD (reconstructIntraLuma -> Reconstruct.cpp:230-353) -- and the champion is the first candidate with the smallest
// rate + ssd * reciprocalLambda (`challenger->cost2() < champion->cost2()`, strict; the reference adds a challenger's mode rate only when it is
// already ahead on distortion, which cannot change who wins).  Rates: the mode's (relative to a non-MPM mode, as the first stage has them)
// and the stand-in `tuRate` for the residual.
// View: havoc_tu_outcome evaluate(int mode, int index).
inline int intraScanIdx(int log2TrafoSize, int mode)      // Global.h:1212-1227, luma
{
    if (log2TrafoSize != 2 && log2TrafoSize != 3) return 0;
    return (mode >= 6 && mode <= 14) ? 2 : ((mode >= 22 && mode <= 30) ? 1 : 0);
}

// the rate of a candidate (mode + residual, Q16 bits): the product's stand-in; kCostMax = "not measured" (the reference measures a challenger's rate only
// when its distortion alone is below the champion's cost, Search.hpp:242-246: one that is not cannot win)
struct StandInIntraRate
{
    const havoc_search_intra_ctx &ic;
    Cost operator()(int mode, int /*j*/, const havoc_tu_outcome &o) const
    {
        const Cost modeRate = mode == ic.cand_mode_list[0] ? ic.rate_a_minus_c : ((mode == ic.cand_mode_list[1] || mode == ic.cand_mode_list[2]) ? ic.rate_b_minus_c : 0);
        return modeRate + tuRate(o);
    }
};

template <class View, class Rate>
havoc_intra_rd_result decideIntraRd(View &view, const havoc_search_intra_result &order, Lambda reciprocalLambda, Rate rate)
{
    havoc_intra_rd_result r;
    r = havoc_intra_rd_result();
    r.mode = -1;
    r.cost = kCostMax;
    for (int j = 0; j < order.count; ++j)
    {
        const int mode = order.order[j];
        const havoc_tu_outcome o = view.evaluate(mode, j);
        const Cost bits = rate(mode, j, o);
        const Cost cost = bits == kCostMax ? kCostMax : bits + reciprocalLambda * int32_t(o.ssd);
        ++r.evaluated;
        if (cost < r.cost)
        {
            r.mode = mode;
            r.index = j;
            r.cost = cost;
            r.outcome = o;
        }
    }
    return r;
}

template <class View>
havoc_intra_rd_result decideIntraRd(View &view, const havoc_search_intra_result &order, const havoc_search_intra_ctx &ic, Lambda reciprocalLambda)
{
    return decideIntraRd(view, order, reciprocalLambda, StandInIntraRate{ic});
}

} // namespace havoc_search
