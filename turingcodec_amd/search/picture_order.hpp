// picture_order.hpp -- the ORDER and the DEPENDENCIES in which a picture's motion searches can be issued, shared by the batch client
// (picture_search.cpp) and by the per-call clients that give the tests their expected values (tests/search_client.cpp).
//
// In the encoder a search's inputs come from earlier decisions (missing #2):
//   * the two motion vector predictors of a PU are taken from the vectors already decided for its left and upper neighbours
//     (turing/Search.hpp:1317-1355 reads predictors->mvp; turing/Mvp.h derives them from the neighbouring PuData);
//   * mvPreviousInteger2Nx2N is what the last 2Nx2N search of the list left behind (Search.hpp:2170-2176 reads, :2332-2335 writes);
//   * CTUs start in wavefront order: CTU (x, y) may start when (x + 1, y - 1) is done (turing/TaskEncodeSubstream.cpp:71-95).
// This file restates that skeleton -- not the encoder's mode decision: every PU of the caller's list is searched, in list order, and its
// vector becomes the neighbourhood of the PUs after it ("last decision covers the area").  Round 5: the predictor derivation IS the reference's
// (amvp.hpp: the five spatial candidates A0, A1, B0, B1, B2 of turing/Mvp.h:195-436, pinned against the encoder's own derivations), read at the positions
// and under the availability rules of the encoder's neighbourPuData (turing/StateSpatial.h:208-246, Global.h:285-370: inside the picture, not in the next CTU
// row, earlier than the PU in z-order).  What the walk still simplifies: a neighbour is a vector of the SAME list into the SAME reference picture (every PU is
// searched in both lists with reference index 0, and the two lists' walks read nothing of each other: Search.hpp:1883-1884), so no candidate is scaled; there
// is no temporal candidate.  The above-right neighbour B0 makes the wavefront's two-CTU lag a requirement (TaskEncodeSubstream.cpp:71-95).
// What matters for the batch client is that the dependency is REAL: a PU's search cannot be replayed before its neighbours' vectors exist.
#pragma once

#include "amvp.hpp"
#include "decision.hpp"
#include "search_abi.h"

#include <cstdint>
#include <vector>

namespace havoc_search {

// decided vectors at 4x4 granularity, per reference list
struct MotionField
{
    int cw = 0, ch = 0;                       // cells
    std::vector<int32_t> mv[2];               // x | y << 16
    std::vector<uint8_t> valid[2];
    void init(int picW, int picH)
    {
        cw = (picW + 3) / 4;
        ch = (picH + 3) / 4;
        for (int l = 0; l < 2; ++l)
        {
            mv[l].assign(size_t(cw) * ch, 0);
            valid[l].assign(size_t(cw) * ch, 0);
        }
    }
    HAVOC_HD static int32_t pack(Mv v) { return int32_t(uint16_t(v.x)) | (int32_t(uint16_t(v.y)) << 16); }
    HAVOC_HD static Mv unpack(int32_t p) { return Mv(int16_t(p & 0xffff), int16_t(uint32_t(p) >> 16)); }
    bool inside(int x, int y) const { return x >= 0 && y >= 0 && (x >> 2) < cw && (y >> 2) < ch; }
    bool get(int list, int x, int y, Mv *v) const
    {
        if (!inside(x, y)) return false;
        const size_t i = size_t(y >> 2) * cw + (x >> 2);
        if (!valid[list][i]) return false;
        *v = unpack(mv[list][i]);
        return true;
    }
    void set(int list, int x0, int y0, int w, int h, Mv v)
    {
        const int32_t p = pack(v);
        for (int y = y0 >> 2; y < (y0 + h + 3) >> 2 && y < ch; ++y)
            for (int x = x0 >> 2; x < (x0 + w + 3) >> 2 && x < cw; ++x)
            {
                mv[list][size_t(y) * cw + x] = p;
                valid[list][size_t(y) * cw + x] = 1;
            }
    }
};

// A CTU's private copy of its own 64 x 64 area of the field: the batch client runs ahead of what is final on a guess, and a guess
// must not be seen by other CTUs (they never read this area during the same wavefront step) nor survive the round
struct LocalField
{
    int x0 = 0, y0 = 0;                       // CTU origin (samples)
    int32_t mv[2][256];
    uint8_t valid[2][256];
    void load(const MotionField &f, int xCtb, int yCtb)
    {
        x0 = xCtb;
        y0 = yCtb;
        for (int l = 0; l < 2; ++l)
            for (int cy = 0; cy < 16; ++cy)
                for (int cx = 0; cx < 16; ++cx)
                {
                    const int gx = (xCtb >> 2) + cx, gy = (yCtb >> 2) + cy;
                    const bool in = gx < f.cw && gy < f.ch;
                    mv[l][cy * 16 + cx] = in ? f.mv[l][size_t(gy) * f.cw + gx] : 0;
                    valid[l][cy * 16 + cx] = in ? f.valid[l][size_t(gy) * f.cw + gx] : 0;
                }
    }
    bool covers(int x, int y) const { return x >= x0 && y >= y0 && x < x0 + 64 && y < y0 + 64; }
    bool get(int list, int x, int y, Mv *v) const
    {
        const int i = ((y - y0) >> 2) * 16 + ((x - x0) >> 2);
        if (!valid[list][i]) return false;
        *v = MotionField::unpack(mv[list][i]);
        return true;
    }
    void set(int list, int px, int py, int w, int h, Mv v)
    {
        const int32_t p = MotionField::pack(v);
        for (int y = (py - y0) >> 2; y < (py - y0 + h + 3) >> 2 && y < 16; ++y)
            for (int x = (px - x0) >> 2; x < (px - x0 + w + 3) >> 2 && x < 16; ++x)
            {
                mv[list][y * 16 + x] = p;
                valid[list][y * 16 + x] = 1;
            }
    }
};

// loop-free comparison of z-order positions (turing/Global.h:285-298): true if (xN, yN) precedes (xC, yC)
HAVOC_HD inline bool precedesInZ(int xC, int yC, int xN, int yN)
{
    const int xNot = ~xN, yNot = ~yN;
    const int yXor = yC ^ yNot, yAnd = yC & yNot;
    const int p = yAnd | (xC & yXor);
    const int q = ~(yAnd | (xNot & yXor));
    return p > q;
}

// (PINNED: tests/test_trace_pin.py holds this against the three tests of the encoder's own neighbourPuData for the five predictor positions of every searchUni of six traced
// encodes: 0 differ)
// may prediction unit q read its neighbour at (xN, yN)?  neighbourPuData (turing/StateSpatial.h:208-246) with AvailabilityCtu::available (Global.h:317-370) for a
// picture of one slice and one tile: not in the next CTU row, inside the picture, in a CTU that exists and precedes this one, earlier than the PU's last sample in z-order
HAVOC_HD inline bool neighbourPositionAvailable(const havoc_picture_pu &q, int ctb, int picW, int picH, int xN, int yN)
{
    const int maskHigh = ~(ctb - 1), log2 = ctb == 64 ? 6 : (ctb == 32 ? 5 : 4);
    const int yCtbCurr = q.y0 & maskHigh;
    if ((yN & maskHigh) > yCtbCurr) return false;
    const int xCurr = q.x0 + q.w - 1, yCurr = q.y0 + q.h - 1;
    if (xN >= picW || yN >= picH) return false;
    const int dx = (xN >> log2) - (xCurr >> log2), dy = (yN >> log2) - (yCurr >> log2);      // (arithmetic shifts: -1 for a position left of / above the picture)
    if (dy == 0)
    {
        if (dx == 0) { if (!precedesInZ(xCurr, yCurr, xN, yN)) return false; }
        else if (dx > 0 || xN < 0) return false;
    }
    else if (dy > 0 || yN < 0 || xN < 0)
        return false;
    return precedesInZ(xCurr, yCurr - yCtbCurr, xN, yN - yCtbCurr);
}

// the two predictors of PU q in `list`, read through `get(list, x, y, &mv)` (false = nothing decided there)
template <class Get>
HAVOC_HD inline void derivePredictors(const havoc_picture_pu &q, int list, int ctb, int picW, int picH, Get get, Mv mvp[2])
{
    const int xN[5] = {q.x0 - 1, q.x0 - 1, q.x0 + q.w, q.x0 + q.w - 1, q.x0 - 1};                 // A0, A1, B0, B1, B2 (Mvp.h:219-222, 286-287)
    const int yN[5] = {q.y0 + q.h, q.y0 + q.h - 1, q.y0 - 1, q.y0 - 1, q.y0 - 1};
    AmvpNeighbour nb[5];
    HAVOC_UNROLL
    for (int k = 0; k < 5; ++k)
    {
        Mv v;
        if (neighbourPositionAvailable(q, ctb, picW, picH, xN[k], yN[k]) && get(list, xN[k], yN[k], &v))
        {
            nb[k].available = true;
            nb[k].predFlag[0] = true;         // the same list into the same reference picture: refPoc == the target's (0 == 0), nothing to scale; with one list in play its
            nb[k].mv[0] = v;                  // name does not enter the rule, so the records are filed under list 0 -- a constant index: indexed by `list` they went to scratch
        }                                     // memory on the device (112 bytes per lane, +2.6 ms per 1080p picture)
    }
    deriveAmvp(0, 0, 0, nb, false, Mv(0, 0), mvp);
}

HAVOC_HD inline PuContext contextOf(const havoc_picture_pu &q, int ctb, const Mv mvp[2], const Cost mvpRate[2], Mv mvPrevious2Nx2N)
{
    PuContext pu;
    pu.x0 = q.x0; pu.y0 = q.y0; pu.w = q.w; pu.h = q.h;
    pu.cuLog2Size = q.cu_log2_size;
    pu.cqtDepth = q.cqt_depth;
    pu.part2Nx2N = q.part_2Nx2N != 0;
    pu.xCtb = q.x0 / ctb * ctb;
    pu.yCtb = q.y0 / ctb * ctb;
    pu.mvp[0] = mvp[0];
    pu.mvp[1] = mvp[1];
    pu.mvpRate[0] = mvpRate[0];
    pu.mvpRate[1] = mvpRate[1];
    pu.mvPrevious2Nx2N = mvPrevious2Nx2N;
    return pu;
}

// The whole picture through a per-call search function, CTU by CTU in raster order (an order the wavefront rule allows: every CTU it
// depends on comes before it).  search(p, list, puContext) -> UniResult.  out[2 * p + list].
// mvPreviousInteger2Nx2N belongs to the substream (turing/StateEncode.h: StateEncodeSubstream), i.e. with WPP to the CTU row: it starts
// at (0, 0) with the row and is handed from CTU to CTU along it.
// With `bi` (searchBi, turing/Search.hpp:1796-1827, the branch without mvd_l1_zero_flag): after a PU's two uni-directional searches -- unless
// nPbW + nPbH == 12 (Search.hpp:1886) -- list 0 is refined against the prediction from list 1's vector starting at its own uni-directional
// vector, then list 1 against the prediction from list 0's REFINED vector.  bi(p, list, puContext, otherMv, startMv) -> BiResult;
// outBi[2 * p + list] (mv, mvd, mvp_flag, calls, cost_subpel = cost; zero where no refinement runs).  The refined vectors do not enter the motion
// field: which of uni / bi / merge a PU ends up with is the mode decision's (not restated), the field keeps the uni-directional vectors.
HAVOC_HD inline bool biRefined(const havoc_picture_pu &q) { return q.w + q.h != 12; }

struct NoBi
{
    BiResult operator()(int, int, const PuContext &, Mv, Mv) const { return BiResult(); }
};

template <class Search, class Bi = NoBi>
void walkPictureSequential(const SearchParams &sp, const havoc_picture_pu *pus, const int32_t *ctuFirst, int ctusX, int ctusY, const Cost mvpRate[2],
                           Search search, havoc_search_result *out, MotionField &field, Bi bi = Bi(), havoc_search_result *outBi = nullptr)
{
    field.init(sp.picWidth, sp.picHeight);
    auto get = [&](int list, int x, int y, Mv *v) { return field.get(list, x, y, v); };
    Mv mvPrev[2];
    for (int c = 0; c < ctusX * ctusY; ++c)
    {
        if (c % ctusX == 0) mvPrev[0] = mvPrev[1] = Mv(0, 0);
        for (int p = ctuFirst[c]; p < ctuFirst[c + 1]; ++p)
        {
            PuContext ctx[2];
            Mv uni[2];
            for (int list = 0; list < 2; ++list)
            {
                Mv mvp[2];
                derivePredictors(pus[p], list, sp.ctbSize, sp.picWidth, sp.picHeight, get, mvp);
                const PuContext pu = contextOf(pus[p], sp.ctbSize, mvp, mvpRate, mvPrev[list]);
                const UniResult r = search(p, list, pu);
                ctx[list] = pu;
                uni[list] = r.mv;
                havoc_search_result &o = out[2 * p + list];
                o.mv[0] = r.mv.x; o.mv[1] = r.mv.y;
                o.mvd[0] = r.mvd.x; o.mvd[1] = r.mvd.y;
                o.mv_integer[0] = r.mvInteger.x; o.mv_integer[1] = r.mvInteger.y;
                o.mvp_flag = int16_t(r.mvpFlag);
                o.wrote_2Nx2N = r.wrote2Nx2N;
                o.calls = r.calls;
                o.replays = 0;
                o.cost_integer = r.costInteger;
                o.cost_subpel = r.costSubPel;
                o.cost_mvd_zero[0] = r.costMvdZero[0];
                o.cost_mvd_zero[1] = r.costMvdZero[1];
                field.set(list, pus[p].x0, pus[p].y0, pus[p].w, pus[p].h, r.mv);
                if (r.wrote2Nx2N) mvPrev[list] = r.mvInteger;
            }
            if (outBi)
            {
                Mv other = uni[1];
                for (int list = 0; list < 2; ++list)
                {
                    havoc_search_result &o = outBi[2 * p + list];
                    o = havoc_search_result();
                    if (!biRefined(pus[p])) continue;
                    const BiResult r = bi(p, list, ctx[list], other, uni[list]);
                    o.mv[0] = r.mv.x; o.mv[1] = r.mv.y;
                    o.mvd[0] = r.mvd.x; o.mvd[1] = r.mvd.y;
                    o.mvp_flag = int16_t(r.mvpFlag);
                    o.calls = r.calls;
                    o.cost_subpel = r.cost;
                    other = r.mv;
                }
            }
        }
    }
}

} // namespace havoc_search
