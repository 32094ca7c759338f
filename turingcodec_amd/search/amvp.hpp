// amvp.hpp -- the reference's derivation of a prediction unit's two motion vector predictors (HEVC 8.5.3.2.6 / 8.5.3.2.7 as turing/Mvp.h:195-436 applies it),
// restated as DATA-ONLY code: what it reads of the encoder's state is handed in as five neighbour records and the temporal candidate.
//
// Why it is here: until round 5 picture_order.hpp's derivePredictors was a two-candidate stand-in (the cell left of the bottom-left sample, the cell
// above the top-right sample; it gives the encoder's predictors for 75 % of the recorded derivations) and every decision-path number depends on it.  This file is the real
// rule, PINNED: the traced reference encoder records, for every searchUni call, the five neighbours as its neighbourPuData() returned them, the temporal candidate and the
// two predictors it derived (the HAVOC_TRACE_AMVP trace point, inserted after Search.hpp:1779); tests/test_trace_pin.py requires deriveAmvp() to give the same two
// predictors for every record of six encodes (0 differ); the temporal candidate is derived below from the collocated picture's cells and pinned likewise.  picture_order.hpp's walk -- on the host and inside k_search_rows -- derives its predictors with it, from the five
// positions under the encoder's availability rules; the walk's neighbours are vectors of the same list into the same reference picture (no scaling, no temporal candidate).
#pragma once

#include "decision.hpp"

namespace havoc_search {

struct AmvpNeighbour
{
    bool available = false;      // neighbourPuData(...).isAvailable(): inside the picture / slice, already coded, not intra (turing/StateSpatial.h:208-260)
    bool predFlag[2] = {false, false};
    int refPoc[2] = {0, 0};      // picture order count of the reference picture the neighbour's list-l vector points into
    Mv mv[2];
};

// 8.5.3.2.7 (turing/Mvp.h: distScale): the vector of a candidate that points into another reference picture, scaled by the ratio of the picture distances
HAVOC_HD inline int amvpClip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
HAVOC_HD inline Mv amvpDistScale(Mv mv, int tD, int tB)
{
    const int td = amvpClip3(-128, 127, tD), tb = amvpClip3(-128, 127, tB);
    const int tx = (16384 + (td < 0 ? -td : td) / 2) / td;
    const int f = amvpClip3(-4096, 4095, (tb * tx + 32) >> 6);
    auto one = [f](int c) {
        const int p = f * c;
        const int a = ((p < 0 ? -p : p) + 127) >> 8;
        return amvpClip3(-32768, 32767, p < 0 ? -a : a);
    };
    return Mv(int16_t(one(mv.x)), int16_t(one(mv.y)));
}

// ---- the temporal candidate (8.5.3.2.8 / 8.5.3.2.9 as turing/Mvp.h:44-181 applies it), data-only: what it reads of the collocated picture is handed in as two of that
// picture's 16x16 motion cells.  PINNED with the rest: the traced encoder records the two cells with every predictMvp and every populateMergeCandidates, and
// tests/test_trace_pin.py requires the candidate (available or not, vector) the encoder derived -- per list, reference index 0 for merge -- from them.
struct ColocatedCell
{
    bool predFlag[2] = {false, false};      // neither: intra, or not coded
    Mv mv[2];
    int refPoc[2] = {0, 0};                 // picture order count of the picture its list-l vector points into
    bool longTerm[2] = {false, false};
};

// one cell's contribution for list X towards the picture `targetPoc` (short-term): Mvp.h:44-121
HAVOC_HD inline bool colocatedVector(const ColocatedCell &c, int X, int colPoc, int curPoc, int targetPoc, bool allBackwards, bool collocatedFromL0, Mv *out)
{
    *out = Mv(0, 0);
    if (!c.predFlag[0] && !c.predFlag[1]) return false;
    // which of the cell's vectors: its only one; with two, the current list's when every reference picture lies in the past, else the list the collocated picture is NOT taken from
    const int listCol = !c.predFlag[0] ? 1 : (!c.predFlag[1] ? 0 : (allBackwards ? X : (collocatedFromL0 ? 1 : 0)));
    if (c.longTerm[listCol]) return false;      // (the target is a short-term picture: the reference encoder has no others)
    const int colPocDiff = colPoc - c.refPoc[listCol], currPocDiff = curPoc - targetPoc;
    *out = colPocDiff == currPocDiff ? c.mv[listCol] : amvpDistScale(c.mv[listCol], colPocDiff, currPocDiff);
    return true;
}

// the prediction unit's temporal candidate for list X: the cell at its bottom-right corner if that lies in the same CTU row and inside the picture and has motion, else the
// cell at its centre (Mvp.h:141-181)
HAVOC_HD inline bool deriveTemporalCandidate(int xPb, int yPb, int nPbW, int nPbH, int picW, int picH, int ctbLog2, const ColocatedCell &bottomRight, const ColocatedCell &centre,
                                             int X, int colPoc, int curPoc, int targetPoc, bool allBackwards, bool collocatedFromL0, Mv *out)
{
    const int xBr = xPb + nPbW, yBr = yPb + nPbH;
    bool available = false;
    if ((yPb >> ctbLog2) == (yBr >> ctbLog2) && yBr < picH && xBr < picW)
        available = colocatedVector(bottomRight, X, colPoc, curPoc, targetPoc, allBackwards, collocatedFromL0, out);
    if (!available) available = colocatedVector(centre, X, colPoc, curPoc, targetPoc, allBackwards, collocatedFromL0, out);
    return available;
}

// nb[0..4] = A0 (below-left), A1 (left), B0 (above-right), B1 (above), B2 (above-left); X = the list being predicted, curPoc / targetPoc = the picture order counts of
// the current picture and of RefPicList(X)[refIdxLX]; every reference picture is a short-term one (the reference encoder uses no long-term pictures);
// colAvailable / col = the temporal candidate (Mvp.h:141-181), consulted only where the rule consults it.  out[0..1] = mvpListLX[0..1].
HAVOC_HD inline void deriveAmvp(int X, int curPoc, int targetPoc, const AmvpNeighbour nb[5], bool colAvailable, Mv col, Mv out[2])
{
    const int Y = 1 - X;
    bool availableA = false, availableB = false;
    Mv mvA, mvB;
    const bool isScaled = nb[0].available || nb[1].available;
    // (every loop has a constant trip count and is unrolled on the device: the five records then live in registers -- indexed, they went to scratch memory and cost the
    // walk 2.6 ms per 1080p picture)
    // A: the first of A0, A1 with a vector into the target picture (own list first) ...
    HAVOC_UNROLL
    for (int k = 0; k < 2; ++k)
        if (!availableA && nb[k].available)
        {
            if (nb[k].predFlag[X] && nb[k].refPoc[X] == targetPoc) { availableA = true; mvA = nb[k].mv[X]; }
            else if (nb[k].predFlag[Y] && nb[k].refPoc[Y] == targetPoc) { availableA = true; mvA = nb[k].mv[Y]; }
        }
    // ... else the first with any vector, scaled to the target picture's distance (step 7)
    HAVOC_UNROLL
    for (int k = 0; k < 2; ++k)
        if (!availableA && nb[k].available)
        {
            int poc = 0;
            if (nb[k].predFlag[X]) { availableA = true; mvA = nb[k].mv[X]; poc = nb[k].refPoc[X]; }
            else if (nb[k].predFlag[Y]) { availableA = true; mvA = nb[k].mv[Y]; poc = nb[k].refPoc[Y]; }
            if (availableA && poc != targetPoc) mvA = amvpDistScale(mvA, curPoc - poc, curPoc - targetPoc);
        }
    // B: the first of B0, B1, B2 with a vector into the target picture
    HAVOC_UNROLL
    for (int k = 2; k < 5; ++k)
        if (!availableB && nb[k].available)
        {
            if (nb[k].predFlag[X] && nb[k].refPoc[X] == targetPoc) { availableB = true; mvB = nb[k].mv[X]; }
            else if (nb[k].predFlag[Y] && nb[k].refPoc[Y] == targetPoc) { availableB = true; mvB = nb[k].mv[Y]; }
        }
    // steps 4 / 5: with neither A0 nor A1 there, B's unscaled candidate becomes A and B is looked for again, this time allowing a scaled one
    if (!isScaled && availableB)
    {
        availableA = true;
        mvA = mvB;
    }
    if (!isScaled)
    {
        availableB = false;
        HAVOC_UNROLL
        for (int k = 2; k < 5; ++k)
            if (!availableB && nb[k].available)
            {
                int poc = 0;
                if (nb[k].predFlag[X]) { availableB = true; mvB = nb[k].mv[X]; poc = nb[k].refPoc[X]; }
                else if (nb[k].predFlag[Y]) { availableB = true; mvB = nb[k].mv[Y]; poc = nb[k].refPoc[Y]; }
                if (availableB && poc != targetPoc) mvB = amvpDistScale(mvB, curPoc - poc, curPoc - targetPoc);
            }
    }
    // the temporal candidate only when A and B do not already give two different predictors
    const bool useCol = !(availableA && availableB && mvA != mvB) && colAvailable;
    // mvpListLX: A, then B unless it repeats A, then the temporal candidate, zero-filled to two (the list's third entry is never used)
    Mv first(0, 0), second(0, 0);
    int n = 0;
    if (availableA) { first = mvA; n = 1; }
    if (availableB && !(n == 1 && mvB == first))
    {
        if (n == 0) first = mvB; else second = mvB;
        ++n;
    }
    if (useCol && n < 2)
    {
        if (n == 0) first = col; else second = col;
        ++n;
    }
    out[0] = first;
    out[1] = second;
}

} // namespace havoc_search
