// TEST INFRASTRUCTURE since round 6 (moved out of turingcodec_amd/search/: nothing in the product derives a merge candidate list -- VERDICT r5 next #4).
// merge.hpp -- the reference's derivation of a prediction unit's merge candidates (HEVC 8.5.3.2.2 - 8.5.3.2.5 as turing/Mvp.h:486-697 applies it), restated as
// DATA-ONLY code: what it reads of the encoder's state is handed in -- the five spatial neighbours, the temporal candidate, the picture order counts of the reference
// lists -- and the candidate list comes out.
//
// PINNED like amvp.hpp: the traced reference encoder records, for every searchMergeModes call, the neighbours as its own PuMergeNeighbour<>::get returned them, the
// temporal candidate, and the list populateMergeCandidates left (the HAVOC_TRACE_MERGE trace point, inserted after Search.hpp:1763); tests/test_trace_pin.py requires
// deriveMergeCandidates() to give the same list -- flags, reference indices, vectors -- for every record of six encodes (0 differ).  The decision step's merge stage
// (decisions.py: merge_candidates, k_merge_jobs) does NOT use it yet: it measures the five spatial positions unpruned, in both lists, as stated there; this is the rule
// it has to grow into, with the walk's motion field holding prediction lists and reference indices first.
#pragma once

#include "decision.hpp"

namespace havoc_search {

struct MergeCandidate
{
    bool predFlag[2] = {false, false};
    int refIdx[2] = {0, 0};
    Mv mv[2];
    HAVOC_HD bool available() const { return predFlag[0] || predFlag[1]; }      // PuData::isAvailable: coded and not intra
    // PuData::operator== (turing/BlockData.h:82-95): per list the same reference index (or both unused) and, where used, the same vector
    HAVOC_HD bool same(const MergeCandidate &o) const
    {
        for (int l = 0; l < 2; ++l)
        {
            if (predFlag[l] != o.predFlag[l]) return false;
            if (predFlag[l] && (refIdx[l] != o.refIdx[l] || mv[l] != o.mv[l])) return false;
        }
        return true;
    }
};

// nb[0..4] = A1 (left), B1 (above), B0 (above-right), A0 (below-left), B2 (above-left) as the neighbour fetch returned them (not available = no prediction flags);
// partIdx / nPbW / nPbH = the prediction unit after the parallel-merge-level adjustment (the second unit of an Nx2N-like split may not take A1, of a 2NxN-like split B1:
// they would rebuild the 2Nx2N unit); col = the temporal candidate (reference index 0 in every list it has), poc0 / poc1 = picture order counts of the lists' entries;
// out[0 .. maxCand - 1].  Returns the number of candidates that came from neighbours, the temporal candidate and their combinations (the rest are zero vectors).
HAVOC_HD inline int deriveMergeList(const MergeCandidate nb[5], int partIdx, int nPbW, int nPbH, bool colAvailable, const MergeCandidate &col, bool sliceB, int numRef0,
                                    int numRef1, const int *poc0, const int *poc1, int maxCand, MergeCandidate *out)
{
    int n = 0;
    const MergeCandidate none;
    // spatial candidates, each compared with the neighbours the standard names -- with what was FETCHED there, taken into the list or not
    const bool takeA1 = !(partIdx && nPbW < nPbH), takeB1 = !(partIdx && nPbH < nPbW);
    const MergeCandidate &a1 = takeA1 ? nb[0] : none, &b1 = takeB1 ? nb[1] : none;
    if (a1.available()) { out[n++] = a1; if (n == maxCand) return n; }
    if (b1.available() && !a1.same(b1)) { out[n++] = b1; if (n == maxCand) return n; }
    if (nb[2].available() && !nb[2].same(b1)) { out[n++] = nb[2]; if (n == maxCand) return n; }
    if (nb[3].available() && !nb[3].same(a1)) { out[n++] = nb[3]; if (n == maxCand) return n; }
    if (n != 4 && nb[4].available() && !nb[4].same(a1) && !nb[4].same(b1)) { out[n++] = nb[4]; if (n == maxCand) return n; }
    if (colAvailable) { out[n++] = col; if (n == maxCand) return n; }
    // combined bi-predictive candidates (B slices): list 0 of one candidate with list 1 of another, unless both name the same picture with the same vector
    if (sliceB)
    {
        const int numOrig = n;
        const int combMax = numOrig * (numOrig - 1);
        static const signed char l0Idx[12] = {0, 1, 0, 2, 1, 2, 0, 3, 1, 3, 2, 3}, l1Idx[12] = {1, 0, 2, 0, 2, 1, 3, 0, 3, 1, 3, 2};
        for (int c = 0; c < combMax && c < 12; ++c)
        {
            const MergeCandidate &p = out[l0Idx[c]], &q = out[l1Idx[c]];
            if (p.predFlag[0] && q.predFlag[1] && (poc0[p.refIdx[0]] != poc1[q.refIdx[1]] || p.mv[0] != q.mv[1]))
            {
                MergeCandidate m = p;
                m.predFlag[1] = true;
                m.refIdx[1] = q.refIdx[1];
                m.mv[1] = q.mv[1];
                out[n++] = m;
                if (n == maxCand) return n;
            }
        }
    }
    const int real = n;
    // zero vectors: one per reference index the lists share, then index 0
    const int numRef = sliceB && numRef1 < numRef0 ? numRef1 : numRef0;
    for (int z = 0; z < numRef && n < maxCand; ++z)
    {
        MergeCandidate m;
        m.predFlag[0] = true;
        m.refIdx[0] = z;
        if (sliceB) { m.predFlag[1] = true; m.refIdx[1] = z; }
        out[n++] = m;
    }
    while (n < maxCand)
    {
        MergeCandidate m;
        m.predFlag[0] = true;
        if (sliceB) m.predFlag[1] = true;
        out[n++] = m;
    }
    return real;
}

// ... and step 10 of 8.5.3.2.2 (Mvp.h:709-716): an 8x4 / 4x8 prediction unit is never bi-predicted -- its candidates keep list 0 only.  nOrigPbW / nOrigPbH: the unit's own
// size (the list above is derived for the unit after the parallel-merge-level adjustment: the same unit unless Log2ParMrgLevel > 2)
HAVOC_HD inline int deriveMergeCandidates(const MergeCandidate nb[5], int partIdx, int nPbW, int nPbH, int nOrigPbW, int nOrigPbH, bool colAvailable, const MergeCandidate &col,
                                          bool sliceB, int numRef0, int numRef1, const int *poc0, const int *poc1, int maxCand, MergeCandidate *out)
{
    const int real = deriveMergeList(nb, partIdx, nPbW, nPbH, colAvailable, col, sliceB, numRef0, numRef1, poc0, poc1, maxCand, out);
    if (nOrigPbW + nOrigPbH == 12)
        for (int k = 0; k < maxCand; ++k)
            if (out[k].predFlag[0] && out[k].predFlag[1])
            {
                out[k].predFlag[1] = false;
                out[k].refIdx[1] = 0;
                out[k].mv[1] = Mv(0, 0);
            }
    return real;
}

} // namespace havoc_search
