// TEST INFRASTRUCTURE ONLY: the trace points of turing/Search.hpp (see trace_common.h for what the trace is and how it is made).
#pragma once
#include "trace_common.h"

namespace havoc_trace {

// the inputs of one (prediction unit, list) motion search, read where searchMotionUni / searchMotionBi read them
template <class H, class Mvdc>
static inline void searchBegin(int kind, H &h, const Mvdc &mvdc)
{
    prediction_unit const *pu = h;
    coding_quadtree const *cqt = h;
    StateEncode *stateEncode = h;
    Speed *speed = h;
    Mvp::Predictors *predictors = h;
    auto *substream = &h[Concrete<StateSubstream>()];
    const int refList = mvdc.refList;
    int32_t a[14];
    a[0] = h[PicOrderCntVal()];
    a[1] = (*h[RefPicList(refList)][0].dp)[PicOrderCntVal()];
    a[2] = refList;
    a[3] = pu->x0;
    a[4] = pu->y0;
    a[5] = pu->nPbW;
    a[6] = pu->nPbH;
    a[7] = cqt->log2CbSize;
    a[8] = cqt->cqtDepth;
    a[9] = h[PartMode()] == PART_2Nx2N;
    a[10] = h[xCtb()];
    a[11] = h[yCtb()];
    a[12] = stateEncode->concurrentFrames;
    a[13] = (stateEncode->met ? 1 : 0) | (speed->useSmallSearchWindow() ? 2 : 0) | (speed->useBiSmallSearchWindow() ? 4 : 0) |
            (speed->doHalfPelRefinement() ? 8 : 0) | (speed->doQuarterPelRefinement() ? 16 : 0) | (stateEncode->useRateControl ? 32 : 0);
    havoc_trace_emit(kind, 14, a);
    EstimateRateBin<mvp_lX_flag> estimateRateMvp(h, 0);
    const auto &mvp = predictors->mvp[0][refList];
    const auto prev = substream->mvPreviousInteger2Nx2N[refList];
    a[0] = mvp[0][0];
    a[1] = mvp[0][1];
    a[2] = mvp[1][0];
    a[3] = mvp[1][1];
    a[4] = prev[0];
    a[5] = prev[1];
    lohi(a + 6, estimateRateMvp.rate(0).value);
    lohi(a + 8, estimateRateMvp.rate(1).value);
    dbl(a + 10, getReciprocalSqrtLambda(h));
    a[12] = h[BitDepthY()];
    a[13] = h[CtbSizeY()];
    havoc_trace_emit(HAVOC_TR_BEGIN2, 14, a);
}

template <class Candidate>
static inline void candidate(int kind, const Candidate &c)
{
    int32_t a[7] = {c.mv[0], c.mv[1], c.mvd[0], c.mvd[1], c.mvpFlag, 0, 0};
    lohi(a + 5, c.cost.value);
    havoc_trace_emit(kind, 7, a);
}

// a partition's reference samples, 14 per record: index k = 0 .. 4n walks p(-1, 2n - 1) .. p(-1, 0), p(-1, -1), p(0, -1) .. p(2n - 1, -1)
template <class Samples>
static inline void neighbours(int kind, Samples &p, int nTbS)
{
    int32_t v[14];
    int n = 0;
    for (int k = 0; k <= 4 * nTbS; ++k)
    {
        const int x = k <= 2 * nTbS ? -1 : k - 2 * nTbS - 1, y = k < 2 * nTbS ? 2 * nTbS - 1 - k : -1;
        v[n++] = p(x, y);
        if (n == 14 || k == 4 * nTbS)
        {
            havoc_trace_emit(kind, n, v);
            n = 0;
        }
    }
}

// what deriveTemporalLumaMotionVectorPredictors (turing/Mvp.h:44-181) can read of the collocated picture for this prediction unit: for the pin of amvp.hpp's
// deriveTemporalCandidate.  Only reads.
template <class H>
static inline void colocated(H &h, const prediction_unit &pu)
{
    if (!h[slice_temporal_mvp_enabled_flag()]) return;
    const StatePicture *colPic = getColPic(h);
    StateCollocatedMotion *motion = colPic ? colPic->motion.get() : 0;
    if (!motion) return;
    const int picW = h[pic_width_in_luma_samples()], picH = h[pic_height_in_luma_samples()], ctbLog2 = h[CtbLog2SizeY()];
    int32_t a[11] = {pu.x0, pu.y0, pu.nPbW, pu.nPbH, int32_t(motion->poc), static_cast<StatePicture *>(h)->allBackwards ? 1 : 0, h[collocated_from_l0_flag()] ? 1 : 0, picW, picH,
                     ctbLog2, h[PicOrderCntVal()]};
    havoc_trace_emit(HAVOC_TR_COL, 11, a);
    const int xs[2] = {pu.x0 + pu.nPbW, pu.x0 + (pu.nPbW >> 1)}, ys[2] = {pu.y0 + pu.nPbH, pu.y0 + (pu.nPbH >> 1)};
    for (int k = 0; k < 2; ++k)
    {
        int32_t b[11] = {k, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const bool looks = k == 1 || ((pu.y0 >> ctbLog2) == (ys[0] >> ctbLog2) && ys[0] < picH && xs[0] < picW);
        if (looks)
        {
            const PuData &c = (*motion)((xs[k] >> 4) << 4, (ys[k] >> 4) << 4);
            for (int l = 0; l < 2; ++l)
                if (c.isAvailable() && c.predFlag(l))
                {
                    b[1 + l] = 1;
                    b[3 + 2 * l] = c.mv(l)[0];
                    b[4 + 2 * l] = c.mv(l)[1];
                    b[7 + l] = motion->getPoc(c, l);
                    b[9 + l] = motion->getReference(c, l) == LONG_TERM ? 1 : 0;
                }
        }
        havoc_trace_emit(HAVOC_TR_COL_PU, 11, b);
    }
}

// what predictMvp (turing/Mvp.h:195-436) read and what it derived, for the pin of turingcodec_amd/search/amvp.hpp: the five spatial neighbours through the encoder's own
// neighbourPuData(), the temporal candidate through its own deriveTemporalLumaMotionVectorPredictors() (called once more: it only reads), the two predictors it stored
template <class H>
static inline void amvp(H &h, int refList, int refIdx)
{
    prediction_unit const &pu = *static_cast<prediction_unit *>(h);
    Mvp::Predictors *predictors = h;
    int32_t a[14] = {0};
    a[0] = h[PicOrderCntVal()];
    a[1] = refList;
    a[2] = refIdx;
    a[3] = pu.x0;
    a[4] = pu.y0;
    a[5] = pu.nPbW;
    a[6] = pu.nPbH;
    a[7] = (*h[RefPicList(refList)][refIdx].dp)[PicOrderCntVal()];
    if (h[slice_temporal_mvp_enabled_flag()])
    {
        PuData col = PuData();
        a[8] = deriveTemporalLumaMotionVectorPredictors(h, col, pu, refList, refIdx) ? 1 : 0;
        a[9] = col.mv(refList)[0];
        a[10] = col.mv(refList)[1];
    }
    const auto &mvp = predictors->mvp[refIdx][refList];
    a[11] = int32_t(uint32_t(uint16_t(mvp[0][0])) | (uint32_t(uint16_t(mvp[0][1])) << 16));
    a[12] = int32_t(uint32_t(uint16_t(mvp[1][0])) | (uint32_t(uint16_t(mvp[1][1])) << 16));
    a[13] = int32_t(uint32_t(h[pic_width_in_luma_samples()]) | (uint32_t(h[pic_height_in_luma_samples()]) << 16));
    havoc_trace_emit(HAVOC_TR_AMVP, 14, a);
    const int xN[5] = {pu.x0 - 1, pu.x0 - 1, pu.x0 + pu.nPbW, pu.x0 + pu.nPbW - 1, pu.x0 - 1};
    const int yN[5] = {pu.y0 + pu.nPbH, pu.y0 + pu.nPbH - 1, pu.y0 - 1, pu.y0 - 1, pu.y0 - 1};
    for (int k = 0; k < 5; ++k)
    {
        const PuData nb = neighbourPuData(h, xN[k], yN[k]);
        // may the unit read that POSITION at all (the three tests of neighbourPuData, turing/StateSpatial.h:208-246, before it looks at what is stored there)?
        const int maskHigh = ~(h[CtbSizeY()] - 1), yCtbCurr = pu.y0 & maskHigh, xCurr = pu.x0 + pu.nPbW - 1, yCurr = pu.y0 + pu.nPbH - 1;
        AvailabilityCtu *availabilityCtu = h;
        const bool position = !((yN[k] & maskHigh) > yCtbCurr) && availabilityCtu->available(xCurr, yCurr, xN[k], yN[k], h[CtbLog2SizeY()]) &&
                              compareZ(xCurr, yCurr - yCtbCurr, xN[k], yN[k] - yCtbCurr);
        int32_t b[11] = {k, nb.isAvailable() ? 1 : 0, nb.predFlag(0) ? 1 : 0, nb.predFlag(1) ? 1 : 0, 0, 0, 0, 0, 0, 0, position ? 1 : 0};
        for (int l = 0; l < 2; ++l)
            if (nb.isAvailable() && nb.predFlag(l))
            {
                b[4 + l] = (*h[RefPicList(l)][nb.refIdx(l)].dp)[PicOrderCntVal()];
                b[6 + 2 * l] = nb.mv(l)[0];
                b[7 + 2 * l] = nb.mv(l)[1];
            }
        havoc_trace_emit(HAVOC_TR_AMVP_NB, 11, b);
    }
    colocated(h, pu);
}

// what populateMergeCandidates (turing/Mvp.h:486-697) read and what it left, for the pin of tests/merge.hpp: the five spatial neighbours through the
// encoder's own PuMergeNeighbour<>::get, the temporal candidate through its own deriveTemporalLumaMotionVectorPredictors (called once more: it only reads), the list
template <class H>
static inline void mergePack(int kind, int k, const PuData &d, bool withIndex)
{
    int32_t b[10] = {k, d.isAvailable() ? 1 : 0, d.predFlag(0) ? 1 : 0, d.predFlag(1) ? 1 : 0, d.predFlag(0) ? d.refIdx(0) : 0, d.predFlag(1) ? d.refIdx(1) : 0,
                     d.predFlag(0) ? d.mv(0)[0] : 0, d.predFlag(0) ? d.mv(0)[1] : 0, d.predFlag(1) ? d.mv(1)[0] : 0, d.predFlag(1) ? d.mv(1)[1] : 0};
    (void)withIndex;
    havoc_trace_emit(kind, 10, b);
}

template <class H>
static inline void merge(H &h, const prediction_unit &puOrig)
{
    StateSubstream *stateSubstream = h;
    coding_quadtree const *cqt = h;
    Mvp::Predictors *predictors = h;
    prediction_unit pu = puOrig;
    int partIdx = stateSubstream->partIdx;
    const int nCbS = 1 << cqt->log2CbSize;
    if (h[Log2ParMrgLevel()] > 2 && nCbS == 8)
    {
        pu.x0 = cqt->x0;
        pu.y0 = cqt->y0;
        pu.nPbW = pu.nPbH = nCbS;
        partIdx = 0;
    }
    const bool isB = h[slice_type()] == B;
    const int n0 = h[num_ref_idx_l0_active_minus1()] + 1, n1 = isB ? h[num_ref_idx_l1_active_minus1()] + 1 : 0, maxCand = h[MaxNumMergeCand()];
    PuData col;
    col.reset();
    bool colAvailable = false;
    if (h[slice_temporal_mvp_enabled_flag()])
    {
        colAvailable = deriveTemporalLumaMotionVectorPredictors(h, col, pu, L0, 0);
        if (isB) colAvailable |= deriveTemporalLumaMotionVectorPredictors(h, col, pu, L1, 0);
    }
    int32_t a[14] = {h[PicOrderCntVal()], pu.x0, pu.y0, pu.nPbW, pu.nPbH, partIdx, isB ? 1 : 0, n0, n1, maxCand, h[slice_temporal_mvp_enabled_flag()] ? 1 : 0,
                     colAvailable ? 1 : 0, h[Log2ParMrgLevel()], 0};
    havoc_trace_emit(HAVOC_TR_MERGE, 14, a);
    PuData nb;
    nb.reset();
    PuMergeNeighbour<-1, 0>::get(h, nb, pu.x0 - 1, pu.y0 + pu.nPbH - 1, pu.x0, pu.y0);
    mergePack<H>(HAVOC_TR_MERGE_NB, 0, nb, true);
    nb.reset();
    PuMergeNeighbour<0, -1>::get(h, nb, pu.x0 + pu.nPbW - 1, pu.y0 - 1, pu.x0, pu.y0);
    mergePack<H>(HAVOC_TR_MERGE_NB, 1, nb, true);
    nb.reset();
    PuMergeNeighbour<0, -1>::get(h, nb, pu.x0 + pu.nPbW, pu.y0 - 1, pu.x0, pu.y0);
    mergePack<H>(HAVOC_TR_MERGE_NB, 2, nb, true);
    nb.reset();
    PuMergeNeighbour<-1, 0>::get(h, nb, pu.x0 - 1, pu.y0 + pu.nPbH, pu.x0, pu.y0);
    mergePack<H>(HAVOC_TR_MERGE_NB, 3, nb, true);
    nb.reset();
    PuMergeNeighbour<-1, -1>::get(h, nb, pu.x0 - 1, pu.y0 - 1, pu.x0, pu.y0);
    mergePack<H>(HAVOC_TR_MERGE_NB, 4, nb, true);
    mergePack<H>(HAVOC_TR_MERGE_COL, 0, col, false);
    int32_t p[8] = {0};
    for (int i = 0; i < 4; ++i)
    {
        if (i < n0) p[i] = (*h[RefPicList(L0)][i].dp)[PicOrderCntVal()];
        if (i < n1) p[4 + i] = (*h[RefPicList(L1)][i].dp)[PicOrderCntVal()];
    }
    havoc_trace_emit(HAVOC_TR_MERGE_POC, 8, p);
    for (int i = 0; i < maxCand && i < 5; ++i) mergePack<H>(HAVOC_TR_MERGE_OUT, i, predictors->merge[i], true);
    colocated(h, pu);
}

} // namespace havoc_trace

// ---- the macros the inserted lines call (each a statement) ----
#define HAVOC_TRACE_MERGE() havoc_trace::merge(h, pu)
#define HAVOC_TRACE_AMVP() havoc_trace::amvp(h, refList, refIdx)
#define HAVOC_TRACE_UNI_BEGIN() havoc_trace::searchBegin(HAVOC_TR_UNI_BEGIN, h, mvdc)
#define HAVOC_TRACE_UNI_INTEGER() havoc_trace::candidate(HAVOC_TR_UNI_INTEGER, best)
#define HAVOC_TRACE_UNI_SUBPEL()                                       \
    do {                                                               \
        int32_t a_[4] = {mv[0], mv[1], mvd[0], mvd[1]};                \
        havoc_trace_emit(HAVOC_TR_UNI_SUBPEL, 4, a_);                  \
    } while (0)
#define HAVOC_TRACE_UNI_END()                                          \
    do {                                                               \
        int32_t a_[3] = {mvd[0], mvd[1], best.mvpFlag};                \
        havoc_trace_emit(HAVOC_TR_UNI_END, 3, a_);                     \
    } while (0)
// inside StateMeFullPel::considerPattern, after functionSad4: mv[] are the limited full-sample positions
#define HAVOC_TRACE_SAD4(m0, m1, m2, m3)                                                                                     \
    do {                                                                                                                     \
        int32_t a_[12] = {(m0)[0], (m0)[1], (m1)[0], (m1)[1], (m2)[0], (m2)[1], (m3)[0], (m3)[1], sads[0], sads[1], sads[2], sads[3]}; \
        havoc_trace_emit(HAVOC_TR_SAD4, 12, a_);                                                                             \
    } while (0)
#define HAVOC_TRACE_SAD(x, y)                                          \
    do {                                                               \
        int32_t a_[3] = {int32_t(x), int32_t(y), int32_t(sad)};        \
        havoc_trace_emit(HAVOC_TR_SAD, 3, a_);                         \
    } while (0)
// inside costDistortionMv after measureSatd: mv has been shifted to full samples, mvFrac holds the fraction
#define HAVOC_TRACE_SATD()                                                                           \
    do {                                                                                             \
        int32_t a_[3] = {mv[0] * 4 + mvFrac[0], mv[1] * 4 + mvFrac[1], int32_t(distortion)};         \
        havoc_trace_emit(HAVOC_TR_SATD, 3, a_);                                                      \
    } while (0)
#define HAVOC_TRACE_BI_BEGIN()                                                                       \
    do {                                                                                             \
        havoc_trace::searchBegin(HAVOC_TR_BI_BEGIN, h, mvdc);                                        \
        int32_t a_[4] = {puData.mv(0)[0], puData.mv(0)[1], puData.mv(1)[0], puData.mv(1)[1]};        \
        havoc_trace_emit(HAVOC_TR_BI_MV, 4, a_);                                                     \
    } while (0)
#define HAVOC_TRACE_BI_END() havoc_trace::candidate(HAVOC_TR_BI_END, best)
#define HAVOC_TRACE_INTRA_BEGIN()                                                                                            \
    do {                                                                                                                     \
        if (log2PartitionSize != 6)                                                                                          \
        {                                                                                                                    \
            havoc_trace::neighbours(HAVOC_TR_INTRA_NB, stateEncodeSubstream->unfiltered[0], 1 << log2PartitionSize);         \
            if (log2PartitionSize != 2)                                                                                      \
                havoc_trace::neighbours(HAVOC_TR_INTRA_NBF, stateEncodeSubstream->filtered, 1 << log2PartitionSize);         \
        }                                                                                                                    \
        int32_t a_[14] = {h[PicOrderCntVal()], xPositionOf(intraPartition), yPositionOf(intraPartition), log2PartitionSize,  \
                          candModeList[0], candModeList[1], candModeList[2],                                                 \
                          candModeList.neighbourModes | (CandModeList::getCandidate<Left>(h, xPositionOf(intraPartition), yPositionOf(intraPartition)) << 8) | \
                              (CandModeList::getCandidate<Up>(h, xPositionOf(intraPartition), yPositionOf(intraPartition)) << 16)};                            \
        havoc_trace::lohi(a_ + 8, (rateA - rateC).value);                                                                    \
        havoc_trace::lohi(a_ + 10, (rateB - rateC).value);                                                                   \
        havoc_trace::dbl(a_ + 12, getReciprocalSqrtLambda(h));                                                               \
        havoc_trace_emit(HAVOC_TR_INTRA_BEGIN, 14, a_);                                                                      \
    } while (0)
#define HAVOC_TRACE_INTRA_SATD()                                       \
    do {                                                               \
        int32_t a_[4] = {n, distortion, 0, 0};                         \
        havoc_trace::lohi(a_ + 2, costs[n].value);                     \
        havoc_trace_emit(HAVOC_TR_INTRA_SATD, 4, a_);                  \
    } while (0)
#define HAVOC_TRACE_INTRA_MAX()                                        \
    do {                                                               \
        int32_t a_[1] = {int32_t(max)};                                \
        havoc_trace_emit(HAVOC_TR_INTRA_MAX, 1, a_);                   \
    } while (0)
#define HAVOC_TRACE_INTRA_PICK()                                       \
    do {                                                               \
        int32_t a_[2] = {int32_t(j), IntraPredModeY};                  \
        havoc_trace_emit(HAVOC_TR_INTRA_PICK, 2, a_);                  \
    } while (0)
#define HAVOC_TRACE_INTRA_SSD()                                        \
    do {                                                               \
        int32_t a_[2] = {int32_t(ssd), int32_t(lambda.value)};         \
        havoc_trace_emit(HAVOC_TR_INTRA_SSD, 2, a_);                   \
    } while (0)
#define HAVOC_TRACE_INTRA_RATE()                                                                     \
    do {                                                                                             \
        int32_t a_[2];                                                                               \
        havoc_trace::lohi(a_, challenger->rate.value - originalCandidate->rate.value);               \
        havoc_trace_emit(HAVOC_TR_INTRA_RATE, 2, a_);                                                \
    } while (0)
#define HAVOC_TRACE_INTRA_SWAP()                                       \
    do {                                                               \
        int32_t a_[1] = {int32_t(j)};                                  \
        havoc_trace_emit(HAVOC_TR_INTRA_SWAP, 1, a_);                  \
    } while (0)
#define HAVOC_TRACE_INTRA_END()                                                                      \
    do {                                                                                             \
        int32_t a_[1] = {champion->codedCu.IntraPredModeY(intraPartition.blkIdx)};                   \
        havoc_trace_emit(HAVOC_TR_INTRA_END, 1, a_);                                                 \
    } while (0)
