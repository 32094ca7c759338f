// TEST INFRASTRUCTURE ONLY (see oracle/README.md).  Nothing under turingcodec_amd/ or include/ knows this file exists.
//
// Trace points for the reference encoder's own decision loops (VERDICT r3 "next" #1): oracle/Makefile target `trace` makes a TEMPORARY copy of
// /root/reference/turing/Search.hpp (+ the two translation units that include it), inserts the one-line macro calls below at the places
// oracle/trace_points.txt names (line number + a token that must be on that line: a changed reference fails the build, it never mis-inserts),
// compiles the copy with `-include oracle/trace_hooks.h` and deletes it.  No reference text is stored in this repository; the macros only READ
// encoder state, and tests/test_trace_pin.py first checks that the traced encoder still writes the committed reference stream.
//
// What is recorded, in call order per thread: for every searchMotionUni / searchMotionBi / searchIntraPartition of the encode
//   * its inputs as the reference's code holds them (prediction unit, list, the two predictors, the rates of mvp_lX_flag in the CABAC state of
//     that moment, mvPreviousInteger2Nx2N, lambda, picture order counts of the picture and of the reference picture),
//   * every primitive call it makes (havoc_sad / havoc_sad_multiref positions and values, costDistortionMv positions and SATD values, the 35
//     predictIntraLuma distortions),
//   * what it decided (integer vector, refined vector, mvd, mvp flag, cost; the order in which intra modes go to RD refinement).
// tests/trace_tools.py replays turingcodec_amd/search/decision.hpp on the same inputs and requires the same call sequence and decisions; the
// `-m gpu` half runs the same searches through the device kernel.
#pragma once
#include <stdint.h>
#include <string.h>

extern "C" void havoc_trace_emit(int kind, int n, const int32_t *values);       // oracle/trace_sink.cpp: 64-byte records, one file, mutex

enum
{
    HAVOC_TR_UNI_BEGIN = 1,    // poc, refPoc, refList, x0, y0, w, h, log2CbSize, cqtDepth, part2Nx2N, xCtb, yCtb, concurrentFrames, flags
    HAVOC_TR_BEGIN2 = 2,       // mvp0.x, mvp0.y, mvp1.x, mvp1.y, prev.x, prev.y, rate0 lo, hi, rate1 lo, hi, reciprocalSqrtLambda (double bits) lo, hi, bitDepth, ctbSize
    HAVOC_TR_SAD = 3,          // x, y (full-sample displacement), value
    HAVOC_TR_SAD4 = 4,         // x0, y0, x1, y1, x2, y2, x3, y3, value0..3
    HAVOC_TR_SATD = 5,         // mv.x, mv.y (quarter-sample), value
    HAVOC_TR_UNI_INTEGER = 6,  // best.mv x, y, best.mvd x, y, mvpFlag, cost lo, hi
    HAVOC_TR_UNI_SUBPEL = 7,   // mv x, y, mvd x, y   (after subPelRefinement)
    HAVOC_TR_UNI_END = 8,      // mvd x, y, mvpFlag
    HAVOC_TR_BI_BEGIN = 9,     // as UNI_BEGIN
    HAVOC_TR_BI_MV = 10,       // mv(L0) x, y, mv(L1) x, y  (setPuDataMvpPredFlags: the two vectors the refinement starts from / predicts from)
    HAVOC_TR_BI_END = 11,      // best.mv x, y, best.mvd x, y, mvpFlag, cost lo, hi
    HAVOC_TR_INTRA_BEGIN = 12, // poc, x, y, log2PartitionSize, cand0, cand1, cand2, neighbourModes, (rateA - rateC) lo, hi, (rateB - rateC) lo, hi, lambda bits lo, hi
    HAVOC_TR_INTRA_SATD = 13,  // mode, distortion, cost lo, hi
    HAVOC_TR_INTRA_MAX = 14,   // nCandidatesIntraRefinement
    HAVOC_TR_INTRA_PICK = 15,  // j, IntraPredModeY
    HAVOC_TR_INTRA_SSD = 16,   // ssd of the candidate just reconstructed
    HAVOC_TR_INTRA_END = 17,   // champion's IntraPredModeY
};

namespace havoc_trace {

static inline void lohi(int32_t *out, int64_t v)
{
    out[0] = int32_t(uint32_t(uint64_t(v)));
    out[1] = int32_t(uint32_t(uint64_t(v) >> 32));
}
static inline void dbl(int32_t *out, double d)
{
    int64_t bits;
    memcpy(&bits, &d, 8);
    lohi(out, bits);
}

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

} // namespace havoc_trace

// ---- the macros the inserted lines call (each a statement) ----
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
        int32_t a_[14] = {h[PicOrderCntVal()], xPositionOf(intraPartition), yPositionOf(intraPartition), log2PartitionSize,  \
                          candModeList[0], candModeList[1], candModeList[2], candModeList.neighbourModes};                   \
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
        int32_t a_[1] = {int32_t(ssd)};                                \
        havoc_trace_emit(HAVOC_TR_INTRA_SSD, 1, a_);                   \
    } while (0)
#define HAVOC_TRACE_INTRA_END()                                                                      \
    do {                                                                                             \
        int32_t a_[1] = {champion->codedCu.IntraPredModeY(intraPartition.blkIdx)};                   \
        havoc_trace_emit(HAVOC_TR_INTRA_END, 1, a_);                                                 \
    } while (0)
