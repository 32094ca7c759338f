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
#pragma once      // the part of the trace points every patched translation unit shares: the sink's entry point, the record kinds
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
    HAVOC_TR_INTRA_BEGIN = 12, // poc, x, y, log2PartitionSize, cand0, cand1, cand2, neighbourModes | candIntraPredModeA << 8 | candIntraPredModeB << 16 (CandModeList::getCandidate), (rateA - rateC) lo, hi, (rateB - rateC) lo, hi, lambda bits lo, hi
    HAVOC_TR_INTRA_SATD = 13,  // mode, distortion, cost lo, hi
    HAVOC_TR_INTRA_MAX = 14,   // nCandidatesIntraRefinement
    HAVOC_TR_INTRA_PICK = 15,  // j, IntraPredModeY
    HAVOC_TR_INTRA_SSD = 16,   // ssd of the candidate just reconstructed, reciprocalLambda (Q16)
    HAVOC_TR_INTRA_END = 17,   // champion's IntraPredModeY
    HAVOC_TR_INTRA_RATE = 18,  // (EstimateRateLuma ran for the candidate just reconstructed) its rate since the partition began, lo, hi
    HAVOC_TR_INTRA_SWAP = 19,  // j: that candidate became the champion
    HAVOC_TR_RQT_ONE = 20,     // x0, y0, log2TrafoSize, ssd[0], ssd[1], ssd[2] of the split tree, its rate lo, hi, reciprocalLambda (Q16), rqt_root_cbf
    HAVOC_TR_RQT_ZERO = 21,    // ssd[0..2] of the unsplit block, its rate lo, hi
    HAVOC_TR_INTRA_NB = 23,    // (round 5) before INTRA_BEGIN, partitions below 64x64: the 4n + 1 UNFILTERED reference samples of the partition as the encoder's
                               // substituteFast left them (Search.hpp:57), 14 per record, from the bottom of the left column over the corner to the end of the row above
    HAVOC_TR_INTRA_NBF = 24,   // the same positions of the FILTERED copy (Search.hpp:59; partitions above 4x4)
    HAVOC_TR_AMVP = 25,        // (round 5) after predictMvp in searchUni (Search.hpp:1779): poc, refList X, refIdx, xPb, yPb, nPbW, nPbH, POC of RefPicList(X)[refIdx],
                               // temporal candidate available (deriveTemporalLumaMotionVectorPredictors; 0 when slice_temporal_mvp_enabled_flag is off), its x, y,
                               // mvp[0] x, y packed (x & 0xffff | y << 16), mvp[1] packed, picture width | height << 16
    HAVOC_TR_AMVP_NB = 26,     // five of them after an AMVP record, k = 0 .. 4 = A0, A1, B0, B1, B2 as neighbourPuData() returned them: k, available, predFlag L0, predFlag L1,
                               // POC of its L0 reference, POC of its L1 reference, mv L0 x, y, mv L1 x, y, the POSITION may be read (neighbourPuData's three tests before it looks at what is stored there)
    HAVOC_TR_MERGE = 27,       // (round 5) after populateMergeCandidates in searchMergeModes (Search.hpp:1763 -> Mvp.h:486-697): poc, xPb, yPb, nPbW, nPbH (after the
                               // parallel-merge-level adjustment), partIdx, slice is B, active references of L0, of L1, MaxNumMergeCand, temporal candidates enabled, the
                               // temporal candidate available, Log2ParMrgLevel
    HAVOC_TR_MERGE_NB = 28,    // five after a MERGE record, k = 0 .. 4 = A1, B1, B0, A0, B2 as PuMergeNeighbour<>::get returned them: k, available, predFlag L0, L1, refIdx L0, L1,
                               // mv L0 x, y, mv L1 x, y
    HAVOC_TR_MERGE_COL = 29,   // the temporal candidate (deriveTemporalLumaMotionVectorPredictors for L0, then L1 in a B slice; refIdx 0): predFlag L0, L1, mv L0 x, y, mv L1 x, y
    HAVOC_TR_MERGE_POC = 30,   // picture order counts of RefPicList(L0)[0 .. 3], RefPicList(L1)[0 .. 3] (0 beyond the active entries)
    HAVOC_TR_MERGE_OUT = 31,   // MaxNumMergeCand of them: i, predFlag L0, L1, refIdx L0, L1, mv L0 x, y, mv L1 x, y of predictors->merge[i]
    HAVOC_TR_COL = 32,         // (round 5) after an AMVP group and after a MERGE group, when temporal candidates are enabled and the collocated picture has a motion field: xPb, yPb,
                               // nPbW, nPbH (what deriveTemporalLumaMotionVectorPredictors was given), POC of the collocated picture, StatePicture::allBackwards,
                               // collocated_from_l0_flag, picture width, height, CtbLog2SizeY, POC of the current picture
    HAVOC_TR_COL_PU = 33,      // two after a COL record: the collocated picture's 16x16 cell at 0 = the bottom-right position (zeros where the rule does not look there: another CTU
                               // row, outside the picture), 1 = the centre: position, predFlag L0, L1, mv L0 x, y, mv L1 x, y, POC of its L0 / L1 reference, those are long-term
    HAVOC_TR_RQT_END = 22,     // chosen rqtdepth, cbfZero (the split tree had no coded block: depth 0 never evaluated)
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

} // namespace havoc_trace
