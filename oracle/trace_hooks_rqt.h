// TEST INFRASTRUCTURE ONLY: the trace points of turing/Reconstruct.cpp's residual-quadtree decision (reconstructInter, :1296-1428) -- what
// tests/test_trace_pin.py holds turingcodec_amd/search/tu_decision.hpp's decideRqt against, with the reference's OWN rates and distortions as inputs.
#pragma once
#include "trace_common.h"

#define HAVOC_TRACE_RQT_ONE()                                                                                                                    \
    do {                                                                                                                                         \
        int32_t a_[10] = {tt.x0, tt.y0, tt.log2TrafoSize, stateEncodeSubstream->ssd[0], stateEncodeSubstream->ssd[1], stateEncodeSubstream->ssd[2]}; \
        havoc_trace::lohi(a_ + 6, contextsAndCostOne.rate.value - backupContextsAndCostBefore.rate.value);                                      \
        a_[8] = reciprocalLambda.value;                                                                                                          \
        a_[9] = 1;                                                                                                                               \
        havoc_trace_emit(HAVOC_TR_RQT_ONE, 10, a_);                                                                                              \
    } while (0)
#define HAVOC_TRACE_RQT_ZERO()                                                                                                                   \
    do {                                                                                                                                         \
        int32_t a_[5] = {stateEncodeSubstream->ssd[0], stateEncodeSubstream->ssd[1], stateEncodeSubstream->ssd[2]};                              \
        havoc_trace::lohi(a_ + 3, candidate->rate.value - backupContextsAndCostBefore.rate.value);                                              \
        havoc_trace_emit(HAVOC_TR_RQT_ZERO, 5, a_);                                                                                              \
    } while (0)
#define HAVOC_TRACE_RQT_END()                                                                                                                    \
    do {                                                                                                                                         \
        int32_t a_[5] = {candidate->rqtdepth, cbfZero ? 1 : 0, tt.x0, tt.y0, tt.log2TrafoSize};                                                  \
        havoc_trace_emit(HAVOC_TR_RQT_END, 5, a_);                                                                                               \
    } while (0)
