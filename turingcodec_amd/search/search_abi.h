/*
 * search_abi.h -- plain-C records shared by the decision-loop clients: libhavoc_search.so (the batch client of
 * include/havoc_mi355x.h, turingcodec_amd/search/batch_search.cpp) and the per-call clients of the classic table API the
 * tests build (tests/search_client.cpp).  Field meaning follows turingcodec_amd/search/decision.hpp, i.e. the state
 * turing/Search.hpp reads (file:line there).
 */
#ifndef HAVOC_SEARCH_ABI_H
#define HAVOC_SEARCH_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct
{
    int32_t pic_width, pic_height, ctb_size, concurrent_frames;
    int32_t met, small_search_window, bi_small_search_window, half_pel, quarter_pel;
    int32_t bit_depth;
    double reciprocal_sqrt_lambda;
} havoc_search_params;

/* one (prediction unit, reference list) search */
typedef struct
{
    int32_t x0, y0, w, h;
    int32_t cu_log2_size, cqt_depth, part_2Nx2N, ref_list;
    int32_t x_ctb, y_ctb;
    int16_t mvp[2][2];             /* predictors, quarter-sample units: [k] = (x, y) */
    int16_t mv_previous_2Nx2N[2];  /* multiple of 4 */
    int16_t mv_other[2];           /* bi search: the other list's final vector (searchMotionBi predicts from it) */
    int64_t mvp_rate[2];           /* Q16 rate of mvp_lX_flag = 0 / 1 */
} havoc_search_pu;                 /* 72 bytes */

typedef struct
{
    int16_t mv[2], mvd[2], mv_integer[2];
    int16_t mvp_flag, wrote_2Nx2N;
    int32_t calls;                 /* primitive calls the reference's loop makes for this search */
    int32_t replays;               /* batch client: times the loop was re-run after a miss (0 for per-call clients) */
    int64_t cost_integer, cost_subpel, cost_mvd_zero[2];
} havoc_search_result;             /* 56 bytes */

/* 35-mode intra stage: per partition */
typedef struct
{
    int32_t cand_mode_list[3], neighbour_modes, max_refine, reserved;
    int64_t rate_a_minus_c, rate_b_minus_c;
} havoc_search_intra_ctx;          /* 40 bytes */

typedef struct
{
    int64_t costs[35];
    int32_t order[35];
    int32_t count;
} havoc_search_intra_result;       /* 424 bytes */

#ifdef __cplusplus
}
#endif
#endif
