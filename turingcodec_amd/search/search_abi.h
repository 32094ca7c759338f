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

/* ---- a whole picture's uni-directional searches, in an order the encoder could issue them (picture_order.hpp) ----
 * One prediction unit; the picture's PUs come CTU by CTU (raster order), inside a CTU in the order the quadtree search meets them
 * (turing/Search.hpp:708-887: a coding unit's part modes, then its four sub-units in z-order).  Each PU is searched in list 0, then list 1
 * (searchMotionUni(L0, 0), (L1, 0), Search.hpp:1883-1884) with predictors DERIVED from the vectors decided before it. */
typedef struct
{
    int32_t x0, y0, w, h;
    int32_t cu_log2_size, cqt_depth, part_2Nx2N;
    int32_t reserved;
} havoc_picture_pu;                /* 32 bytes */

typedef struct
{
    int32_t steps;                 /* wavefront steps (CTU anti-diagonals with the two-CTU lag of WPP) */
    int32_t rounds;                /* launch + replay rounds over all steps */
    int32_t max_rounds_in_step;
    int32_t launches, surfaces_small, surfaces_zero, surfaces_large, satd_jobs;
    int32_t speculative_runs;      /* searches run ahead on a guessed predecessor */
    int32_t reruns;                /* searches run again because the guess was wrong or data was missing */
    int64_t bytes_down;
    double seconds_gpu, seconds_host, seconds_total;
} havoc_picture_stats;

/* ---- the residual quadtree decision of inter coding units (tu_decision.hpp; turing/Reconstruct.cpp:1296-1428) ---- */
typedef struct
{
    int32_t x0, y0, log2_size;     /* luma coding unit = root of its transform tree */
    int32_t ctx_index;             /* which CABAC-state snapshot RDOQ reads (the CTU's) */
} havoc_rqt_cu;                    /* 16 bytes */

typedef struct                     /* what one transform block candidate came to (Reconstruct.cpp:740-860) */
{
    int32_t cbf;                   /* Rdoq::runQuantisation's return value */
    uint32_t ssd;                  /* source vs reconstruction */
    int32_t nonzero, sum_abs;      /* of the quantised levels: what the rate estimate reads */
} havoc_tu_outcome;                /* 16 bytes */

typedef struct
{
    int32_t depth;                 /* chosen candidate->rqtdepth: 0 = one transform block, 1 = four */
    int32_t tried_zero;            /* 0: all four blocks of the split came out uncoded, depth 0 was never evaluated (Reconstruct.cpp:1328, 1419-1423) */
    havoc_tu_outcome zero, one[4];
    int64_t cost_zero, cost_one;   /* Q16: rate + ssd * reciprocal lambda */
} havoc_rqt_result;                /* 104 bytes */

/* the RD refinement of an intra partition (tu_decision.hpp: decideIntraRd; turing/Search.hpp:143-255): which of the candidate modes won */
typedef struct
{
    int32_t mode;                  /* IntraPredModeY of the champion */
    int32_t index;                 /* its place in the refinement order */
    int32_t evaluated;             /* candidates reconstructed */
    int32_t reserved;
    int64_t cost;                  /* Q16: mode rate + residual rate + ssd * reciprocal lambda */
    havoc_tu_outcome outcome;      /* of the champion's transform block */
} havoc_intra_rd_result;           /* 40 bytes */

/* quantiser parameters of one transform size (turing/QpState.h:85-94) */
typedef struct
{
    int32_t quant_scale, quant_shift, inv_scale, inv_shift;
} havoc_rqt_quant;

typedef struct
{
    int32_t launches, candidates;
    double seconds_gpu, seconds_host, seconds_total;
} havoc_rqt_stats;

/* havoc_search_intra_chain: a level's mode-order step raised a flag (havoc_mi355x_intra_order's d_total[1]): returns HAVOC_SEARCH_EORDER - flags, i.e. -101 = an order
 * was cut at HAVOC_MI355X_INTRA_MAX_ORDER, -102 = a record was out of range, -103 = both; no result is written */
#define HAVOC_SEARCH_EORDER (-100)

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

/* the intra partitions of ONE size of a picture, everything but `out` in device memory (havoc_search_intra_device) */
typedef struct
{
    int32_t log2, n;
    const void *d_neighbours;                       /* as for havoc_mi355x_intra_satd35 */
    const void *d_jobs;                             /* havoc_mi355x_intra_search_job[n]: source block, neighbour arrays, filter mask */
    const havoc_search_intra_ctx *d_ictx;           /* n: most probable modes and their rates */
    const int32_t *d_ctx_index;                     /* n: which CABAC snapshot Rdoq reads */
    void *d_rec;                                    /* champions' reconstructions: block i = n x n samples at i * n * n */
    havoc_intra_rd_result *out;                     /* HOST, n */
} havoc_intra_group;               /* 56 bytes */

/* one partition size of an INTRA picture's dependency chain (havoc_search_intra_chain): the size's partitions ordered by level, everything but `first` and `out` in
 * device memory.  Level l holds the partitions [first[l], first[l + 1]): all their neighbours are final when level l - 1 is. */
typedef struct
{
    int32_t log2, n;
    void *d_neighbours;                             /* 2 * (4 * size + 1) samples per partition: written by the gather, read through the jobs' nb_off / nbf_off */
    const void *d_jobs;                             /* havoc_mi355x_intra_search_job[n] */
    havoc_search_intra_ctx *d_ictx;                 /* n: rates and max_refine given; cand_mode_list / neighbour_modes derived on the device */
    const int32_t *d_ctx_index;                     /* n: which CABAC snapshot Rdoq reads */
    const void *d_parts;                            /* havoc_mi355x_intra_chain_part[n]: position, size, index in coding order */
    void *d_blocks;                                 /* champions' reconstructions: block i = size x size samples at i * size * size */
    const int32_t *first;                           /* HOST, levels + 1 */
    havoc_intra_rd_result *out;                     /* HOST, n */
} havoc_intra_chain_size;          /* 72 bytes */

#ifdef __cplusplus
}
#endif
#endif
