// libhavoc_search.so -- the decision loops of decision.hpp as a BATCH CLIENT of libhavoc_mi355x.so.
//
// The reference's motion search asks for one block at a time and decides before it asks again (turing/Search.hpp:1447-1482,
// 2060-2336): through a per-call interface that is a launch per question.  Here a whole picture's searches run together:
//
//   round 0   one SAD-surface launch: for every (PU, list) the SADs of all integer positions within +-16 of the co-located
//             block (havoc_mi355x_sad_surface) -- a super-set of what most searches will ask;
//   replay    the loops of decision.hpp run on the host, every sad / sad4 question answered by a look-up.  A question
//             outside the data at hand (a far predictor, the raster stage, the sub-sample stage) stops that search with
//             a note of what it needs;
//   round k   one launch per kind for everything asked: +-64 surfaces centred on the missed position, and for searches that
//             reached the sub-sample stage the PU SATDs of all 49 quarter-sample positions around their integer vector
//             against the reference's phase planes (havoc_mi355x_satd_multi);  then the stopped searches are replayed
//             from the start (they are deterministic and cheap), until none stops.
//
// A stopped search leaves its loop with longjmp, not with a C++ exception: a throw costs ~20 us and takes a process-wide lock in the
// unwinder, which made 16 replay threads no faster than one (the loops of decision.hpp hold no state and nothing with a destructor,
// asserted below, so there is nothing to unwind).
//
// havoc_search_motion_bi runs searchMotionBi the same way on "ideal second predictors" it builds on the device first.
//
// Results are the reference's by construction: the same loop code as the per-call clients, fed values the GPU kernels
// computed for exactly the positions asked (tests/test_search.py compares with the reference library's tables).  Plain
// C++ on include/havoc_mi355x.h only.
#include "batch_common.hpp"

using namespace havoc_search;

namespace havoc_search {

std::mutex g_poolMu;
std::map<havoc_mi355x_ctx *, Pool> g_pools;

Pool *poolOf(havoc_mi355x_ctx *ctx)
{
    std::lock_guard<std::mutex> lock(g_poolMu);
    return &g_pools[ctx];
}

} // namespace havoc_search

namespace {

#define RC(call) HAVOC_SEARCH_RC(call)

// what differs between the uni-directional search and the bi-directional refinement: where a PU's source block is, where the first
// SAD surface of a search sits, and which loop of decision.hpp is replayed
struct Flavour
{
    const void *d_a;                    // plane holding every PU's source block
    intptr_t a_stride;
    std::vector<int64_t> a_off;         // per PU: sample offset of its block in d_a
    int r0;                             // half-width of the round-0 surfaces
    std::vector<Mv> centre0;            // per PU: where the round-0 surface is centred (integer samples, relative to the PU)
    std::function<void(int, BatchView &, havoc_search_result &)> replay;
};

int runSearches(havoc_mi355x_ctx *ctx, int S, const havoc_search_params *params, const Flavour &fl,
                const void *d_ref, int64_t ref_origin, intptr_t ref_stride, int ref_pad, const void *d_phase, intptr_t plane_elems,
                int64_t phase_origin, const havoc_search_pu *pus, int n, havoc_search_result *out, int threads,
                havoc_search_stats *stats)
{
    const double tStart = now();
    const SearchParams sp = paramsOf(*params);
    const int kR0 = fl.r0;
    havoc_search_stats stt;
    std::memset(&stt, 0, sizeof(stt));
    std::vector<SearchState> state(n);
    std::vector<Geom> geom(n);
    Arena arena(ctx);
    if (threads < 1) threads = 1;
    const int W = sp.picWidth, H = sp.picHeight;
    Launcher launch{ctx, S, fl.d_a, fl.a_stride, d_ref, ref_stride, ref_pad, d_phase, plane_elems, W, H, &arena, &stt};

    std::vector<Want> wantSurf[2];          // [0] small (round 0), [1] large
    std::vector<Want> wantSub;
    for (int i = 0; i < n; ++i)
    {
        const havoc_search_pu &q = pus[i];
        if (q.w < 4 || q.h < 4 || q.w > 64 || q.h > 64 || (q.w & 3) || q.x0 < 0 || q.y0 < 0 || q.x0 + q.w > W || q.y0 + q.h > H) return HAVOC_MI355X_EINVAL;
        geom[i] = Geom{q.x0, q.y0, q.w, q.h, fl.a_off[i], ref_origin, phase_origin};
        int cx = fl.centre0[i].x, cy = fl.centre0[i].y;
        if (!launch.clampCentre(geom[i], kR0, &cx, &cy)) return HAVOC_MI355X_EINVAL;   // picture (with its padding) smaller than a search window
        wantSurf[0].push_back({i, cx, cy, 0});
    }

    std::vector<int> pending(n);
    for (int i = 0; i < n; ++i) pending[i] = i;
    ReplayThreads replayers(n >= 64 ? threads - 1 : 0);

    while (!pending.empty())
    {
        ++stt.rounds;
        if (stt.rounds > 64) return HAVOC_MI355X_EINVAL;   // cannot happen: every round serves what stopped a search
        const double tGpu = now();
        // ---- launches of this round ----
        RC(launch.surfaces(wantSurf[0], kR0, geom.data(), state.data(), false));
        RC(launch.surfaces(wantSurf[1], kR1, geom.data(), state.data(), true));
        wantSurf[0].clear();
        wantSurf[1].clear();
        RC(launch.subSets(wantSub, geom.data(), state.data()));
        wantSub.clear();
        RC(havoc_mi355x_sync(ctx));
        stt.seconds_gpu += now() - tGpu;

        // ---- replay the stopped searches on the host ----
        const double tHost = now();
        std::atomic<int> next{0};
        auto worker = [&]() {
            std::jmp_buf stop;
            for (;;)
            {
                const int k = next.fetch_add(1);
                if (k >= int(pending.size())) return;
                const int i = pending[k];
                SearchState &st = state[i];
                if (setjmp(stop) == 0)
                {
                    BatchView view(st, &stop);
                    havoc_search_result &o = out[i];
                    std::memset(&o, 0, sizeof(o));
                    fl.replay(i, view, o);
                    o.replays = st.replays;
                    st.done = true;
                }
                else      // the view noted what is missing in st.miss
                    ++st.replays;
            }
        };
        if (pending.size() < 64 || threads == 1) worker();
        else replayers.run(worker);
        stt.seconds_host += now() - tHost;

        std::vector<int> still;
        for (int i : pending)
        {
            SearchState &st = state[i];
            if (st.done) continue;
            ++stt.replays;
            if (st.miss.kind == 1)
            {
                int cx = st.miss.x, cy = st.miss.y;
                // LimitFullPelMv keeps every candidate within reach of a +-64 window that stays inside the 96-sample padding
                if (!launch.clampCentre(geom[i], kR1, &cx, &cy) || std::abs(cx - st.miss.x) > kR1 || std::abs(cy - st.miss.y) > kR1) return HAVOC_MI355X_EINVAL;
                wantSurf[1].push_back({i, cx, cy, 0});
            }
            else if (st.miss.kind == 2)
                // the 49 positions are centred on the full-sample vector being refined: the first sub-sample question is that vector
                // itself (uni, Search.hpp:2340-2358) or its (-2, -2) half-sample neighbour (bi, Search.hpp:1627-1650)
                wantSub.push_back({i, ((st.miss.x + 2) >> 2) * 4, ((st.miss.y + 2) >> 2) * 4, kSub});
            else
                return HAVOC_MI355X_EINVAL;   // a sub-sample position outside the phase planes: the caller's planes are too small
            still.push_back(i);
        }
        pending.swap(still);
    }
    stt.seconds_total = now() - tStart;
    if (stats)
    {
        stats->rounds += stt.rounds;
        stats->launches += stt.launches;
        stats->surfaces_small += stt.surfaces_small;
        stats->surfaces_large += stt.surfaces_large;
        stats->satd_jobs += stt.satd_jobs;
        stats->replays += stt.replays;
        stats->bytes_down += stt.bytes_down;
        stats->seconds_gpu += stt.seconds_gpu;
        stats->seconds_host += stt.seconds_host;
        stats->seconds_total += stt.seconds_total;
    }
    return 0;
}

} // namespace

extern "C" {

// Uni-directional motion search (searchMotionUni, turing/Search.hpp:1317-1355) of n (PU, list) pairs of ONE picture against
// ONE reference picture.  Planes are device memory: *_origin = sample offset of picture sample (0, 0) from the base pointer,
// strides in samples; the reference plane has `ref_pad` samples of replicated border; d_phase = its 16 fractional-sample
// planes (havoc_mi355x_interp_planes / havoc_mi355x_picture_phase_planes), phase k sample (x, y) at
// d_phase[k * plane_elems + phase_origin + y * ref_stride + x].  out[i] = what the per-call loop decides for pus[i].
int havoc_search_motion_uni(havoc_mi355x_ctx *ctx, int S, const havoc_search_params *params, const void *d_src, int64_t src_origin, intptr_t src_stride,
                            const void *d_ref, int64_t ref_origin, intptr_t ref_stride, int ref_pad, const void *d_phase, intptr_t plane_elems,
                            int64_t phase_origin, const havoc_search_pu *pus, int n, havoc_search_result *out, int threads,
                            havoc_search_stats *stats)
{
    if (!ctx || !params || !pus || !out || n < 0 || (S != 1 && S != 2)) return HAVOC_MI355X_EINVAL;
    if (stats) std::memset(stats, 0, sizeof(*stats));
    const SearchParams sp = paramsOf(*params);
    Flavour fl;
    fl.d_a = d_src;
    fl.a_stride = src_stride;
    fl.r0 = 16;      // round 0: +-16 around the co-located block.  Measured on the 1080p clip geometry: +-8 / 12 / 16 / 20 / 24 / 32 move
                     // 27.3 / 22.9 / 23.1 / 24.1 / 25.2 / 28.0 MB per 600 searches in the same 9 rounds (a smaller window is cheaper for
                     // every search but sends more of them to a +-64 surface); +-32 or +-48 for the miss surfaces doubles the rounds
    fl.a_off.resize(n);
    fl.centre0.assign(n, Mv(0, 0));
    for (int i = 0; i < n; ++i) fl.a_off[i] = src_origin + int64_t(pus[i].y0) * src_stride + pus[i].x0;
    fl.replay = [&](int i, BatchView &view, havoc_search_result &o) {
        const PuContext pu = puOf(pus[i]);
        MotionSearch<BatchView> search(sp, pu, view);
        fillUni(search.run(&view.st.integer), o);
    };
    return runSearches(ctx, S, params, fl, d_ref, ref_origin, ref_stride, ref_pad, d_phase, plane_elems, phase_origin, pus, n, out, threads, stats);
}

// The same n searches with the loops inside the kernel (havoc_mi355x_search_motion_uni, csrc/kernels_search.hip: a workgroup per search, ONE
// launch, no rounds): the records go down (72 B each), the results come back (56 B each).  Arguments as havoc_search_motion_uni; the planes must
// reach 84 samples beyond the picture (ref_pad >= 96).  Results identical except `replays` (0).
int havoc_search_motion_uni_device(havoc_mi355x_ctx *ctx, int S, const havoc_search_params *params, const void *d_src, int64_t src_origin, intptr_t src_stride,
                                   const void *d_ref, int64_t ref_origin, intptr_t ref_stride, int ref_pad, const void *d_phase, intptr_t plane_elems,
                                   int64_t phase_origin, const havoc_search_pu *pus, int n, havoc_search_result *out, havoc_search_stats *stats)
{
    if (!ctx || !params || !pus || !out || n < 0 || (S != 1 && S != 2) || ref_pad < 96) return HAVOC_MI355X_EINVAL;
    const double tStart = now();
    if (stats) std::memset(stats, 0, sizeof(*stats));
    if (n == 0) return 0;
    const int W = params->pic_width, H = params->pic_height;
    for (int i = 0; i < n; ++i)
    {
        const havoc_search_pu &q = pus[i];
        if (q.w < 4 || q.h < 4 || q.w > 64 || q.h > 64 || (q.w & 3) || (q.h & 3) || q.x0 < 0 || q.y0 < 0 || q.x0 + q.w > W || q.y0 + q.h > H) return HAVOC_MI355X_EINVAL;
        if (q.mvp_rate[0] < 0 || q.mvp_rate[1] < 0) return HAVOC_MI355X_EINVAL;      // the device compares costs as non-negative numbers (decision.hpp: costLess)
    }
    Arena arena(ctx);
    void *dPus, *hPus, *dOut, *hOut;
    RC(arena.get(size_t(n) * sizeof(havoc_search_pu), &dPus, &hPus));
    RC(arena.get(size_t(n) * sizeof(havoc_search_result), &dOut, &hOut));
    std::memcpy(hPus, pus, size_t(n) * sizeof(havoc_search_pu));
    RC(havoc_mi355x_h2d_async(ctx, dPus, hPus, size_t(n) * sizeof(havoc_search_pu)));
    havoc_mi355x_search_params dp;
    static_assert(sizeof(dp) == sizeof(*params), "search ABI");
    std::memcpy(&dp, params, sizeof(dp));
    RC(havoc_mi355x_search_motion_uni(ctx, S, &dp, d_src, src_origin, src_stride, d_ref, ref_origin, ref_stride, d_phase, plane_elems, phase_origin, dPus, n, dOut));
    RC(havoc_mi355x_d2h_async(ctx, hOut, dOut, size_t(n) * sizeof(havoc_search_result)));
    RC(havoc_mi355x_sync(ctx));
    std::memcpy(out, hOut, size_t(n) * sizeof(havoc_search_result));
    if (stats)
    {
        stats->launches = 1;
        stats->rounds = 1;
        stats->bytes_down = int64_t(n) * sizeof(havoc_search_result);
        stats->seconds_total = stats->seconds_gpu = now() - tStart;
    }
    return 0;
}

// Bi-directional refinement (searchMotionBi, turing/Search.hpp:1498-1657) of n (PU, list) pairs: list pus[i].ref_list is refined
// around start[i] (quarter samples) against the prediction the OTHER list's vector pus[i].mv_other gives.  The "ideal second
// predictor" clip(2 * source - other prediction) of every PU is built on the device (HavocPredUni from d_ref_other, then
// havoc::SubtractBi; Search.hpp:1512-1546) and takes the place of the source block in the SAD surfaces (+-8 around the start:
// the 11 x 11 grid and its four-wide calls) and the sub-sample SATDs.  d_ref / d_phase: the picture of the list being refined.
int havoc_search_motion_bi(havoc_mi355x_ctx *ctx, int S, const havoc_search_params *params, const void *d_src, int64_t src_origin, intptr_t src_stride,
                           const void *d_ref, int64_t ref_origin, intptr_t ref_stride, int ref_pad, const void *d_phase, intptr_t plane_elems,
                           int64_t phase_origin, const void *d_ref_other, int64_t ref_other_origin, const havoc_search_pu *pus, const int16_t *start, int n,
                           havoc_search_result *out, int threads, havoc_search_stats *stats)
{
    if (!ctx || !params || !pus || !out || !start || n < 0 || (S != 1 && S != 2)) return HAVOC_MI355X_EINVAL;
    if (stats) std::memset(stats, 0, sizeof(*stats));
    if (n == 0) return 0;
    const SearchParams sp = paramsOf(*params);
    // the ideal predictors: slot i = 64 x 64 samples, stride 64.  Not from the arena: it is reset by the search rounds below.
    void *dIdeal = nullptr, *dOther = nullptr, *dJobs = nullptr;
    const size_t slotBytes = size_t(64) * 64 * S;
    int rc = havoc_mi355x_malloc(ctx, &dIdeal, n * slotBytes + 256);
    if (!rc) rc = havoc_mi355x_malloc(ctx, &dOther, n * slotBytes + 256);
    if (!rc) rc = havoc_mi355x_malloc(ctx, &dJobs, size_t(n) * 64 + 256);
    std::vector<havoc_mi355x_pred_uni_job> pj(n);
    std::vector<havoc_mi355x_subtract_bi_job> sj(n);
    for (int i = 0; i < n && !rc; ++i)
    {
        const havoc_search_pu &q = pus[i];
        const PuContext pu = puOf(q);
        const LimitFullPelMv limit(pu, sp);
        const Mv other(q.mv_other[0], q.mv_other[1]);
        Mv full = shr2(other);
        limit(full);
        std::memset(&pj[i], 0, sizeof(pj[i]));
        pj[i].dst_off = i * 4096;
        pj[i].ref_off = int32_t(ref_other_origin + int64_t(q.y0 + full.y) * ref_stride + q.x0 + full.x);
        pj[i].w = q.w; pj[i].h = q.h;
        pj[i].xFrac = other.x & 3; pj[i].yFrac = other.y & 3;
        std::memset(&sj[i], 0, sizeof(sj[i]));
        sj[i].dst_off = i * 4096;
        sj[i].pred_off = i * 4096;
        sj[i].src_off = int32_t(src_origin + int64_t(q.y0) * src_stride + q.x0);
        sj[i].w = q.w; sj[i].h = q.h;
    }
    if (!rc) rc = havoc_mi355x_h2d(ctx, dJobs, pj.data(), n * sizeof(pj[0]));
    if (!rc) rc = havoc_mi355x_pred_uni(ctx, S, 8, sp.bitDepth, 64, 64, dOther, 64, d_ref_other, ref_stride, static_cast<const havoc_mi355x_pred_uni_job *>(dJobs), n);
    if (!rc) rc = havoc_mi355x_sync(ctx);
    if (!rc) rc = havoc_mi355x_h2d(ctx, dJobs, sj.data(), n * sizeof(sj[0]));
    // SubtractBi is called with 6 + 2 * sizeof(Sample) as its bit depth (turing/Search.hpp:1542-1546), whatever the picture's
    if (!rc) rc = havoc_mi355x_subtract_bi(ctx, S, 6 + 2 * S, dIdeal, 64, dOther, 64, d_src, src_stride, static_cast<const havoc_mi355x_subtract_bi_job *>(dJobs), n);
    if (!rc) rc = havoc_mi355x_sync(ctx);
    if (!rc)
    {
        Flavour fl;
        fl.d_a = dIdeal;
        fl.a_stride = 64;
        fl.r0 = 8;
        fl.a_off.resize(n);
        fl.centre0.resize(n);
        for (int i = 0; i < n; ++i)
        {
            fl.a_off[i] = int64_t(i) * 4096;
            const PuContext pu = puOf(pus[i]);
            const LimitFullPelMv limit(pu, sp);
            Mv c = shr2(Mv(int16_t(start[2 * i] + 1), int16_t(start[2 * i + 1] + 1)));      // where searchMotionBi puts its grid
            limit(c);
            fl.centre0[i] = c;
        }
        fl.replay = [&](int i, BatchView &view, havoc_search_result &o) {
            const PuContext pu = puOf(pus[i]);
            const BiResult r = searchMotionBi(sp, pu, view, Mv(start[2 * i], start[2 * i + 1]));
            o.mv[0] = r.mv.x; o.mv[1] = r.mv.y;
            o.mvd[0] = r.mvd.x; o.mvd[1] = r.mvd.y;
            o.mvp_flag = int16_t(r.mvpFlag);
            o.calls = r.calls;
            o.cost_subpel = r.cost;
        };
        rc = runSearches(ctx, S, params, fl, d_ref, ref_origin, ref_stride, ref_pad, d_phase, plane_elems, phase_origin, pus, n, out, threads, stats);
        if (stats) stats->launches += 2;
    }
    if (dIdeal) (void)havoc_mi355x_free(ctx, dIdeal);
    if (dOther) (void)havoc_mi355x_free(ctx, dOther);
    if (dJobs) (void)havoc_mi355x_free(ctx, dJobs);
    return rc;
}

// The 35-mode luma stage of n intra partitions of one size (searchIntraPartition, turing/Search.hpp:40-190): one
// havoc_mi355x_intra_satd35 launch gives every partition's 35 prediction + SATD costs, then the host applies the rate
// offsets and takes the modes in the order the reference would refine them.  d_jobs / d_neighbours as for
// havoc_mi355x_intra_satd35; ictx[i] = the partition's most probable modes and rates.  satd35 (optional, host, n * 35)
// receives the raw distortions.
int havoc_search_intra_modes(havoc_mi355x_ctx *ctx, int S, int bitDepth, int log2TrafoSize, const void *d_src, intptr_t stride_src,
                             const void *d_neighbours, const havoc_mi355x_intra_search_job *d_jobs, int n, const havoc_search_intra_ctx *ictx,
                             double reciprocal_sqrt_lambda, havoc_search_intra_result *out, int32_t *satd35)
{
    if (!ctx || !ictx || !out || n < 0) return HAVOC_MI355X_EINVAL;
    if (n == 0) return 0;
    Arena arena(ctx);
    void *dCost, *hCost;
    RC(arena.get(size_t(n) * 35 * 4, &dCost, &hCost));
    RC(havoc_mi355x_intra_satd35(ctx, S, bitDepth, log2TrafoSize, d_src, stride_src, d_neighbours, d_jobs, n, static_cast<int32_t *>(dCost)));
    RC(havoc_mi355x_d2h_async(ctx, hCost, dCost, size_t(n) * 35 * 4));
    RC(havoc_mi355x_sync(ctx));
    const int32_t *cost = static_cast<const int32_t *>(hCost);
    if (satd35) std::memcpy(satd35, cost, size_t(n) * 35 * 4);
    for (int i = 0; i < n; ++i)
    {
        IntraContext ic;
        for (int k = 0; k < 3; ++k) ic.candModeList[k] = ictx[i].cand_mode_list[k];
        ic.neighbourModes = ictx[i].neighbour_modes;
        ic.maxRefine = ictx[i].max_refine;
        ic.rateAminusC = ictx[i].rate_a_minus_c;
        ic.rateBminusC = ictx[i].rate_b_minus_c;
        const IntraResult r = intraModeOrder(ic, reciprocal_sqrt_lambda, cost + 35 * i);
        std::memset(&out[i], 0, sizeof(out[i]));
        for (int m = 0; m < 35; ++m) out[i].costs[m] = r.costs[m];
        for (int m = 0; m < r.count; ++m) out[i].order[m] = r.order[m];
        out[i].count = r.count;
    }
    return 0;
}

// The bi-directional refinements with the loop inside the kernel (havoc_mi355x_search_motion_bi, csrc/kernels_search.hip: a workgroup per refinement,
// ONE launch): the list form of the refinement launches of havoc_search_picture_uni_device.  d_phase_other = the OTHER list's 16 phase planes (the
// prediction the ideal block is built against is read from them).  Results identical to havoc_search_motion_bi except `replays` (0).
int havoc_search_motion_bi_device(havoc_mi355x_ctx *ctx, int S, const havoc_search_params *params, const void *d_src, int64_t src_origin, intptr_t src_stride,
                                  const void *d_ref, int64_t ref_origin, intptr_t ref_stride, int ref_pad, const void *d_phase, intptr_t plane_elems,
                                  int64_t phase_origin, const void *d_phase_other, int64_t phase_other_origin, const havoc_search_pu *pus, const int16_t *start, int n,
                                  havoc_search_result *out, havoc_search_stats *stats)
{
    if (!ctx || !params || !pus || !out || !start || n < 0 || (S != 1 && S != 2) || ref_pad < 96) return HAVOC_MI355X_EINVAL;
    const double tStart = now();
    if (stats) std::memset(stats, 0, sizeof(*stats));
    if (n == 0) return 0;
    const int W = params->pic_width, H = params->pic_height;
    for (int i = 0; i < n; ++i)
    {
        const havoc_search_pu &q = pus[i];
        if (q.w < 4 || q.h < 4 || q.w > 64 || q.h > 64 || (q.w & 3) || (q.h & 3) || q.x0 < 0 || q.y0 < 0 || q.x0 + q.w > W || q.y0 + q.h > H) return HAVOC_MI355X_EINVAL;
        if (q.mvp_rate[0] < 0 || q.mvp_rate[1] < 0) return HAVOC_MI355X_EINVAL;
    }
    Arena arena(ctx);
    void *dPus, *hPus, *dStart, *hStart, *dOut, *hOut;
    RC(arena.get(size_t(n) * sizeof(havoc_search_pu), &dPus, &hPus));
    RC(arena.get(size_t(n) * 4, &dStart, &hStart));
    RC(arena.get(size_t(n) * sizeof(havoc_search_result), &dOut, &hOut));
    std::memcpy(hPus, pus, size_t(n) * sizeof(havoc_search_pu));
    std::memcpy(hStart, start, size_t(n) * 4);
    RC(havoc_mi355x_h2d_async(ctx, dPus, hPus, size_t(n) * sizeof(havoc_search_pu)));
    RC(havoc_mi355x_h2d_async(ctx, dStart, hStart, size_t(n) * 4));
    havoc_mi355x_search_params dp;
    std::memcpy(&dp, params, sizeof(dp));
    RC(havoc_mi355x_search_motion_bi(ctx, S, &dp, d_src, src_origin, src_stride, d_ref, ref_origin, ref_stride, d_phase, plane_elems, phase_origin, d_phase_other,
                                     phase_other_origin, dPus, static_cast<const int16_t *>(dStart), n, dOut));
    RC(havoc_mi355x_d2h_async(ctx, hOut, dOut, size_t(n) * sizeof(havoc_search_result)));
    RC(havoc_mi355x_sync(ctx));
    std::memcpy(out, hOut, size_t(n) * sizeof(havoc_search_result));
    if (stats)
    {
        stats->launches = 1;
        stats->rounds = 1;
        stats->bytes_down = int64_t(n) * sizeof(havoc_search_result);
        stats->seconds_total = stats->seconds_gpu = now() - tStart;
    }
    return 0;
}

// frees the work memory libhavoc_search keeps for a context (call before havoc_mi355x_destroy)
void havoc_search_release(havoc_mi355x_ctx *ctx)
{
    std::lock_guard<std::mutex> lock(havoc_search::g_poolMu);
    auto &g_pools = havoc_search::g_pools;
    auto it = g_pools.find(ctx);
    if (it == g_pools.end()) return;
    it->second.release(ctx);
    g_pools.erase(it);
}

const char *havoc_search_version(void) { return "havoc_search 0.1 (batch client of havoc_mi355x)"; }

} // extern "C"
