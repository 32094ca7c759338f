// picture_search.cpp (libhavoc_search.so) -- a whole picture's uni-directional motion searches as a batch client, issued in an order
// the encoder could issue them: CTUs in WPP wavefront order, every search's predictors taken from the vectors decided before it
// (picture_order.hpp; VERDICT r2 "next" #1).
//
// A search's integer stage reads SADs around positions that depend on its predictors, and its predictors are the RESULTS of the searches
// before it in its CTU and in the CTUs to the left and above.  Launching one batch per search would be a launch per decision; so, per
// wavefront step (the CTUs (x, y) with x + 2y == step, which depend only on CTUs of earlier steps):
//
//   round 0   one SAD-surface launch for every (PU, list) of the step's CTUs: +-16 around the vector decided next to the CTU (where its
//             predictors will most likely point), and a small +-4 surface around the zero vector where the first does not cover it (the
//             zero vector is the first start point of fullPelMotionEstimation, Search.hpp:2100-2124);
//   replay    each CTU's chain of searches runs on a host thread through decision.hpp, in list order.  A search that reaches its
//             sub-sample stage without data notes "49 positions around my integer vector" and the chain RUNS AHEAD on a guess (the
//             integer vector standing in for the refined one), so that the searches after it name what THEY need in the same round;
//             a search whose integer stage misses its surfaces notes the position and stops its chain;
//   round k   everything noted is launched (one +-64 surface launch, <= 4 SATD launches), and the chains run again from their first
//             search that is not final.  A search is FINAL when it ran to its end with every search before it final: only then its
//             vector is written to the motion field other CTUs read.  Guessed results are never visible outside their CTU and never kept.
//
// The decisions are those of the sequential per-call walk (walkPictureSequential over the reference's tables) by construction: the same
// loop code, replayed until its inputs are the final ones, on values the GPU kernels computed for exactly the positions asked
// (tests/test_search.py::test_picture_*).
#include "batch_common.hpp"
#include "picture_order.hpp"
#include <cstdio>
#include <cstdlib>

using namespace havoc_search;

namespace {

#define RC(call) HAVOC_SEARCH_RC(call)

constexpr int kR0Default = 16;     // round-0 surface around the predicted vector
constexpr int kRL = 72;     // half-width of a miss surface: the star search's +-64 around a start that may sit a few samples off the surface's centre
constexpr int kRz = 4;      // round-0 surface around the zero vector (MET probe: diamond +-1, hexagon +-2, Search.hpp:2112-2124)

// the integer stage kept for a search is only good for the predictors it was computed with
struct KeptFor
{
    Mv mvp[2], prev;
    bool same(const PuContext &pu) const { return mvp[0] == pu.mvp[0] && mvp[1] == pu.mvp[1] && prev == pu.mvPrevious2Nx2N; }
};

struct Chain                                    // one CTU's searches
{
    int ctuX, ctuY, first, last;                // PU range [first, last)
    int cursor = 0;                             // searches (2 per PU, list-major inside a PU) that are final
    Mv prevFinal[2];                            // mvPreviousInteger2Nx2N after the last final search
    LocalField local;
    bool finished() const { return cursor == 2 * (last - first); }
};

} // namespace

extern "C" {

// Uni-directional motion search (searchMotionUni, turing/Search.hpp:1317-1355) of every PU of ONE picture in both reference lists, in
// wavefront order with derived predictors (picture_order.hpp).  pus[ctu_first[c] .. ctu_first[c + 1]) = the PUs of CTU c (raster order).
// Planes are device memory: the source picture at d_src (+ src_origin = sample offset of picture sample (0, 0)); BOTH reference pictures in
// one allocation d_ref (ref_origin[list]) with one stride and `ref_pad` samples of border; their 16 fractional-sample planes each in one
// allocation d_phase (phase_origin[list] = offset of plane 0 sample (0, 0); plane k is k * plane_elems further).  mvp_rate[k] = Q16 rate of
// mvp_lX_flag == k.  out[2 * p + list]; field_out (optional): int16 [2 lists][picture 4x4 cells][x, y] final vectors.
int havoc_search_picture_uni(havoc_mi355x_ctx *ctx, int S, const havoc_search_params *params, const void *d_src, int64_t src_origin, intptr_t src_stride,
                             const void *d_ref, const int64_t ref_origin[2], intptr_t ref_stride, int ref_pad, const void *d_phase, intptr_t plane_elems,
                             const int64_t phase_origin[2], const havoc_picture_pu *pus, const int32_t *ctu_first, int ctus_x, int ctus_y,
                             const int64_t mvp_rate[2], havoc_search_result *out, int16_t *field_out, int threads, havoc_picture_stats *stats)
{
    if (!ctx || !params || !pus || !ctu_first || !out || !ref_origin || !phase_origin || !mvp_rate || (S != 1 && S != 2) || ctus_x < 1 || ctus_y < 1)
        return HAVOC_MI355X_EINVAL;
    const double tStart = now();
    const SearchParams sp = paramsOf(*params);
    const int W = sp.picWidth, H = sp.picHeight, ctb = sp.ctbSize;
    if (ctb != 64 || ctus_x != (W + 63) / 64 || ctus_y != (H + 63) / 64) return HAVOC_MI355X_EINVAL;
    const int nCtus = ctus_x * ctus_y, nPus = ctu_first[nCtus], n = 2 * nPus;
    if (threads < 1) threads = 1;
    const Cost mvpRate[2] = {mvp_rate[0], mvp_rate[1]};

    havoc_picture_stats pst;
    std::memset(&pst, 0, sizeof(pst));
    havoc_search_stats stt;
    std::memset(&stt, 0, sizeof(stt));
    Arena arena(ctx);
    Launcher launch{ctx, S, d_src, src_stride, d_ref, ref_stride, ref_pad, d_phase, plane_elems, W, H, &arena, &stt};
    launch.direct = getenv("HAVOC_PICTURE_STAGED") == nullptr;     // diagnostic switch: staged copies instead of mapped host memory
    // diagnostic switches (profiles/): half-width of the sub-sample sets (3 = the 49 positions of one refinement, 7 = +- a sample), alternatives off
    // measured on the 1080p clip (profiles/r03/picture_1080p_k*_a*_t*.json), one picture alone: K = 3 with alternatives 21.6 ms in 243 rounds;
    // K = 7: 203 rounds but 28 ms (4.6 x the tile SATDs per set); K = 3 without alternatives 23.5 ms in 269 rounds; before any of it 24.9 ms in 359
    const int subK = getenv("HAVOC_PICTURE_SUBK") ? std::max(3, std::min(11, atoi(getenv("HAVOC_PICTURE_SUBK")))) : 3;
    const bool alternatives = !(getenv("HAVOC_PICTURE_ALT") && atoi(getenv("HAVOC_PICTURE_ALT")) == 0);
    const bool debug = getenv("HAVOC_PICTURE_DEBUG") != nullptr;
    const int kR0 = getenv("HAVOC_PICTURE_R0") ? std::max(8, std::min(48, atoi(getenv("HAVOC_PICTURE_R0")))) : kR0Default;

    std::vector<SearchState> state(n);
    std::vector<Geom> geom(n);
    std::vector<KeptFor> kept(n);
    for (int c = 0; c < nCtus; ++c)
        for (int p = ctu_first[c]; p < ctu_first[c + 1]; ++p)
        {
            const havoc_picture_pu &q = pus[p];
            if (q.w < 4 || q.h < 4 || q.w > 64 || q.h > 64 || (q.w & 3) || q.x0 < 0 || q.y0 < 0 || q.x0 + q.w > W || q.y0 + q.h > H ||
                q.x0 / 64 != c % ctus_x || q.y0 / 64 != c / ctus_x || (q.x0 + q.w - 1) / 64 != c % ctus_x || (q.y0 + q.h - 1) / 64 != c / ctus_x)
                return HAVOC_MI355X_EINVAL;
            for (int l = 0; l < 2; ++l)
                geom[2 * p + l] = Geom{q.x0, q.y0, q.w, q.h, src_origin + int64_t(q.y0) * src_stride + q.x0, ref_origin[l], phase_origin[l]};
        }

    MotionField field;
    field.init(W, H);
    std::vector<Chain> chains(nCtus);
    for (int c = 0; c < nCtus; ++c)
    {
        chains[c].ctuX = c % ctus_x;
        chains[c].ctuY = c / ctus_x;
        chains[c].first = ctu_first[c];
        chains[c].last = ctu_first[c + 1];
    }
    ReplayThreads replayers(threads - 1);

    std::vector<Want> wantR0, wantZero, wantLarge, wantSub;
    std::mutex wantMu;
    const int nSteps = ctus_x + 2 * (ctus_y - 1);
    for (int step = 0; step < nSteps; ++step)
    {
        std::vector<Chain *> active;
        for (int y = 0; y < ctus_y; ++y)
        {
            const int x = step - 2 * y;
            if (x >= 0 && x < ctus_x && chains[y * ctus_x + x].last > chains[y * ctus_x + x].first) active.push_back(&chains[y * ctus_x + x]);
        }
        ++pst.steps;
        if (active.empty()) continue;

        // the substream's mvPreviousInteger2Nx2N comes from the CTU to the left (final since an earlier step); round 0: surfaces around the
        // vector decided next to the CTU and around zero
        for (Chain *ch : active)
        {
            if (ch->ctuX > 0)
            {
                const Chain &left = chains[ch->ctuY * ctus_x + ch->ctuX - 1];
                ch->prevFinal[0] = left.prevFinal[0];
                ch->prevFinal[1] = left.prevFinal[1];
            }
            const int xc = ch->ctuX * 64, yc = ch->ctuY * 64;
            for (int l = 0; l < 2; ++l)
            {
                Mv guess(0, 0);
                if (!field.get(l, xc - 1, std::min(yc + 32, H - 1), &guess)) (void)field.get(l, std::min(xc + 32, W - 1), yc - 1, &guess);
                const Mv full = shr2(Mv(int16_t(guess.x + 2), int16_t(guess.y + 2)));
                for (int p = ch->first; p < ch->last; ++p)
                {
                    const int i = 2 * p + l;
                    int cx = full.x, cy = full.y;
                    if (!launch.clampCentre(geom[i], kR0, &cx, &cy)) return HAVOC_MI355X_EINVAL;
                    wantR0.push_back({i, cx, cy, 0});
                    if (alternatives)
                    {
                        // ... and already the sub-sample positions around that vector (+- a sample) and around zero: where the search ends on the
                        // vector its neighbourhood moves with -- the common case -- its chain needs no second round
                        wantSub.push_back({i, 4 * cx, 4 * cy, subK});
                        if (std::abs(4 * cx) + kSub > subK || std::abs(4 * cy) + kSub > subK) wantSub.push_back({i, 0, 0, kSub});
                    }
                    if (std::abs(cx) > kR0 - kRz || std::abs(cy) > kR0 - kRz)
                    {
                        int zx = 0, zy = 0;
                        if (launch.clampCentre(geom[i], kRz, &zx, &zy) && zx == 0 && zy == 0) wantZero.push_back({i, 0, 0, 0});
                    }
                }
            }
        }

        int roundsInStep = 0;
        for (;;)
        {
            ++roundsInStep;
            ++pst.rounds;
            if (roundsInStep > 4096) return HAVOC_MI355X_EINVAL;   // cannot happen: every round makes the first non-final search of a chain final
            const double tGpu = now();
            RC(launch.surfaces(wantR0, kR0, geom.data(), state.data(), false));
            pst.surfaces_zero += int32_t(wantZero.size());
            RC(launch.surfaces(wantZero, kRz, geom.data(), state.data(), false));
            RC(launch.surfaces(wantLarge, kRL, geom.data(), state.data(), true));
            RC(launch.subSets(wantSub, geom.data(), state.data()));
            wantR0.clear();
            wantZero.clear();
            wantLarge.clear();
            wantSub.clear();
            RC(havoc_mi355x_sync(ctx));
            pst.seconds_gpu += now() - tGpu;

            const double tHost = now();
            std::atomic<int> next{0};
            std::atomic<int> bad{0}, specRuns{0}, reruns{0};
            auto worker = [&]() {
                std::jmp_buf stop;
                std::vector<Want> myLarge, mySub;
                for (;;)
                {
                    const int k = next.fetch_add(1);
                    if (k >= int(active.size())) break;
                    Chain &ch = *active[k];
                    if (ch.finished()) continue;
                    ch.local.load(field, ch.ctuX * 64, ch.ctuY * 64);
                    Mv prev[2] = {ch.prevFinal[0], ch.prevFinal[1]};
                    auto get = [&](int list, int x, int y, Mv *v) { return ch.local.covers(x, y) ? ch.local.get(list, x, y, v) : field.get(list, x, y, v); };
                    bool guessing = false;
                    const int total = 2 * (ch.last - ch.first);
                    for (volatile int s = ch.cursor; s < total; ++s)
                    {
                        const int p = ch.first + s / 2, list = s & 1, i = 2 * p + list;
                        const havoc_picture_pu &q = pus[p];
                        Mv mvp[2];
                        derivePredictors(q, list, ctb, W, H, get, mvp);
                        const PuContext pu = contextOf(q, ctb, mvp, mvpRate, prev[list]);
                        SearchState &st = state[i];
                        if (st.integer.valid && !kept[i].same(pu)) st.integer.valid = false;
                        kept[i].mvp[0] = pu.mvp[0];
                        kept[i].mvp[1] = pu.mvp[1];
                        kept[i].prev = pu.mvPrevious2Nx2N;
                        if (st.replays) reruns.fetch_add(1, std::memory_order_relaxed);
                        if (guessing) specRuns.fetch_add(1, std::memory_order_relaxed);
                        Mv decided;
                        bool wrote = false;
                        Mv integerMv;
                        if (setjmp(stop) == 0)
                        {
                            BatchView view(st, &stop);
                            MotionSearch<BatchView> search(sp, pu, view);
                            const UniResult r = search.run(&st.integer);
                            decided = r.mv;
                            wrote = r.wrote2Nx2N;
                            integerMv = r.mvInteger;
                            if (!guessing)
                            {
                                havoc_search_result &o = out[i];
                                std::memset(&o, 0, sizeof(o));
                                fillUni(r, o);
                                o.replays = st.replays;
                                st.done = true;
                                field.set(list, q.x0, q.y0, q.w, q.h, r.mv);      // final: visible to the CTUs of later steps
                                ch.cursor = s + 1;
                                if (wrote) ch.prevFinal[list] = r.mvInteger;
                            }
                        }
                        else
                        {
                            ++st.replays;
                            if (debug && !guessing)
                                fprintf(stderr, "stop step=%d round=%d kind=%d hadsub=%d surfaces=%d miss=%d,%d pu=%d,%d %dx%d last=%d,%d R%d\n", step, roundsInStep, st.miss.kind, int(st.haveSub()), int(st.surfaces.size()), st.miss.x, st.miss.y, q.x0, q.y0, q.w, q.h, st.surfaces.back().cx, st.surfaces.back().cy, st.surfaces.back().R);
                            if (st.miss.kind == 2 && st.integer.valid)
                            {
                                // sub-sample data missing: ask for the 49 positions around the integer vector and run ahead on that vector
                                integerMv = st.integer.best.mv;
                                const int cxq = ((st.miss.x + 2) >> 2) * 4, cyq = ((st.miss.y + 2) >> 2) * 4;
                                if (debug && st.haveSub())
                                {
                                    fprintf(stderr, "resub want=%d,%d (miss %d,%d) mvp0=%d,%d have:", cxq, cyq, st.miss.x, st.miss.y, pu.mvp[0].x, pu.mvp[0].y);
                                    for (const auto &u : st.subs) fprintf(stderr, " (%d,%d K%d)", u.cx, u.cy, u.K);
                                    fprintf(stderr, "\n");
                                }
                                // The first time a search asks it also gets the 49 positions around the other vectors it most often ends on when
                                // the searches before it become final: zero, its first predictor, the vector decided next to its CTU.  (A set may
                                // be wider than one refinement needs -- HAVOC_PICTURE_SUBK = 7 also covers the integer vector moving by a sample:
                                // fewer rounds, more tile SATDs per round; measured slower, see above.)
                                const bool first = st.subs.size() <= 2 && !st.askedAlternatives;
                                mySub.push_back({i, cxq, cyq, subK});
                                st.askedAlternatives = true;
                                if (first && alternatives)
                                {
                                    // ... zero, its first predictor, and the vector decided next to its CTU (where its round-0 surface sits)
                                    const Mv p0 = shl2(shr2(Mv(int16_t(pu.mvp[0].x + 2), int16_t(pu.mvp[0].y + 2))));
                                    const int ax[3] = {0, p0.x, 4 * st.surfaces[0].cx}, ay[3] = {0, p0.y, 4 * st.surfaces[0].cy};
                                    for (int a = 0; a < 3; ++a)
                                    {
                                        bool covered = std::abs(ax[a] - cxq) + kSub <= subK && std::abs(ay[a] - cyq) + kSub <= subK;
                                        for (int b = 0; b < a; ++b) covered |= ax[b] == ax[a] && ay[b] == ay[a];
                                        for (const auto &u : st.subs) covered |= std::abs(ax[a] - u.cx) + kSub <= u.K && std::abs(ay[a] - u.cy) + kSub <= u.K;
                                        if (!covered) mySub.push_back({i, ax[a], ay[a], kSub});
                                    }
                                }
                                decided = integerMv;
                                wrote = st.integer.wrote2Nx2N;
                                guessing = true;
                            }
                            else if (st.miss.kind == 1)
                            {
                                // Where to centre the +-64 surface.  What asks far from the round-0 window is the star search at its large distances
                                // (+-64 around its START, which is almost always the first predictor rounded, Search.hpp:2196-2215) and the raster
                                // refinement (absolute positions +-60 around zero, :2253-2267): a surface centred on the missed position serves half
                                // of either.  So: the first predictor, else zero, else the position itself -- the first whose window holds the
                                // position and that this search does not have yet.
                                const Mv p0 = shr2(Mv(int16_t(pu.mvp[0].x + 2), int16_t(pu.mvp[0].y + 2)));
                                const int tryX[3] = {p0.x, 0, st.miss.x}, tryY[3] = {p0.y, 0, st.miss.y};
                                int cx = st.miss.x, cy = st.miss.y;
                                for (int t = 0; t < 3; ++t)
                                {
                                    int tx = tryX[t], ty = tryY[t];
                                    if (!launch.clampCentre(geom[i], kRL, &tx, &ty) || std::abs(tx - st.miss.x) > kRL || std::abs(ty - st.miss.y) > kRL) continue;
                                    bool have = false;
                                    for (const SurfaceRef &f : st.surfaces) have |= f.R == kRL && f.cx == tx && f.cy == ty;
                                    if (have) continue;
                                    cx = tx;
                                    cy = ty;
                                    break;
                                }
                                if (!launch.clampCentre(geom[i], kRL, &cx, &cy) || std::abs(cx - st.miss.x) > kRL || std::abs(cy - st.miss.y) > kRL)
                                {
                                    bad = 1;
                                    break;
                                }
                                myLarge.push_back({i, cx, cy, 0});
                                if (!st.askedAlternatives && alternatives)
                                {
                                    st.askedAlternatives = true;
                                    // ... and, for the sub-sample stage that will follow the integer stage next round, the positions around the two
                                    // vectors it will most likely end on: the first predictor and zero
                                    const Mv g = shl2(p0);
                                    mySub.push_back({i, g.x, g.y, subK});
                                    if (std::abs(g.x) + kSub > subK || std::abs(g.y) + kSub > subK) mySub.push_back({i, 0, 0, kSub});
                                }
                                // the integer stage left its surfaces: guess the first predictor (rounded to full samples) for this search, so that
                                // the searches after it can still say what they need in this round
                                decided = shl2(shr2(Mv(int16_t(pu.mvp[0].x + 2), int16_t(pu.mvp[0].y + 2))));
                                integerMv = decided;
                                wrote = pu.part2Nx2N;
                                guessing = true;
                            }
                            else
                            {
                                bad = 1;    // a sub-sample position outside the phase planes: the caller's planes are too small
                                break;
                            }
                        }
                        ch.local.set(list, q.x0, q.y0, q.w, q.h, decided);
                        if (wrote) prev[list] = integerMv;
                    }
                }
                std::lock_guard<std::mutex> lock(wantMu);
                wantLarge.insert(wantLarge.end(), myLarge.begin(), myLarge.end());
                wantSub.insert(wantSub.end(), mySub.begin(), mySub.end());
            };
            if (active.size() < 2 || threads == 1) worker();
            else replayers.run(worker);
            pst.seconds_host += now() - tHost;
            pst.speculative_runs += specRuns.load();
            pst.reruns += reruns.load();
            if (bad.load()) return HAVOC_MI355X_EINVAL;

            bool all = true;
            for (Chain *ch : active) all &= ch->finished();
            if (all) break;
            // a set asked for twice in one round (a search replayed under two guesses cannot happen: one replay per round) -- but the same
            // search may have asked for a set it already has when its integer vector moved: the newest request wins, older data is dropped
        }
        pst.max_rounds_in_step = std::max(pst.max_rounds_in_step, roundsInStep);
    }

    if (field_out)
        for (int l = 0; l < 2; ++l)
            for (size_t c = 0; c < size_t(field.cw) * field.ch; ++c)
            {
                const Mv v = MotionField::unpack(field.mv[l][c]);
                field_out[(size_t(l) * field.cw * field.ch + c) * 2 + 0] = field.valid[l][c] ? v.x : 0;
                field_out[(size_t(l) * field.cw * field.ch + c) * 2 + 1] = field.valid[l][c] ? v.y : 0;
            }
    pst.launches = stt.launches;
    pst.surfaces_small = stt.surfaces_small - pst.surfaces_zero;
    pst.surfaces_large = stt.surfaces_large;
    pst.satd_jobs = stt.satd_jobs;
    pst.bytes_down = stt.bytes_down;
    pst.seconds_total = now() - tStart;
    if (stats) *stats = pst;
    return 0;
}

// The same picture with the decision loops ON THE DEVICE (havoc_mi355x_search_picture_uni, csrc/kernels_search.hip: decision.hpp compiled for
// gfx950, one workgroup per chain of dependent searches, one launch per wavefront step): the PU list goes down, the results come back, nothing in
// between -- no surfaces, no rounds, no replay threads.  Arguments as havoc_search_picture_uni; the phase planes must reach 84 samples beyond the
// picture (ref_pad >= 96 with the planes of havoc_mi355x_interp_planes(12, 4, ...)); the SOURCE plane is read in whole CTUs, so it must be readable up to the
// next multiple of the CTU size to the right of and below the picture (a border of ctb_size - 8 samples; include/havoc_mi355x.h).  Results identical to havoc_search_picture_uni's except
// `replays`.  d_field_keep (optional, device): where the decided field stays for later launches (int16 [2][cells][2]).  out_bi (optional, host,
// [2 * n]): the bi-directional refinements of searchBi after each PU's two searches (include/havoc_mi355x.h: havoc_mi355x_search_picture_uni).
int havoc_search_picture_uni_device(havoc_mi355x_ctx *ctx, int S, const havoc_search_params *params, const void *d_src, int64_t src_origin, intptr_t src_stride,
                                    const void *d_ref, const int64_t ref_origin[2], intptr_t ref_stride, int ref_pad, const void *d_phase, intptr_t plane_elems,
                                    const int64_t phase_origin[2], const havoc_picture_pu *pus, const int32_t *ctu_first, int ctus_x, int ctus_y,
                                    const int64_t mvp_rate[2], havoc_search_result *out, int16_t *field_out, int16_t *d_field_keep, havoc_search_result *out_bi,
                                    havoc_picture_stats *stats)
{
    if (!ctx || !params || !pus || !ctu_first || !out || !ref_origin || !phase_origin || !mvp_rate || (S != 1 && S != 2) || ctus_x < 1 || ctus_y < 1 || ref_pad < 96)
        return HAVOC_MI355X_EINVAL;
    const double tStart = now();
    const int W = params->pic_width, H = params->pic_height;
    if (params->ctb_size != 64 || ctus_x != (W + 63) / 64 || ctus_y != (H + 63) / 64) return HAVOC_MI355X_EINVAL;
    const int nCtus = ctus_x * ctus_y, nPus = ctu_first[nCtus];
    for (int c = 0; c < nCtus; ++c)
        for (int p = ctu_first[c]; p < ctu_first[c + 1]; ++p)
        {
            const havoc_picture_pu &q = pus[p];
            if (q.w < 4 || q.h < 4 || q.w > 64 || q.h > 64 || (q.w & 3) || (q.h & 3) || q.x0 < 0 || q.y0 < 0 || q.x0 + q.w > W || q.y0 + q.h > H ||
                q.x0 / 64 != c % ctus_x || q.y0 / 64 != c / ctus_x || (q.x0 + q.w - 1) / 64 != c % ctus_x || (q.y0 + q.h - 1) / 64 != c / ctus_x)
                return HAVOC_MI355X_EINVAL;
        }
    havoc_picture_stats pst;
    std::memset(&pst, 0, sizeof(pst));
    Arena arena(ctx);
    const int stepLaunches = getenv("HAVOC_SEARCH_STEP_LAUNCHES") && atoi(getenv("HAVOC_SEARCH_STEP_LAUNCHES")) ? 1 : 0;      // diagnostic switch (profiles/)
    const size_t cells = size_t((W + 3) / 4) * ((H + 3) / 4);
    void *dPus, *hPus, *vPus, *dFirst, *hFirst, *vFirst, *dOut, *hOut, *dField, *hField, *dWork, *hx;
    HAVOC_SEARCH_RC(arena.get(size_t(std::max(1, nPus)) * sizeof(havoc_picture_pu), &dPus, &hPus, &vPus));
    HAVOC_SEARCH_RC(arena.get(size_t(nCtus + 1) * 4, &dFirst, &hFirst, &vFirst));
    HAVOC_SEARCH_RC(arena.get(size_t(std::max(1, 2 * nPus)) * sizeof(havoc_search_result), &dOut, &hOut));
    HAVOC_SEARCH_RC(arena.get(2 * cells * 4, &dField, &hField));
    HAVOC_SEARCH_RC(arena.get(havoc_mi355x_search_workspace(W, H), &dWork, &hx));
    void *dBi = nullptr, *hBi = nullptr;
    if (out_bi) HAVOC_SEARCH_RC(arena.get(size_t(std::max(1, 2 * nPus)) * sizeof(havoc_search_result), &dBi, &hBi));
    std::memcpy(hPus, pus, size_t(nPus) * sizeof(havoc_picture_pu));
    std::memcpy(hFirst, ctu_first, size_t(nCtus + 1) * 4);
    HAVOC_SEARCH_RC(havoc_mi355x_h2d_async(ctx, dPus, hPus, size_t(nPus) * sizeof(havoc_picture_pu)));
    HAVOC_SEARCH_RC(havoc_mi355x_h2d_async(ctx, dFirst, hFirst, size_t(nCtus + 1) * 4));
    int16_t *field = d_field_keep ? d_field_keep : static_cast<int16_t *>(dField);
    havoc_mi355x_search_params dp;
    static_assert(sizeof(dp) == sizeof(*params), "search ABI");
    std::memcpy(&dp, params, sizeof(dp));
    HAVOC_SEARCH_RC(havoc_mi355x_search_picture_uni(ctx, S, &dp, mvp_rate, d_src, src_origin, src_stride, d_ref, ref_origin, ref_stride, d_phase, plane_elems, phase_origin,
                                                    dPus, static_cast<const int32_t *>(dFirst), ctus_x, ctus_y, nPus, dOut, dBi, field, dWork, stepLaunches));
    if (out_bi) HAVOC_SEARCH_RC(havoc_mi355x_d2h_async(ctx, hBi, dBi, size_t(2 * nPus) * sizeof(havoc_search_result)));
    void *dFlag, *hFlag;
    HAVOC_SEARCH_RC(arena.get(4, &dFlag, &hFlag));
    HAVOC_SEARCH_RC(havoc_mi355x_d2h_async(ctx, hFlag, static_cast<char *>(dWork) + havoc_mi355x_search_workspace(W, H) - 4, 4));
    HAVOC_SEARCH_RC(havoc_mi355x_d2h_async(ctx, hOut, dOut, size_t(2 * nPus) * sizeof(havoc_search_result)));
    if (field_out) HAVOC_SEARCH_RC(havoc_mi355x_d2h_async(ctx, hField, field, 2 * cells * 4));
    HAVOC_SEARCH_RC(havoc_mi355x_sync(ctx));
    if (*static_cast<const int32_t *>(hFlag)) return HAVOC_MI355X_EDEVICE;      // a row's wait gave up
    std::memcpy(out, hOut, size_t(2 * nPus) * sizeof(havoc_search_result));
    if (out_bi) std::memcpy(out_bi, hBi, size_t(2 * nPus) * sizeof(havoc_search_result));
    if (field_out) std::memcpy(field_out, hField, 2 * cells * 4);
    pst.steps = ctus_x + 2 * (ctus_y - 1);
    pst.launches = (stepLaunches ? pst.steps : 1) + (out_bi ? 2 : 0);
    pst.bytes_down = int64_t(2 * nPus) * sizeof(havoc_search_result) * (out_bi ? 2 : 1) + (field_out ? int64_t(2 * cells * 4) : 0);
    pst.seconds_gpu = pst.seconds_total = now() - tStart;
    if (stats) *stats = pst;
    return 0;
}

} // extern "C"
