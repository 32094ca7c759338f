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
#include "../../include/havoc_mi355x.h"
#include "decision.hpp"
#include "search_abi.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <csetjmp>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <type_traits>
#include <vector>

using namespace havoc_search;

extern "C" {
typedef struct
{
    int32_t rounds, launches, surfaces_small, surfaces_large, satd_jobs, replays;
    int64_t bytes_down;
    double seconds_gpu, seconds_host, seconds_total;
} havoc_search_stats;
}

namespace {

constexpr int kR1 = 64;             // half-width of the surfaces launched for a miss
constexpr int kSub = 3, kSubSide = 7, kSubCands = 49;

struct Miss
{
    int kind;       // 1: integer position (x, y) needed; 2: sub-sample centre (quarter units) needed
    int x, y;
};

struct SurfaceRef
{
    int cx, cy, R;
    const int32_t *data;    // (2R+1)^2, row = dy
};

struct SearchState
{
    std::vector<SurfaceRef> surfaces;
    bool haveSub = false;
    int subCx = 0, subCy = 0;       // quarter units, relative to the PU position
    const int32_t *sub = nullptr;   // 49 PU SATDs (-1: outside the phase planes)
    bool done = false;
    int replays = 0;
    Miss miss{0, 0, 0};
    MotionSearch<struct BatchView>::IntegerStage integer;   // uni search: kept once the integer stage has run to its end
};

// the View of decision.hpp over precomputed data.  A question it cannot answer ends the replay: the miss is noted in the search's state
// and control returns to the setjmp in the replay worker.
struct BatchView
{
    SearchState &st;
    std::jmp_buf *stop;
    BatchView(SearchState &s, std::jmp_buf *j) : st(s), stop(j) {}
    [[noreturn]] void miss(int kind, int x, int y)
    {
        st.miss = Miss{kind, x, y};
        std::longjmp(*stop, 1);
    }
    bool lookup(int dx, int dy, int32_t *v) const
    {
        for (const SurfaceRef &f : st.surfaces)
            if (std::abs(dx - f.cx) <= f.R && std::abs(dy - f.cy) <= f.R)
            {
                *v = f.data[(dy - f.cy + f.R) * (2 * f.R + 1) + (dx - f.cx + f.R)];
                return true;
            }
        return false;
    }
    int sad(int dx, int dy)
    {
        int32_t v;
        if (!lookup(dx, dy, &v)) miss(1, dx, dy);
        return v;
    }
    void sad4(const Mv d[4], int32_t out[4])
    {
        for (int i = 0; i < 4; ++i)
            if (!lookup(d[i].x, d[i].y, &out[i])) miss(1, d[i].x, d[i].y);
    }
    int satdQpel(Mv mv)
    {
        if (!st.haveSub || std::abs(mv.x - st.subCx) > kSub || std::abs(mv.y - st.subCy) > kSub) miss(2, mv.x, mv.y);
        const int32_t v = st.sub[(mv.y - st.subCy + kSub) * kSubSide + (mv.x - st.subCx + kSub)];
        if (v < 0) miss(3, mv.x, mv.y);   // the position's window leaves the phase planes: cannot be served
        return v;
    }
};
// what longjmp passes over on its way out of a replay
static_assert(std::is_trivially_destructible<MotionSearch<BatchView>>::value && std::is_trivially_destructible<PuContext>::value &&
                  std::is_trivially_destructible<UniResult>::value && std::is_trivially_destructible<BatchView>::value,
              "a stopped replay leaves by longjmp: nothing on its stack may need a destructor");

// Replay threads of one runSearches call: started once, handed a round's work through run() (each round is a few thousand replays of a few
// microseconds: starting 15 threads per round cost as much as the round)
class ReplayThreads
{
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable wake_, done_;
    std::function<void()> work_;
    int generation_ = 0, busy_ = 0;
    bool quit_ = false;
    void loop()
    {
        int seen = 0;
        for (;;)
        {
            std::function<void()> w;
            {
                std::unique_lock<std::mutex> l(m_);
                wake_.wait(l, [&] { return quit_ || generation_ != seen; });
                if (quit_) return;
                seen = generation_;
                w = work_;
            }
            w();
            {
                std::lock_guard<std::mutex> l(m_);
                if (--busy_ == 0) done_.notify_one();
            }
        }
    }
public:
    explicit ReplayThreads(int extra)
    {
        for (int t = 0; t < extra; ++t) threads_.emplace_back([this] { loop(); });
    }
    ~ReplayThreads()
    {
        {
            std::lock_guard<std::mutex> l(m_);
            quit_ = true;
        }
        wake_.notify_all();
        for (auto &t : threads_) t.join();
    }
    // every pool thread and the caller run `w` once; returns when all have
    void run(const std::function<void()> &w)
    {
        {
            std::lock_guard<std::mutex> l(m_);
            work_ = w;
            busy_ = int(threads_.size());
            ++generation_;
        }
        wake_.notify_all();
        w();
        std::unique_lock<std::mutex> l(m_);
        done_.wait(l, [&] { return busy_ == 0; });
    }
};

SearchParams paramsOf(const havoc_search_params &p)
{
    SearchParams sp;
    sp.picWidth = p.pic_width;
    sp.picHeight = p.pic_height;
    sp.ctbSize = p.ctb_size;
    sp.concurrentFrames = p.concurrent_frames;
    sp.met = p.met != 0;
    sp.smallSearchWindow = p.small_search_window != 0;
    sp.biSmallSearchWindow = p.bi_small_search_window != 0;
    sp.halfPel = p.half_pel != 0;
    sp.quarterPel = p.quarter_pel != 0;
    sp.reciprocalSqrtLambda = p.reciprocal_sqrt_lambda;
    sp.bitDepth = p.bit_depth;
    return sp;
}

PuContext puOf(const havoc_search_pu &q)
{
    PuContext pu;
    pu.x0 = q.x0; pu.y0 = q.y0; pu.w = q.w; pu.h = q.h;
    pu.cuLog2Size = q.cu_log2_size;
    pu.cqtDepth = q.cqt_depth;
    pu.part2Nx2N = q.part_2Nx2N != 0;
    pu.xCtb = q.x_ctb; pu.yCtb = q.y_ctb;
    for (int k = 0; k < 2; ++k)
    {
        pu.mvp[k] = Mv(q.mvp[k][0], q.mvp[k][1]);
        pu.mvpRate[k] = q.mvp_rate[k];
    }
    pu.mvPrevious2Nx2N = Mv(q.mv_previous_2Nx2N[0], q.mv_previous_2Nx2N[1]);
    return pu;
}

double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Work memory of a context: device chunks with pinned host mirrors, bump-allocated within one call and kept between calls
// (allocating pinned memory costs milliseconds; a picture's searches need the same amount every time).
struct Pool
{
    struct Chunk { char *dev, *host; size_t cap, used; };
    std::vector<Chunk> chunks;
    void reset() { for (Chunk &c : chunks) c.used = 0; }
    int get(havoc_mi355x_ctx *ctx, size_t bytes, void **d, void **h)
    {
        bytes = (bytes + 255) & ~size_t(255);
        for (Chunk &c : chunks)
            if (c.cap - c.used >= bytes)
            {
                *d = c.dev + c.used;
                *h = c.host + c.used;
                c.used += bytes;
                return 0;
            }
        Chunk c{nullptr, nullptr, std::max(bytes, size_t(64) << 20), 0};
        void *dp = nullptr, *hp = nullptr, *hd = nullptr;
        int rc = havoc_mi355x_malloc(ctx, &dp, c.cap);
        if (rc) return rc;
        if ((rc = havoc_mi355x_host_alloc(ctx, c.cap, &hp, &hd)))
        {
            (void)havoc_mi355x_free(ctx, dp);
            return rc;
        }
        c.dev = static_cast<char *>(dp);
        c.host = static_cast<char *>(hp);
        c.used = bytes;
        chunks.push_back(c);
        *d = c.dev;
        *h = c.host;
        return 0;
    }
    void release(havoc_mi355x_ctx *ctx)
    {
        for (Chunk &c : chunks)
        {
            (void)havoc_mi355x_free(ctx, c.dev);
            (void)havoc_mi355x_host_free(ctx, c.host);
        }
        chunks.clear();
    }
};

std::mutex g_poolMu;
std::map<havoc_mi355x_ctx *, Pool> g_pools;

struct Arena   // one call's view of its context's pool (a context runs one call at a time: its launches share one stream)
{
    havoc_mi355x_ctx *ctx;
    Pool *pool;
    explicit Arena(havoc_mi355x_ctx *c) : ctx(c)
    {
        std::lock_guard<std::mutex> lock(g_poolMu);
        pool = &g_pools[c];
        pool->reset();
    }
    int get(size_t bytes, void **d, void **h) { return pool->get(ctx, bytes, d, h); }
};

#define RC(call) do { const int rc_ = (call); if (rc_) return rc_; } while (0)

} // namespace

extern "C" {

} // extern "C"

namespace {

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
    const void *d_src = fl.d_a;
    const intptr_t src_stride = fl.a_stride;
    const int kR0 = fl.r0;
    havoc_search_stats stt;
    std::memset(&stt, 0, sizeof(stt));
    std::vector<SearchState> state(n);
    Arena arena(ctx);
    if (threads < 1) threads = 1;
    const int W = sp.picWidth, H = sp.picHeight;

    // where a surface of half-width R for PU i may be centred so that its window stays inside the padded plane
    auto clampCentre = [&](const havoc_search_pu &q, int R, int *cx, int *cy) {
        const int loX = -ref_pad + R - q.x0, hiX = W + ref_pad - q.w - R - 4 - q.x0;
        const int loY = -ref_pad + R - q.y0, hiY = H + ref_pad - q.h - R - q.y0;
        if (loX > hiX || loY > hiY) return false;
        *cx = std::min(std::max(*cx, loX), hiX);
        *cy = std::min(std::max(*cy, loY), hiY);
        return true;
    };

    struct Want { int i, cx, cy; };
    std::vector<Want> wantSurf[2];          // [0] small (round 0), [1] large
    std::vector<Want> wantSub;
    for (int i = 0; i < n; ++i)
    {
        const havoc_search_pu &q = pus[i];
        if (q.w < 4 || q.h < 4 || q.w > 64 || q.h > 64 || (q.w & 3) || q.x0 < 0 || q.y0 < 0 || q.x0 + q.w > W || q.y0 + q.h > H) return HAVOC_MI355X_EINVAL;
        int cx = fl.centre0[i].x, cy = fl.centre0[i].y;
        if (!clampCentre(q, kR0, &cx, &cy)) return HAVOC_MI355X_EINVAL;   // picture (with its padding) smaller than a search window
        wantSurf[0].push_back({i, cx, cy});
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
        for (int big = 0; big < 2; ++big)
        {
            std::vector<Want> &w = wantSurf[big];
            if (w.empty()) continue;
            // a surface whose clamped centre cannot reach the wanted position is dropped to radius 0 at the exact position
            const int R = big ? kR1 : kR0, side = 2 * R + 1;
            void *dJobs, *hJobs, *dOut, *hOut;
            RC(arena.get(w.size() * sizeof(havoc_mi355x_surface_job), &dJobs, &hJobs));
            RC(arena.get(w.size() * size_t(side) * side * 4, &dOut, &hOut));
            havoc_mi355x_surface_job *jobs = static_cast<havoc_mi355x_surface_job *>(hJobs);
            for (size_t k = 0; k < w.size(); ++k)
            {
                const havoc_search_pu &q = pus[w[k].i];
                jobs[k] = {int32_t(fl.a_off[w[k].i]),
                           int32_t(ref_origin + int64_t(q.y0 + w[k].cy) * ref_stride + q.x0 + w[k].cx), q.w, q.h, int32_t(k * size_t(side) * side), {0, 0, 0}};
                state[w[k].i].surfaces.push_back({w[k].cx, w[k].cy, R, static_cast<const int32_t *>(hOut) + k * size_t(side) * side});
            }
            RC(havoc_mi355x_h2d_async(ctx, dJobs, hJobs, w.size() * sizeof(havoc_mi355x_surface_job)));
            RC(havoc_mi355x_sad_surface(ctx, S, R, 64, 64, d_src, src_stride, d_ref, ref_stride, static_cast<const havoc_mi355x_surface_job *>(dJobs),
                                        int(w.size()), static_cast<int32_t *>(dOut)));
            RC(havoc_mi355x_d2h_async(ctx, hOut, dOut, w.size() * size_t(side) * side * 4));
            ++stt.launches;
            (big ? stt.surfaces_large : stt.surfaces_small) += int32_t(w.size());
            stt.bytes_down += int64_t(w.size() * size_t(side) * side * 4);
            w.clear();
        }
        if (!wantSub.empty())
        {
            // 49 quarter-sample positions per search = 4 jobs of <= 16 candidates; one launch per lane-group class of the
            // SATD kernel (rows of 8 samples per PU), as the reference's table is indexed by size
            struct Cls { int lo, hi, mw, mh; };
            static const Cls classes[4] = {{0, 8, 8, 8}, {8, 16, 16, 8}, {16, 32, 16, 16}, {32, 1 << 30, 64, 64}};
            for (const Cls &c : classes)
            {
                std::vector<int> sel;
                for (size_t k = 0; k < wantSub.size(); ++k)
                {
                    const havoc_search_pu &q = pus[wantSub[k].i];
                    const int rows = ((q.w + 7) / 8) * q.h;
                    if (rows > c.lo && rows <= c.hi) sel.push_back(int(k));
                }
                if (sel.empty()) continue;
                void *dJobs, *hJobs, *dOut, *hOut;
                RC(arena.get(sel.size() * 4 * sizeof(havoc_mi355x_satd_multi_job), &dJobs, &hJobs));
                RC(arena.get(sel.size() * 64 * 4, &dOut, &hOut));
                havoc_mi355x_satd_multi_job *jobs = static_cast<havoc_mi355x_satd_multi_job *>(hJobs);
                int32_t *res = static_cast<int32_t *>(hOut);
                for (size_t k = 0; k < sel.size(); ++k)
                {
                    const Want &wn = wantSub[sel[k]];
                    const havoc_search_pu &q = pus[wn.i];
                    SearchState &st = state[wn.i];
                    st.haveSub = true;
                    st.subCx = wn.cx;
                    st.subCy = wn.cy;
                    st.sub = res + k * 64;      // slot c of 49 at [c / 16 * 16 + c % 16]: dense since jobs are consecutive
                    for (int j = 0; j < 4; ++j)
                    {
                        havoc_mi355x_satd_multi_job &mj = jobs[4 * k + j];
                        std::memset(&mj, 0, sizeof(mj));
                        mj.a_off = int32_t(fl.a_off[wn.i]);
                        mj.w = q.w;
                        mj.h = q.h;
                        mj.count = j < 3 ? 16 : 1;
                        for (int e = 0; e < mj.count; ++e)
                        {
                            const int c2 = 16 * j + e;
                            const int qx = wn.cx + c2 % kSubSide - kSub, qy = wn.cy + c2 / kSubSide - kSub;
                            const int X = q.x0 + (qx >> 2), Y = q.y0 + (qy >> 2);
                            // positions whose 8-tap window leaves the padded plane are not in the phase planes: point at the
                            // integer position instead; the value is flagged unusable after the launch
                            const bool ok = X >= -ref_pad + 12 && Y >= -ref_pad + 4 && X + q.w <= W + ref_pad - 12 && Y + q.h <= H + ref_pad - 4;
                            mj.b_off[e] = ok ? int32_t(int64_t(4 * (qy & 3) + (qx & 3)) * plane_elems + phase_origin + int64_t(Y) * ref_stride + X)
                                             : int32_t(phase_origin + int64_t(q.y0) * ref_stride + q.x0);
                        }
                    }
                }
                RC(havoc_mi355x_h2d_async(ctx, dJobs, hJobs, sel.size() * 4 * sizeof(havoc_mi355x_satd_multi_job)));
                RC(havoc_mi355x_satd_multi(ctx, S, c.mw, c.mh, d_src, src_stride, d_phase, ref_stride, static_cast<const havoc_mi355x_satd_multi_job *>(dJobs),
                                           int(sel.size() * 4), static_cast<int32_t *>(dOut)));
                RC(havoc_mi355x_d2h_async(ctx, hOut, dOut, sel.size() * 64 * 4));
                RC(havoc_mi355x_sync(ctx));
                for (size_t k = 0; k < sel.size(); ++k)   // re-flag the positions outside the phase planes
                {
                    const Want &wn = wantSub[sel[k]];
                    const havoc_search_pu &q = pus[wn.i];
                    for (int c2 = 0; c2 < kSubCands; ++c2)
                    {
                        const int qx = wn.cx + c2 % kSubSide - kSub, qy = wn.cy + c2 / kSubSide - kSub;
                        const int X = q.x0 + (qx >> 2), Y = q.y0 + (qy >> 2);
                        if (!(X >= -ref_pad + 12 && Y >= -ref_pad + 4 && X + q.w <= W + ref_pad - 12 && Y + q.h <= H + ref_pad - 4)) res[k * 64 + c2] = -1;
                    }
                }
                ++stt.launches;
                stt.satd_jobs += int32_t(sel.size() * 4);
                stt.bytes_down += int64_t(sel.size() * 64 * 4);
            }
            wantSub.clear();
        }
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
                if (!clampCentre(pus[i], kR1, &cx, &cy) || std::abs(cx - st.miss.x) > kR1 || std::abs(cy - st.miss.y) > kR1) return HAVOC_MI355X_EINVAL;
                wantSurf[1].push_back({i, cx, cy});
            }
            else if (st.miss.kind == 2)
                // the 49 positions are centred on the full-sample vector being refined: the first sub-sample question is that vector
                // itself (uni, Search.hpp:2340-2358) or its (-2, -2) half-sample neighbour (bi, Search.hpp:1627-1650)
                wantSub.push_back({i, ((st.miss.x + 2) >> 2) * 4, ((st.miss.y + 2) >> 2) * 4});
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

void fillUni(const UniResult &r, havoc_search_result &o)
{
    o.mv[0] = r.mv.x; o.mv[1] = r.mv.y;
    o.mvd[0] = r.mvd.x; o.mvd[1] = r.mvd.y;
    o.mv_integer[0] = r.mvInteger.x; o.mv_integer[1] = r.mvInteger.y;
    o.mvp_flag = int16_t(r.mvpFlag);
    o.wrote_2Nx2N = r.wrote2Nx2N;
    o.calls = r.calls;
    o.cost_integer = r.costInteger;
    o.cost_subpel = r.costSubPel;
    o.cost_mvd_zero[0] = r.costMvdZero[0];
    o.cost_mvd_zero[1] = r.costMvdZero[1];
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

// frees the work memory libhavoc_search keeps for a context (call before havoc_mi355x_destroy)
void havoc_search_release(havoc_mi355x_ctx *ctx)
{
    std::lock_guard<std::mutex> lock(g_poolMu);
    auto it = g_pools.find(ctx);
    if (it == g_pools.end()) return;
    it->second.release(ctx);
    g_pools.erase(it);
}

const char *havoc_search_version(void) { return "havoc_search 0.1 (batch client of havoc_mi355x)"; }

} // extern "C"
