// batch_common.hpp -- what the batch clients of libhavoc_search.so share (batch_search.cpp: a list of independent searches;
// picture_search.cpp: a picture's searches in wavefront order with predictors taken from earlier decisions):
//   BatchView      the View of decision.hpp over precomputed data (SAD surfaces, 49-position sub-sample SATD sets); a question it cannot
//                  answer ends the replay with a note of what is missing (longjmp: see batch_search.cpp's header)
//   Launcher       the launches that produce that data: havoc_mi355x_sad_surface per surface size, havoc_mi355x_satd_multi per lane-group
//                  class of the SATD kernel, results in pinned host memory
//   Pool / Arena   a context's work memory (device chunks with pinned host mirrors, kept between calls)
//   ReplayThreads  host threads that live for one call
// Plain C++ on include/havoc_mi355x.h only.
#pragma once

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

extern "C" {
typedef struct
{
    int32_t rounds, launches, surfaces_small, surfaces_large, satd_jobs, replays;
    int64_t bytes_down;
    double seconds_gpu, seconds_host, seconds_total;
} havoc_search_stats;
}

namespace havoc_search {

constexpr int kR1 = 64;             // half-width of the surfaces launched for a miss
constexpr int kSub = 3, kSubSide = 7, kSubCands = 49;

struct Miss
{
    int kind;       // 1: integer position (x, y) needed; 2: sub-sample centre (quarter units) needed; 3: outside the phase planes
    int x, y;
};

struct SurfaceRef
{
    int cx, cy, R;
    const int32_t *data;    // (2R+1)^2, row = dy
};

struct BatchView;

struct SearchState
{
    std::vector<SurfaceRef> surfaces;
    // sub-sample data: PU SATDs of the (2K + 1)^2 quarter-sample positions around (cx, cy) (quarter units, relative to the PU position),
    // -1 where the position's window leaves the phase planes.  The newest set is looked at first; a search keeps a few.
    struct SubSet { int cx, cy, K; const int32_t *data; };
    std::vector<SubSet> subs;
    bool haveSub() const { return !subs.empty(); }
    bool askedAlternatives = false;
    bool done = false;
    int replays = 0;
    Miss miss{0, 0, 0};
    MotionSearch<BatchView>::IntegerStage integer;   // uni search: kept once the integer stage has run to its end
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
        for (size_t k = st.subs.size(); k-- > 0;)
        {
            const SearchState::SubSet &u = st.subs[k];
            if (std::abs(mv.x - u.cx) > u.K || std::abs(mv.y - u.cy) > u.K) continue;
            const int32_t v = u.data[(mv.y - u.cy + u.K) * (2 * u.K + 1) + (mv.x - u.cx + u.K)];
            if (v < 0) miss(3, mv.x, mv.y);   // the position's window leaves the phase planes: cannot be served
            return v;
        }
        miss(2, mv.x, mv.y);
    }
};
// what longjmp passes over on its way out of a replay
static_assert(std::is_trivially_destructible<MotionSearch<BatchView>>::value && std::is_trivially_destructible<PuContext>::value &&
                  std::is_trivially_destructible<UniResult>::value && std::is_trivially_destructible<BatchView>::value,
              "a stopped replay leaves by longjmp: nothing on its stack may need a destructor");

// Replay threads of one call: started once, handed a round's work through run() (each round is a few thousand replays of a few
// microseconds: starting 15 threads per round cost as much as the round).  Rounds follow each other within tens of microseconds, so an
// idle worker first spins on the generation counter (no system call on either side in the steady state) and only then sleeps.
class ReplayThreads
{
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable wake_;
    std::function<void()> work_;
    std::atomic<int> generation_{0}, busy_{0}, sleepers_{0};
    std::atomic<bool> quit_{false};
    static void relax() { __builtin_ia32_pause(); }
    void loop()
    {
        int seen = 0;
        for (;;)
        {
            int spins = 0;
            while (generation_.load(std::memory_order_acquire) == seen && !quit_.load(std::memory_order_relaxed))
            {
                if (++spins < 4000) relax();       // ~40 us
                else
                {
                    std::unique_lock<std::mutex> l(m_);
                    sleepers_.fetch_add(1);
                    wake_.wait(l, [&] { return quit_.load() || generation_.load() != seen; });
                    sleepers_.fetch_sub(1);
                }
            }
            if (quit_.load()) return;
            seen = generation_.load(std::memory_order_acquire);
            work_();
            busy_.fetch_sub(1, std::memory_order_release);
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
    int extra() const { return int(threads_.size()); }
    // every pool thread and the caller run `w` once; returns when all have (work_ is only written while no worker runs: busy_ == 0)
    void run(const std::function<void()> &w)
    {
        work_ = w;
        busy_.store(int(threads_.size()), std::memory_order_relaxed);
        generation_.fetch_add(1, std::memory_order_release);
        if (sleepers_.load() > 0)
        {
            std::lock_guard<std::mutex> l(m_);
            wake_.notify_all();
        }
        w();
        while (busy_.load(std::memory_order_acquire) != 0) relax();
    }
};

inline SearchParams paramsOf(const havoc_search_params &p)
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

inline PuContext puOf(const havoc_search_pu &q)
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

inline void fillUni(const UniResult &r, havoc_search_result &o)
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

inline double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Work memory of a context: device chunks with pinned host mirrors, bump-allocated within one call and kept between calls
// (allocating pinned memory costs milliseconds; a picture's searches need the same amount every time).
struct Pool
{
    struct Chunk { char *dev, *host, *hostDev; size_t cap, used; };   // hostDev: the device's view of the pinned host chunk
    std::vector<Chunk> chunks;
    void reset() { for (Chunk &c : chunks) c.used = 0; }
    int get(havoc_mi355x_ctx *ctx, size_t bytes, void **d, void **h, void **hd = nullptr)
    {
        bytes = (bytes + 255) & ~size_t(255);
        for (Chunk &c : chunks)
            if (c.cap - c.used >= bytes)
            {
                *d = c.dev + c.used;
                *h = c.host + c.used;
                if (hd) *hd = c.hostDev + c.used;
                c.used += bytes;
                return 0;
            }
        Chunk c{nullptr, nullptr, nullptr, std::max(bytes, size_t(64) << 20), 0};
        void *dp = nullptr, *hp = nullptr, *hdp = nullptr;
        int rc = havoc_mi355x_malloc(ctx, &dp, c.cap);
        if (rc) return rc;
        if ((rc = havoc_mi355x_host_alloc(ctx, c.cap, &hp, &hdp)))
        {
            (void)havoc_mi355x_free(ctx, dp);
            return rc;
        }
        c.dev = static_cast<char *>(dp);
        c.host = static_cast<char *>(hp);
        c.hostDev = static_cast<char *>(hdp);
        c.used = bytes;
        chunks.push_back(c);
        *d = c.dev;
        *h = c.host;
        if (hd) *hd = c.hostDev;
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

// the pools of all contexts (defined in batch_search.cpp)
Pool *poolOf(havoc_mi355x_ctx *ctx);

struct Arena   // one call's view of its context's pool (a context runs one call at a time: its launches share one stream)
{
    havoc_mi355x_ctx *ctx;
    Pool *pool;
    explicit Arena(havoc_mi355x_ctx *c) : ctx(c), pool(poolOf(c)) { pool->reset(); }
    int get(size_t bytes, void **d, void **h, void **hd = nullptr) { return pool->get(ctx, bytes, d, h, hd); }
};

#define HAVOC_SEARCH_RC(call) do { const int rc_ = (call); if (rc_) return rc_; } while (0)

// one (block, reference picture) pair as the launches see it
struct Geom
{
    int x0, y0, w, h;
    int64_t a_off;          // sample offset of the source block in the plane holding it
    int64_t ref_origin;     // sample offset of the reference picture's sample (0, 0) from d_ref
    int64_t phase_origin;   // sample offset of its phase plane 0, sample (0, 0), from d_phase
};

struct Want { int i, cx, cy, K; };     // K: half-width of a sub-sample set in quarter samples (surfaces ignore it)

// The launches of a round.  Planes: d_src (stride src_stride) holds every source block; d_ref the reference picture(s) (stride ref_stride,
// `ref_pad` samples of border, picture W x H); d_phase their 16 fractional-sample planes of plane_elems samples each.
struct Launcher
{
    havoc_mi355x_ctx *ctx;
    int S;
    const void *d_src;
    intptr_t src_stride;
    const void *d_ref;
    intptr_t ref_stride;
    int ref_pad;
    const void *d_phase;
    intptr_t plane_elems;
    int W, H;
    Arena *arena;
    havoc_search_stats *stt;
    // direct: the kernels read their job tables from, and write their results to, the pinned host memory itself (its device view) -- no
    // staging copies.  What a round costs is then launches + one synchronisation; right for many small rounds (picture_search.cpp).
    bool direct = false;

    // where a surface of half-width R may be centred so that its window stays inside the padded plane
    bool clampCentre(const Geom &q, int R, int *cx, int *cy) const
    {
        const int loX = -ref_pad + R - q.x0, hiX = W + ref_pad - q.w - R - 4 - q.x0;
        const int loY = -ref_pad + R - q.y0, hiY = H + ref_pad - q.h - R - q.y0;
        if (loX > hiX || loY > hiY) return false;
        *cx = std::min(std::max(*cx, loX), hiX);
        *cy = std::min(std::max(*cy, loY), hiY);
        return true;
    }

    // one SAD-surface launch of half-width R for every entry of w (already clamped); the results are attached to the searches' states
    int surfaces(const std::vector<Want> &w, int R, const Geom *geom, SearchState *state, bool large)
    {
        if (w.empty()) return 0;
        const int side = 2 * R + 1;
        void *dJobs, *hJobs, *dOut, *hOut, *vJobs, *vOut;
        HAVOC_SEARCH_RC(arena->get(w.size() * sizeof(havoc_mi355x_surface_job), &dJobs, &hJobs, &vJobs));
        HAVOC_SEARCH_RC(arena->get(w.size() * size_t(side) * side * 4, &dOut, &hOut, &vOut));
        if (direct)
        {
            dJobs = vJobs;
            dOut = vOut;
        }
        havoc_mi355x_surface_job *jobs = static_cast<havoc_mi355x_surface_job *>(hJobs);
        for (size_t k = 0; k < w.size(); ++k)
        {
            const Geom &q = geom[w[k].i];
            jobs[k] = {int32_t(q.a_off), int32_t(q.ref_origin + int64_t(q.y0 + w[k].cy) * ref_stride + q.x0 + w[k].cx), q.w, q.h,
                       int32_t(k * size_t(side) * side), {0, 0, 0}};
            state[w[k].i].surfaces.push_back({w[k].cx, w[k].cy, R, static_cast<const int32_t *>(hOut) + k * size_t(side) * side});
        }
        if (!direct) HAVOC_SEARCH_RC(havoc_mi355x_h2d_async(ctx, dJobs, hJobs, w.size() * sizeof(havoc_mi355x_surface_job)));
        HAVOC_SEARCH_RC(havoc_mi355x_sad_surface(ctx, S, R, 64, 64, d_src, src_stride, d_ref, ref_stride, static_cast<const havoc_mi355x_surface_job *>(dJobs),
                                                 int(w.size()), static_cast<int32_t *>(dOut)));
        if (!direct) HAVOC_SEARCH_RC(havoc_mi355x_d2h_async(ctx, hOut, dOut, w.size() * size_t(side) * side * 4));
        ++stt->launches;
        (large ? stt->surfaces_large : stt->surfaces_small) += int32_t(w.size());
        stt->bytes_down += int64_t(w.size() * size_t(side) * side * 4);
        return 0;
    }

    bool insidePhasePlanes(const Geom &q, int qx, int qy) const
    {
        const int X = q.x0 + (qx >> 2), Y = q.y0 + (qy >> 2);
        return X >= -ref_pad + 12 && Y >= -ref_pad + 4 && X + q.w <= W + ref_pad - 12 && Y + q.h <= H + ref_pad - 4;
    }

    // the PU SATDs of the (2K + 1)^2 quarter-sample positions around (cx, cy) for every entry of w (K = 3: the 49 positions one sub-sample
    // refinement can touch; K = 7 also covers the integer vector moving by a sample): jobs of <= 16 candidates, one launch per lane-group class
    // of the SATD kernel (rows of 8 samples per PU), as the reference's table is indexed by size.
    // Synchronises the context (the positions outside the phase planes are re-flagged on the host).
    int subSets(const std::vector<Want> &wantSub, const Geom *geom, SearchState *state)
    {
        if (wantSub.empty()) return 0;
        struct Cls { int lo, hi, mw, mh; };
        static const Cls classes[4] = {{0, 8, 8, 8}, {8, 16, 16, 8}, {16, 32, 16, 16}, {32, 1 << 30, 64, 64}};
        struct Batch { std::vector<int> sel; std::vector<int> first; int32_t *res; };
        Batch batch[4];
        for (int ci = 0; ci < 4; ++ci)
        {
            const Cls &c = classes[ci];
            std::vector<int> &sel = batch[ci].sel;
            int njobs = 0;
            for (size_t k = 0; k < wantSub.size(); ++k)
            {
                const Geom &q = geom[wantSub[k].i];
                const int rows = ((q.w + 7) / 8) * q.h;
                if (rows > c.lo && rows <= c.hi)
                {
                    sel.push_back(int(k));
                    batch[ci].first.push_back(njobs);
                    const int side = 2 * wantSub[k].K + 1;
                    njobs += (side * side + 15) / 16;
                }
            }
            if (sel.empty()) continue;
            void *dJobs, *hJobs, *dOut, *hOut, *vJobs, *vOut;
            HAVOC_SEARCH_RC(arena->get(size_t(njobs) * sizeof(havoc_mi355x_satd_multi_job), &dJobs, &hJobs, &vJobs));
            HAVOC_SEARCH_RC(arena->get(size_t(njobs) * 16 * 4, &dOut, &hOut, &vOut));
            if (direct)
            {
                dJobs = vJobs;
                dOut = vOut;
            }
            havoc_mi355x_satd_multi_job *jobs = static_cast<havoc_mi355x_satd_multi_job *>(hJobs);
            int32_t *res = batch[ci].res = static_cast<int32_t *>(hOut);
            for (size_t k = 0; k < sel.size(); ++k)
            {
                const Want &wn = wantSub[sel[k]];
                const Geom &q = geom[wn.i];
                SearchState &st = state[wn.i];
                const int side = 2 * wn.K + 1, ncand = side * side, j0 = batch[ci].first[k];
                if (st.subs.size() >= 6) st.subs.erase(st.subs.begin());
                st.subs.push_back({wn.cx, wn.cy, wn.K, res + size_t(j0) * 16});      // candidate c at [c]: dense since the jobs are consecutive
                for (int j = 0; j * 16 < ncand; ++j)
                {
                    havoc_mi355x_satd_multi_job &mj = jobs[j0 + j];
                    std::memset(&mj, 0, sizeof(mj));
                    mj.a_off = int32_t(q.a_off);
                    mj.w = q.w;
                    mj.h = q.h;
                    mj.count = std::min(16, ncand - 16 * j);
                    for (int e = 0; e < mj.count; ++e)
                    {
                        const int c2 = 16 * j + e;
                        const int qx = wn.cx + c2 % side - wn.K, qy = wn.cy + c2 / side - wn.K;
                        const int X = q.x0 + (qx >> 2), Y = q.y0 + (qy >> 2);
                        // positions whose 8-tap window leaves the padded plane are not in the phase planes: point at the
                        // integer position instead; the value is flagged unusable after the launch
                        mj.b_off[e] = insidePhasePlanes(q, qx, qy)
                                          ? int32_t(int64_t(4 * (qy & 3) + (qx & 3)) * plane_elems + q.phase_origin + int64_t(Y) * ref_stride + X)
                                          : int32_t(q.phase_origin + int64_t(q.y0) * ref_stride + q.x0);
                    }
                }
            }
            if (!direct) HAVOC_SEARCH_RC(havoc_mi355x_h2d_async(ctx, dJobs, hJobs, size_t(njobs) * sizeof(havoc_mi355x_satd_multi_job)));
            HAVOC_SEARCH_RC(havoc_mi355x_satd_multi(ctx, S, c.mw, c.mh, d_src, src_stride, d_phase, ref_stride, static_cast<const havoc_mi355x_satd_multi_job *>(dJobs),
                                                    njobs, static_cast<int32_t *>(dOut)));
            if (!direct) HAVOC_SEARCH_RC(havoc_mi355x_d2h_async(ctx, hOut, dOut, size_t(njobs) * 16 * 4));
            ++stt->launches;
            stt->satd_jobs += njobs;
            stt->bytes_down += int64_t(njobs) * 16 * 4;
        }
        HAVOC_SEARCH_RC(havoc_mi355x_sync(ctx));
        for (int ci = 0; ci < 4; ++ci)      // re-flag the positions outside the phase planes
            for (size_t k = 0; k < batch[ci].sel.size(); ++k)
            {
                const Want &wn = wantSub[batch[ci].sel[k]];
                const Geom &q = geom[wn.i];
                const int side = 2 * wn.K + 1;
                for (int c2 = 0; c2 < side * side; ++c2)
                    if (!insidePhasePlanes(q, wn.cx + c2 % side - wn.K, wn.cy + c2 / side - wn.K)) batch[ci].res[size_t(batch[ci].first[k]) * 16 + c2] = -1;
            }
        return 0;
    }
};

} // namespace havoc_search
