// TEST CLIENT of the decision loops (turingcodec_amd/search/decision.hpp) over the reference's per-block TABLE API.
// Compiled by tests/test_search.py against OUR include/havoc headers and linked
//   * against oracle/_ref/libhavoc_ref.so (the reference's own havoc library)  -> expected decisions
//   * against turingcodec_amd/libhavoc_classic.so (the MI355X implementation)  -> decisions under test (-DHAVOC_CLASSIC_EXT)
// One source, two libraries: the table layouts, accessor functions and populate symbols are the same ABI.
//   * with -DSEARCH_ORACLE: no table API at all, the calls go to the CPU oracle (oracle/liboracle.so) -- the variant the CPU
//     tests use to check the restated loops themselves against the reference's tables.
#include "search_abi.h"

#include <cstring>

#ifndef SEARCH_ORACLE
#include "table_view.hpp"
#include "havoc/pred_intra.h"
#else
#include "decision.hpp"
#include "../oracle/havoc_oracle.h"
#define HAVOC_ALIGN(n, T, v) T v __attribute__((aligned(n)))
typedef struct { void *implementation; } havoc_code;
static havoc_code havoc_new_code(int, int) { return havoc_code{nullptr}; }
static void havoc_delete_code(havoc_code) {}
static int havoc_instruction_set_support() { return 3; }
typedef int havoc_instruction_set;
namespace havoc_search {
template <typename Sample> struct MotionTables { void populate(havoc_code) {} };
template <typename Sample> struct Plane
{
    const Sample *origin;
    intptr_t stride;
    const Sample *at(int x, int y) const { return origin + intptr_t(y) * stride + x; }
};
// the same View interface, every call evaluated by the oracle's plain-C functions
template <typename Sample> struct TableView
{
    const Sample *src;
    intptr_t srcStride;
    Plane<Sample> ref;
    int x0, y0, w, h, bitDepth;
    TableView(MotionTables<Sample> &, const Sample *src_, intptr_t ss, Plane<Sample> ref_, int x0_, int y0_, int w_, int h_, int bd)
        : src(src_), srcStride(ss), ref(ref_), x0(x0_), y0(y0_), w(w_), h(h_), bitDepth(bd) {}
    int sad(int dx, int dy) { return oracle_sad(src, srcStride, ref.at(x0 + dx, y0 + dy), ref.stride, w, h, sizeof(Sample)); }
    void sad4(const Mv d[4], int32_t out[4])
    {
        const void *refs[4];
        for (int i = 0; i < 4; ++i) refs[i] = ref.at(x0 + d[i].x, y0 + d[i].y);
        int sads[4];
        oracle_sad4(src, srcStride, refs, ref.stride, sads, w, h, sizeof(Sample));
        for (int i = 0; i < 4; ++i) out[i] = sads[i];
    }
    int satdQpel(Mv mv)
    {
        Sample buffer[64 * 64];
        oracle_pred_uni(buffer, 64, ref.at(x0 + (mv.x >> 2), y0 + (mv.y >> 2)), ref.stride, w, h, mv.x & 3, mv.y & 3, bitDepth, 8, sizeof(Sample));
        return oracle_pu_satd(src, srcStride, buffer, 64, w, h, sizeof(Sample));
    }
};
template <typename Sample>
void makeIdealPredictor(MotionTables<Sample> &, Sample *ideal, const Sample *input, intptr_t inputStride, Plane<Sample> refOther, Mv mvOther,
                        const LimitFullPelMv &limit, int x0, int y0, int w, int h, int bitDepthY)
{
    Sample other[64 * 64];
    Mv full = shr2(mvOther);
    limit(full);
    oracle_pred_uni(other, 64, refOther.at(x0 + full.x, y0 + full.y), refOther.stride, w, h, mvOther.x & 3, mvOther.y & 3, bitDepthY, 8, sizeof(Sample));
    oracle_subtract_bi(ideal, 64, other, 64, input, inputStride, w, h, 6 + 2 * sizeof(Sample), sizeof(Sample));
}
} // namespace havoc_search
#endif

#ifdef HAVOC_CLASSIC_EXT
#include "havoc_classic_ext.h"
#endif
#include "picture_order.hpp"
#include "amvp.hpp"
#include "merge.hpp"
#include "cand_mode_list.hpp"
#include "tu_decision.hpp"
#ifndef SEARCH_ORACLE
#include "havoc/quantize.h"
#include "havoc/ssd.h"
#include "havoc/transform.h"
#endif

using namespace havoc_search;

namespace {

havoc_code g_code;
bool g_open = false;
MotionTables<uint8_t> g_t8;
MotionTables<uint16_t> g_t16;

template <typename Sample> MotionTables<Sample> &tables();
template <> MotionTables<uint8_t> &tables<uint8_t>() { return g_t8; }
template <> MotionTables<uint16_t> &tables<uint16_t>() { return g_t16; }

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

// every call a loop makes through its View, in order: what tests/trace_tools.py holds against the call sequence the reference encoder's own
// loops made (oracle/trace_hooks.h).  A row = {kind (3 sad, 4 sad4, 5 satd as the trace numbers them), x0, y0, .. x3, y3, value0..3}
struct CallLog
{
    int32_t *rows;
    int64_t capacity, count;
    int32_t *push(int kind)
    {
        static int32_t overflow[13];
        int32_t *r = count < capacity ? rows + 13 * count : overflow;
        ++count;
        std::memset(r, 0, 13 * sizeof(int32_t));
        r[0] = kind;
        return r;
    }
};

template <class Inner>
struct LoggedView
{
    Inner &inner;
    CallLog &log;
    int sad(int dx, int dy)
    {
        const int v = inner.sad(dx, dy);
        int32_t *r = log.push(3);
        r[1] = dx; r[2] = dy; r[9] = v;
        return v;
    }
    void sad4(const Mv d[4], int32_t out[4])
    {
        inner.sad4(d, out);
        int32_t *r = log.push(4);
        for (int i = 0; i < 4; ++i)
        {
            r[1 + 2 * i] = d[i].x; r[2 + 2 * i] = d[i].y; r[9 + i] = out[i];
        }
    }
    int satdQpel(Mv mv)
    {
        const int v = inner.satdQpel(mv);
        int32_t *r = log.push(5);
        r[1] = mv.x; r[2] = mv.y; r[9] = v;
        return v;
    }
};

// The step hooks of decision.hpp (foldPatternStep / foldSubpelStep) as the device view implements them -- a candidate per lane, the winner = the smallest
// of the keys (cost << 2 | index) -- emulated lane by lane on the host, so that the FORMULATION (not the DPP plumbing) is held against the generic loops and the
// reference encoder's traces without a GPU (client_uni_lanes; tests/test_trace_pin.py)
template <class Inner>
struct LaneEmuView
{
    Inner &inner;
    int sad(int dx, int dy) { return inner.sad(dx, dy); }
    void sad4(const Mv d[4], int32_t out[4]) { inner.sad4(d, out); }
    int satdQpel(Mv mv) { return inner.satdQpel(mv); }
    bool patternStep(Mv d0, Mv d1, Mv d2, Mv d3, const PuContext &pu, Lambda lambda, MvCandidate &best)
    {
        const Mv d[4] = {d0, d1, d2, d3};
        int32_t sads[4];
        inner.sad4(d, sads);
        uint64_t key = ~0ull;
        Mv mvds[4];
        int flags[4];
        for (int i = 0; i < 4; ++i)
        {
            const Mv mv = shl2(d[i]);
            const Mv m0 = mv - pu.mvp[0], m1 = mv - pu.mvp[1];
            const Cost c0 = rateOf(m0) + pu.mvpRate[0], c1 = rateOf(m1) + pu.mvpRate[1];
            const bool second = c1 < c0;
            const Cost c = (second ? c1 : c0) + lambda * sads[i];
            mvds[i] = second ? m1 : m0;
            flags[i] = second;
            const uint64_t k = (uint64_t(c) << 2) | uint32_t(i);
            if (k < key) key = k;
        }
        const int w = int(key & 3);
        const Cost cw = Cost(key >> 2);
        if (!(cw < best.cost)) return false;
        best.cost = cw;
        best.mv = shl2(d[w]);
        best.mvd = mvds[w];
        best.mvpFlag = flags[w];
        return true;
    }
    // a candidate's cost as a lane computes it; `flag` and `mvd` of the cheaper predictor
    static Cost laneCost(Mv full, const PuContext &pu, Lambda lambda, int sad, Mv &mvd, int &flag)
    {
        const Mv mv = shl2(full);
        const Mv m0 = mv - pu.mvp[0], m1 = mv - pu.mvp[1];
        const Cost c0 = rateOf(m0) + pu.mvpRate[0], c1 = rateOf(m1) + pu.mvpRate[1];
        flag = c1 < c0;
        mvd = flag ? m1 : m0;
        return (flag ? c1 : c0) + lambda * sad;
    }
    int sadAtFull(Mv p) { return inner.sad(p.x, p.y); }
    int patternRing(Mv origin, const Mv *pattern, int n, int step, int dist, const LimitFullPelMv &limit, const PuContext &pu, Lambda lambda, MvCandidate &best)
    {
        const int count = n / step;
        uint64_t key = ~0ull;
        Mv pos[16], mvds[16];
        int flags[16];
        for (int c = 0; c < count; ++c)
        {
            const Mv p = pattern[c * step];
            pos[c] = Mv(int16_t((origin.x + dist * p.x) / 4), int16_t((origin.y + dist * p.y) / 4));
            limit(pos[c]);
            const Cost cost = laneCost(pos[c], pu, lambda, sadAtFull(pos[c]), mvds[c], flags[c]);
            const uint64_t k = (uint64_t(cost) << 4) | uint32_t(c);
            if (k < key) key = k;
        }
        const int w = int(key & 15);
        if (!(Cost(key >> 4) < best.cost)) return 0;
        best.cost = Cost(key >> 4);
        best.mv = shl2(pos[w]);
        best.mvd = mvds[w];
        best.mvpFlag = flags[w];
        return 1;
    }
    int startProbe(Mv mvQ, int forcedFlag, bool met, bool hexagon, const LimitFullPelMv &limit, const PuContext &pu, Lambda lambda, MvCandidate &best, Cost *costOut, int &calls)
    {
        static const Mv probe[13] = {{0, 0}, {-4, 0}, {0, 4}, {4, 0}, {0, -4}, {0, -8}, {8, -4}, {8, 4}, {0, 8}, {-8, 4}, {-8, -4}, {-8, 4}, {-8, -4}};
        const int count = met ? (hexagon ? 13 : 5) : 1;
        Mv pos[13], mvds[13];
        int flags[13];
        Cost cost[13];
        for (int c = 0; c < count; ++c)
        {   // everything is measured before anything is decided (the device: one exchange)
            pos[c] = Mv(int16_t((mvQ.x + probe[c].x) / 4), int16_t((mvQ.y + probe[c].y) / 4));
            if (c) limit(pos[c]);
            cost[c] = laneCost(pos[c], pu, lambda, sadAtFull(pos[c]), mvds[c], flags[c]);
        }
        if (forcedFlag >= 0)
        {
            flags[0] = forcedFlag;
            mvds[0] = mvQ - pu.mvp[forcedFlag];
            cost[0] = rateOf(mvds[0]) + pu.mvpRate[forcedFlag] + lambda * sadAtFull(pos[0]);
        }
        ++calls;
        if (costOut) *costOut = cost[0];
        if (!(cost[0] < best.cost)) return 0;
        auto take = [&](int w) { best.cost = cost[w]; best.mv = shl2(pos[w]); best.mvd = mvds[w]; best.mvpFlag = flags[w]; };
        take(0);
        if (!met) return 0;
        auto firstMin = [&](int b, int e) { int w = b; for (int c = b + 1; c < e; ++c) if (cost[c] < cost[w]) w = c; return w; };
        ++calls;
        int w = firstMin(1, 5);
        if (cost[w] < best.cost) { take(w); return 0; }
        if (!hexagon) return 1;
        calls += 2;
        w = firstMin(5, 13);
        if (cost[w] < best.cost) { take(w); return 0; }
        return 1;
    }
    void rasterSweep(int rasterSearch, const LimitFullPelMv &limit, const PuContext &pu, Lambda lambda, MvCandidate &best)
    {
        const int rows = 2 * rasterSearch / 20 + 1, perRow = 4 * (2 * rasterSearch / 80 + 1);
        uint64_t key = ~0ull;
        for (int idx = 0; idx < rows * perRow; ++idx)
        {
            const int r = idx / perRow, k = idx - r * perRow;
            Mv p(int16_t(-rasterSearch / 4 + 5 * k), int16_t(-rasterSearch / 4 + 5 * r));
            limit(p);
            Mv mvd;
            int flag;
            const Cost cost = laneCost(p, pu, lambda, sadAtFull(p), mvd, flag);
            const uint64_t kk = (uint64_t(cost) << 10) | uint32_t(idx);
            if (kk < key) key = kk;
        }
        if (!(Cost(key >> 10) < best.cost)) return;
        const int idx = int(key & 1023), r = idx / perRow, k = idx - r * perRow;
        Mv p(int16_t(-rasterSearch / 4 + 5 * k), int16_t(-rasterSearch / 4 + 5 * r));
        limit(p);
        Mv mvd;
        int flag;
        laneCost(p, pu, lambda, 0, mvd, flag);
        best.cost = Cost(key >> 10);
        best.mv = shl2(p);
        best.mvd = mvd;
        best.mvpFlag = flag;
    }
    // searchMotionBi's exhaustive grid as the device view takes it (csrc/kernels_search.hip: DeviceView::biGrid): every candidate's position, the position its SAD is taken
    // at (the quirk of the four-column groups) and its cost computed per "lane", the first of the cheapest on a (cost, index) key
    bool biGrid(Mv origin, int range, const LimitFullPelMv &limit, const PuContext &pu, Lambda lambda, MvCandidate &best)
    {
        const int side = 2 * range + 1, ox = origin.x >> 2, oy = origin.y >> 2;
        uint64_t key = 0x7fffffffffffffffull;
        auto place = [&](int idx, Mv &cand, Mv &sadAt) {
            const int yy = idx / side, xx = idx - yy * side, i = xx & 3;
            Mv first(int16_t(ox + xx - range - i), int16_t(oy + yy - range));
            limit(first);
            sadAt = first;
            cand = first;
            if (i)
            {
                sadAt = Mv(int16_t(first.x + i), first.y);
                limit(sadAt);
                cand = Mv(int16_t(ox + xx - range), int16_t(oy + yy - range));
                limit(cand);
            }
        };
        for (int idx = 0; idx < side * side; ++idx)
        {
            Mv cand, sadAt, mvd;
            int flag;
            place(idx, cand, sadAt);
            const Cost cost = laneCost(cand, pu, lambda, sadAtFull(sadAt), mvd, flag);
            const uint64_t kk = (uint64_t(cost) << 8) | uint32_t(idx);
            if (kk < key) key = kk;
        }
        if (!(Cost(key >> 8) < best.cost)) return true;
        Mv cand, sadAt, mvd;
        int flag;
        place(int(key & 255), cand, sadAt);
        laneCost(cand, pu, lambda, 0, mvd, flag);
        best.cost = Cost(key >> 8);
        best.mv = shl2(cand);
        best.mvd = mvd;
        best.mvpFlag = flag;
        return true;
    }
    int subpelStep(Mv mv, Mv mvd, int scale, bool tryOrigin, Lambda lambda, Cost &bestCost)
    {
        Cost start = bestCost;
        if (tryOrigin) start = rateOf(mvd) + lambda * inner.satdQpel(mv);
        uint64_t key = ~0ull;
        for (int j = 0; j < 8; ++j)
        {
            const int g = j < 4 ? j : j + 1;
            const Mv off(int16_t((g % 3 - 1) * scale), int16_t((g / 3 - 1) * scale));
            const Cost c = rateOf(mvd + off) + lambda * inner.satdQpel(mv + off);
            const uint64_t k = (uint64_t(c) << 4) | uint32_t(j);
            if (k < key) key = k;
        }
        int bestI = -1;
        if (Cost(key >> 4) < start)
        {
            start = Cost(key >> 4);
            bestI = int(key & 15);
        }
        bestCost = start;
        return bestI;
    }
};

template <typename Sample>
void runUni(const Sample *src, intptr_t ss, const Sample *ref, intptr_t rs, const havoc_search_params &p, const havoc_search_pu *pus, int b, int e,
            havoc_search_result *out, CallLog *log = nullptr, int64_t *logFirst = nullptr, bool lanes = false)
{
    const SearchParams sp = paramsOf(p);
    for (int i = b; i < e; ++i)
    {
        const PuContext pu = puOf(pus[i]);
        TableView<Sample> view(tables<Sample>(), src + intptr_t(pu.y0) * ss + pu.x0, ss, Plane<Sample>{ref, rs}, pu.x0, pu.y0, pu.w, pu.h, sp.bitDepth);
        UniResult r;
        if (log)
        {
            logFirst[i - b] = log->count;
            LoggedView<TableView<Sample>> logged{view, *log};
            MotionSearch<LoggedView<TableView<Sample>>> search(sp, pu, logged);
            r = search.run();
            logFirst[i - b + 1] = log->count;
        }
        else if (lanes)
        {
            LaneEmuView<TableView<Sample>> emu{view};
            MotionSearch<LaneEmuView<TableView<Sample>>> search(sp, pu, emu);
            r = search.run();
        }
        else
        {
            MotionSearch<TableView<Sample>> search(sp, pu, view);
            r = search.run();
        }
        havoc_search_result &o = out[i];
        std::memset(&o, 0, sizeof(o));
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
}

template <typename Sample>
void runBi(const Sample *src, intptr_t ss, const Sample *ref, const Sample *refOther, intptr_t rs, const havoc_search_params &p, const havoc_search_pu *pus,
           const int16_t *start, int b, int e, havoc_search_result *out, CallLog *log = nullptr, int64_t *logFirst = nullptr, bool lanes = false)
{
    const SearchParams sp = paramsOf(p);
    for (int i = b; i < e; ++i)
    {
        const PuContext pu = puOf(pus[i]);
        HAVOC_ALIGN(32, Sample, ideal[64 * 64]);
        const LimitFullPelMv limit(pu, sp);
        makeIdealPredictor<Sample>(tables<Sample>(), ideal, src + intptr_t(pu.y0) * ss + pu.x0, ss, Plane<Sample>{refOther, rs},
                                   Mv(pus[i].mv_other[0], pus[i].mv_other[1]), limit, pu.x0, pu.y0, pu.w, pu.h, sp.bitDepth);
        TableView<Sample> view(tables<Sample>(), ideal, 64, Plane<Sample>{ref, rs}, pu.x0, pu.y0, pu.w, pu.h, sp.bitDepth);
        BiResult r;
        if (log)
        {
            logFirst[i - b] = log->count;
            LoggedView<TableView<Sample>> logged{view, *log};
            r = searchMotionBi(sp, pu, logged, Mv(start[2 * i], start[2 * i + 1]));
            logFirst[i - b + 1] = log->count;
        }
        else if (lanes)
        {
            LaneEmuView<TableView<Sample>> emu{view};
            r = searchMotionBi(sp, pu, emu, Mv(start[2 * i], start[2 * i + 1]));
        }
        else
            r = searchMotionBi(sp, pu, view, Mv(start[2 * i], start[2 * i + 1]));
        havoc_search_result &o = out[i];
        std::memset(&o, 0, sizeof(o));
        o.mv[0] = r.mv.x; o.mv[1] = r.mv.y;
        o.mvd[0] = r.mvd.x; o.mvd[1] = r.mvd.y;
        o.mvp_flag = int16_t(r.mvpFlag);
        o.calls = r.calls;
        o.cost_subpel = r.cost;
    }
}

} // namespace

// the 35-mode luma SATD stage of n partitions of one size, one block at a time as PredictIntraLumaBlock does (turing/Reconstruct.cpp:630-701):
// predict into a 32-stride block, then 8x8 (4x4) Hadamard tiles against the source.  jobs = havoc_mi355x_intra_search_job rows.
#ifndef SEARCH_ORACLE
namespace {
havoc::intra::Table<uint8_t> g_i8;
havoc::intra::Table<uint16_t> g_i16;
bool g_intraReady = false;
template <typename Sample> havoc::intra::Table<Sample> &intraTable();
template <> havoc::intra::Table<uint8_t> &intraTable<uint8_t>() { return g_i8; }
template <> havoc::intra::Table<uint16_t> &intraTable<uint16_t>() { return g_i16; }
}
#endif
template <typename Sample>
static void runIntra35(int bitDepth, int log2, const Sample *src, intptr_t ss, const Sample *nb, const int32_t *jobs, int n, int32_t *satd35)
{
    const int N = 1 << log2, ts = log2 == 2 ? 4 : 8;
    HAVOC_ALIGN(32, Sample, pred[32 * 32]);
    for (int i = 0; i < n; ++i)
    {
        const int32_t *j = jobs + 8 * i;
        const uint64_t mask = (uint64_t)(uint32_t)j[3] | ((uint64_t)(uint32_t)j[4] << 32);
        for (int mode = 0; mode < 35; ++mode)
        {
            const Sample *neigh = nb + (((mask >> mode) & 1) ? j[2] : j[1]);
            int c = 0;
#ifndef SEARCH_ORACLE
            intraTable<Sample>().lookup(j[5] ? 0 : 1, bitDepth, log2, mode)(pred, 32, neigh, mode);
            auto satd = *havoc_get_hadamard_satd<Sample>(&tables<Sample>().satd, log2 == 2 ? 2 : 3);
            for (int y = 0; y < N; y += ts)
                for (int x = 0; x < N; x += ts) c += satd(src + j[0] + y * ss + x, ss, pred + y * 32 + x, 32);
#else
            oracle_intra(pred, 32, neigh, log2, mode, (j[5] && log2 < 5) ? 1 : 0, bitDepth, sizeof(Sample));
            for (int y = 0; y < N; y += ts)
                for (int x = 0; x < N; x += ts) c += oracle_satd(src + j[0] + y * ss + x, ss, pred + y * 32 + x, 32, ts, sizeof(Sample));
#endif
            satd35[35 * i + mode] = c;
        }
    }
}

extern "C" {

// mask: the reference library honours it (HAVOC_C_REF | HAVOC_C_OPT = 3: plain C tables; -1: everything the CPU supports = JIT)
int client_open(int mask)
{
    if (!g_open)
    {
        g_code = havoc_new_code(mask < 0 ? havoc_instruction_set_support() : (havoc_instruction_set)mask, 16 << 20);
        g_t8.populate(g_code);
        g_t16.populate(g_code);
        g_open = true;
    }
    return 0;
}

void client_close(void)
{
    if (g_open) havoc_delete_code(g_code);
    g_open = false;
}

// origins: pointer to sample (0, 0) of the padded host planes; strides in samples; jobs [b, e)
int client_uni(int S, const void *src, intptr_t ss, const void *ref, intptr_t rs, const havoc_search_params *p, const havoc_search_pu *pus, int b, int e,
               havoc_search_result *out)
{
    if (!g_open) return -1;
    if (S == 1) runUni<uint8_t>((const uint8_t *)src, ss, (const uint8_t *)ref, rs, *p, pus, b, e, out);
    else runUni<uint16_t>((const uint16_t *)src, ss, (const uint16_t *)ref, rs, *p, pus, b, e, out);
    return 0;
}

// the searches through the step hooks in their lane formulation (LaneEmuView)
int client_uni_lanes(int S, const void *src, intptr_t ss, const void *ref, intptr_t rs, const havoc_search_params *p, const havoc_search_pu *pus, int b, int e,
                     havoc_search_result *out)
{
    if (!g_open) return -1;
    if (S == 1) runUni<uint8_t>((const uint8_t *)src, ss, (const uint8_t *)ref, rs, *p, pus, b, e, out, nullptr, nullptr, true);
    else runUni<uint16_t>((const uint16_t *)src, ss, (const uint16_t *)ref, rs, *p, pus, b, e, out, nullptr, nullptr, true);
    return 0;
}

int client_bi(int S, const void *src, intptr_t ss, const void *ref, const void *refOther, intptr_t rs, const havoc_search_params *p,
              const havoc_search_pu *pus, const int16_t *start, int b, int e, havoc_search_result *out)
{
    if (!g_open) return -1;
    if (S == 1) runBi<uint8_t>((const uint8_t *)src, ss, (const uint8_t *)ref, (const uint8_t *)refOther, rs, *p, pus, start, b, e, out);
    else runBi<uint16_t>((const uint16_t *)src, ss, (const uint16_t *)ref, (const uint16_t *)refOther, rs, *p, pus, start, b, e, out);
    return 0;
}

// the bi-directional refinements through the grid step in its lane formulation (LaneEmuView::biGrid)
int client_bi_lanes(int S, const void *src, intptr_t ss, const void *ref, const void *refOther, intptr_t rs, const havoc_search_params *p, const havoc_search_pu *pus,
                    const int16_t *start, int b, int e, havoc_search_result *out)
{
    if (!g_open) return -1;
    if (S == 1) runBi<uint8_t>((const uint8_t *)src, ss, (const uint8_t *)ref, (const uint8_t *)refOther, rs, *p, pus, start, b, e, out, nullptr, nullptr, true);
    else runBi<uint16_t>((const uint16_t *)src, ss, (const uint16_t *)ref, (const uint16_t *)refOther, rs, *p, pus, start, b, e, out, nullptr, nullptr, true);
    return 0;
}

// the same, with every View call logged: rows = int32 [capacity][13], first = int64 [e - b + 1] (search i's rows are [first[i - b], first[i - b + 1]));
// returns the number of rows the searches made (more than capacity: the log is incomplete)
int64_t client_uni_logged(int S, const void *src, intptr_t ss, const void *ref, intptr_t rs, const havoc_search_params *p, const havoc_search_pu *pus, int b, int e,
                          havoc_search_result *out, int32_t *rows, int64_t capacity, int64_t *first)
{
    if (!g_open) return -1;
    CallLog log{rows, capacity, 0};
    if (S == 1) runUni<uint8_t>((const uint8_t *)src, ss, (const uint8_t *)ref, rs, *p, pus, b, e, out, &log, first);
    else runUni<uint16_t>((const uint16_t *)src, ss, (const uint16_t *)ref, rs, *p, pus, b, e, out, &log, first);
    return log.count;
}

int64_t client_bi_logged(int S, const void *src, intptr_t ss, const void *ref, const void *refOther, intptr_t rs, const havoc_search_params *p,
                         const havoc_search_pu *pus, const int16_t *start, int b, int e, havoc_search_result *out, int32_t *rows, int64_t capacity, int64_t *first)
{
    if (!g_open) return -1;
    CallLog log{rows, capacity, 0};
    if (S == 1) runBi<uint8_t>((const uint8_t *)src, ss, (const uint8_t *)ref, (const uint8_t *)refOther, rs, *p, pus, start, b, e, out, &log, first);
    else runBi<uint16_t>((const uint16_t *)src, ss, (const uint16_t *)ref, (const uint16_t *)refOther, rs, *p, pus, start, b, e, out, &log, first);
    return log.count;
}

// ---- tu_decision.hpp on RECORDED numbers (tests/test_trace_pin.py): the reference encoder's own distortions and rates of the two transform-tree
// candidates of an inter unit / of the refinement candidates of an intra partition go in, the decision must be the encoder's.
// rqt rows (int64 [n][6]): root cbf of the split tree, its weighted ssd (ssd0 + 4 ssd1 + 4 ssd2), its rate (Q16), the unsplit block's weighted ssd,
// its rate, reciprocalLambda (Q16).  out (int32 [n][2]): depth, tried_zero
int client_rqt_decide(const int64_t *rows, int n, int32_t *out)
{
    struct View
    {
        const int64_t *r;
        havoc_tu_outcome evaluate(int x0, int y0, int, int depth)
        {
            havoc_tu_outcome o = havoc_tu_outcome();
            if (depth == 1 && x0 == 0 && y0 == 0)
            {   // the first block of the split tree carries the tree's totals (decideRqt sums the four)
                o.cbf = int32_t(r[0]);
                o.ssd = uint32_t(r[1]);
            }
            else if (depth == 0)
            {
                o.cbf = 1;
                o.ssd = uint32_t(r[3]);
            }
            return o;
        }
    };
    struct Rate
    {
        const int64_t *r;
        Cost operator()(int depth, const havoc_tu_outcome *, int) const { return depth ? r[2] : r[4]; }
    };
    for (int i = 0; i < n; ++i)
    {
        const int64_t *r = rows + 6 * i;
        View view{r};
        havoc_rqt_cu cu = {0, 0, 5, 0};
        Lambda rl;
        rl.value = int32_t(r[5]);
        const havoc_rqt_result res = decideRqt(view, cu, rl, Rate{r});
        out[2 * i] = res.depth;
        out[2 * i + 1] = res.tried_zero;
    }
    return 0;
}

// The neighbour cells k_search_rows keeps in LDS (csrc/kernels_search.hip: Lds::mv / valid, load_neighbours, search_ctu's `get`) restated on the host and held against the
// picture-wide field: every PU's predictors derived through both must be the same (fake decided vectors: a hash of (PU, list)).  Returns the number of derivations that differ;
// example[0..7] = p, list, the two predictors through the field, the two through the emulated LDS (packed)
int client_check_lds_neighbours(const havoc_picture_pu *pus, const int32_t *ctu_first, int ctus_x, int ctus_y, int pic_w, int pic_h, int ctb, int32_t *example)
{
    MotionField field;
    field.init(pic_w, pic_h);
    int bad = 0;
    for (int cy = 0; cy < ctus_y; ++cy)
        for (int cx = 0; cx < ctus_x; ++cx)
        {
            const int c = cy * ctus_x + cx, xCtb = cx * ctb, yCtb = cy * ctb;
            for (int list = 0; list < 2; ++list)
            {
                int32_t mv[256 + 36];
                uint8_t valid[256 + 36];
                std::memset(mv, 0, sizeof(mv));
                std::memset(valid, 0, sizeof(valid));
                for (int t = 0; t < 34; ++t)      // load_neighbours(left = true, top = true): the row walk keeps the left column instead, the same values
                {
                    const int gx = t < 16 ? cx * 16 - 1 : (t < 32 ? cx * 16 + t - 16 : (t == 32 ? cx * 16 - 1 : cx * 16 + 16));
                    const int gy = t < 16 ? cy * 16 + t : cy * 16 - 1;
                    const bool in = gx >= 0 && gy >= 0 && gx < field.cw && gy < field.ch;
                    mv[256 + t] = in ? field.mv[list][size_t(gy) * field.cw + gx] : 0;
                    valid[256 + t] = in ? field.valid[list][size_t(gy) * field.cw + gx] : 0;
                }
                auto getLds = [&](int, int px, int py, Mv *v) {
                    const int rx = px - xCtb, ry = py - yCtb;
                    int i;
                    if (rx >= 0 && ry >= 0 && rx < 64 && ry < 64) i = (ry >> 2) * 16 + (rx >> 2);
                    else if (rx >= -4 && rx < 0 && ry >= 0 && ry < 64) i = 256 + (ry >> 2);
                    else if (ry >= -4 && ry < 0 && rx >= 0 && rx < 64) i = 272 + (rx >> 2);
                    else if (ry >= -4 && ry < 0 && rx >= -4 && rx < 0) i = 288;
                    else if (ry >= -4 && ry < 0 && rx >= 64 && rx < 68) i = 289;
                    else
                        return false;
                    if (!valid[i]) return false;
                    *v = MotionField::unpack(mv[i]);
                    return true;
                };
                // the field as it is while THIS list's walk of the CTU runs: a private copy that receives this CTU's decisions
                MotionField f2 = field;
                auto getField = [&](int l, int x, int y, Mv *v) { return f2.get(l, x, y, v); };
                for (int p = ctu_first[c]; p < ctu_first[c + 1]; ++p)
                {
                    Mv a[2], b[2];
                    derivePredictors(pus[p], list, ctb, pic_w, pic_h, getField, a);
                    derivePredictors(pus[p], list, ctb, pic_w, pic_h, getLds, b);
                    if (a[0] != b[0] || a[1] != b[1])
                    {
                        if (!bad && example)
                        {
                            example[0] = p; example[1] = list;
                            example[2] = MotionField::pack(a[0]); example[3] = MotionField::pack(a[1]);
                            example[4] = MotionField::pack(b[0]); example[5] = MotionField::pack(b[1]);
                        }
                        ++bad;
                    }
                    const uint32_t hsh = uint32_t(p) * 2654435761u + uint32_t(list) * 40503u;
                    const Mv v(int16_t((hsh >> 8) % 61) - 30, int16_t((hsh >> 16) % 41) - 20);
                    f2.set(list, pus[p].x0, pus[p].y0, pus[p].w, pus[p].h, v);
                    const int cw4 = pus[p].w >> 2;
                    for (int t = 0; t < cw4 * (pus[p].h >> 2); ++t)
                    {
                        const int gx = (pus[p].x0 >> 2) + t % cw4, gy = (pus[p].y0 >> 2) + t / cw4;
                        const int li = (gy - (yCtb >> 2)) * 16 + gx - (xCtb >> 2);
                        mv[li] = MotionField::pack(v);
                        valid[li] = 1;
                    }
                }
                for (int p = ctu_first[c]; p < ctu_first[c + 1]; ++p)
                {
                    const uint32_t hsh = uint32_t(p) * 2654435761u + uint32_t(list) * 40503u;
                    field.set(list, pus[p].x0, pus[p].y0, pus[p].w, pus[p].h, Mv(int16_t((hsh >> 8) % 61) - 30, int16_t((hsh >> 16) % 41) - 20));
                }
            }
        }
    return bad;
}

// cand_mode_list.hpp on recorded neighbour modes: ab (int32 [n][2]) -> out (int32 [n][4]): candModeList[0..2], neighbourModes
int client_cand_mode_list(const int32_t *ab, int n, int32_t *out)
{
    for (int i = 0; i < n; ++i)
    {
        int cand[3];
        out[4 * i + 3] = candModeListOf(ab[2 * i], ab[2 * i + 1], cand);
        out[4 * i] = cand[0]; out[4 * i + 1] = cand[1]; out[4 * i + 2] = cand[2];
    }
    return 0;
}

// picture_order.hpp: neighbourPositionAvailable for the five predictor positions of recorded prediction units: rows (int32 [n][8]): x0, y0, w, h, ctb size, picture width,
// height, 0.  out (int32 [n][5]): A0, A1, B0, B1, B2 may be read
int client_positions_available(const int32_t *rows, int n, int32_t *out)
{
    for (int i = 0; i < n; ++i)
    {
        const int32_t *r = rows + 8 * i;
        havoc_picture_pu q = havoc_picture_pu();
        q.x0 = r[0]; q.y0 = r[1]; q.w = r[2]; q.h = r[3];
        const int xN[5] = {q.x0 - 1, q.x0 - 1, q.x0 + q.w, q.x0 + q.w - 1, q.x0 - 1}, yN[5] = {q.y0 + q.h, q.y0 + q.h - 1, q.y0 - 1, q.y0 - 1, q.y0 - 1};
        for (int k = 0; k < 5; ++k) out[5 * i + k] = neighbourPositionAvailable(q, r[4], r[5], r[6], xN[k], yN[k]) ? 1 : 0;
    }
    return 0;
}

// amvp.hpp: deriveTemporalCandidate on recorded inputs: rows (int32 [n][36]): X, current POC, target POC, POC of the collocated picture, allBackwards, collocated_from_l0 |
// xPb, yPb, nPbW, nPbH, picture width, height, CtbLog2SizeY, 0 | the bottom-right cell, the centre cell: predFlag0, predFlag1, mv0.x, mv0.y, mv1.x, mv1.y, refPoc0, refPoc1,
// longTerm0, longTerm1.  out (int32 [n][3]): available, x, y
int client_temporal(const int32_t *rows, int n, int32_t *out)
{
    auto unpack = [](const int32_t *q) {
        ColocatedCell c;
        c.predFlag[0] = q[0] != 0; c.predFlag[1] = q[1] != 0;
        c.mv[0] = Mv(int16_t(q[2]), int16_t(q[3])); c.mv[1] = Mv(int16_t(q[4]), int16_t(q[5]));
        c.refPoc[0] = q[6]; c.refPoc[1] = q[7];
        c.longTerm[0] = q[8] != 0; c.longTerm[1] = q[9] != 0;
        return c;
    };
    for (int i = 0; i < n; ++i)
    {
        const int32_t *r = rows + 36 * i;
        Mv v;
        const bool ok = deriveTemporalCandidate(r[6], r[7], r[8], r[9], r[10], r[11], r[12], unpack(r + 14), unpack(r + 24), r[0], r[3], r[1], r[2], r[4] != 0, r[5] != 0, &v);
        out[3 * i] = ok; out[3 * i + 1] = v.x; out[3 * i + 2] = v.y;
    }
    return 0;
}

// merge.hpp on recorded inputs: rows (int32 [n][64]): partIdx, nPbW, nPbH, slice is B, active references of L0, of L1, MaxNumMergeCand, temporal candidate available |
// per neighbour A1, B1, B0, A0, B2: predFlag0, predFlag1, refIdx0, refIdx1, mv0.x, mv0.y, mv1.x, mv1.y | the temporal candidate likewise | POC of L0[0..3], L1[0..3].
// out (int32 [n][40]): five candidates in the same eight-value form (beyond MaxNumMergeCand: zeros)
int client_merge(const int32_t *rows, int n, int32_t *out)
{
    auto unpack = [](const int32_t *q) {
        MergeCandidate c;
        c.predFlag[0] = q[0] != 0; c.predFlag[1] = q[1] != 0;
        c.refIdx[0] = q[2]; c.refIdx[1] = q[3];
        c.mv[0] = Mv(int16_t(q[4]), int16_t(q[5])); c.mv[1] = Mv(int16_t(q[6]), int16_t(q[7]));
        return c;
    };
    for (int i = 0; i < n; ++i)
    {
        const int32_t *r = rows + 64 * i;
        MergeCandidate nb[5], list[5];
        for (int k = 0; k < 5; ++k) nb[k] = unpack(r + 8 + 8 * k);
        const MergeCandidate col = unpack(r + 48);
        const int maxCand = r[6] < 5 ? r[6] : 5;
        deriveMergeCandidates(nb, r[0], r[1], r[2], r[1], r[2], r[7] != 0, col, r[3] != 0, r[4], r[5], r + 56, r + 60, maxCand, list);      // (Log2ParMrgLevel = 2 in every trace: the unit is its own)
        int32_t *o = out + 40 * i;
        std::memset(o, 0, 40 * sizeof(int32_t));
        for (int k = 0; k < maxCand; ++k)
        {
            const MergeCandidate &c = list[k];
            int32_t *q = o + 8 * k;
            q[0] = c.predFlag[0]; q[1] = c.predFlag[1];
            q[2] = c.predFlag[0] ? c.refIdx[0] : 0; q[3] = c.predFlag[1] ? c.refIdx[1] : 0;
            q[4] = c.predFlag[0] ? c.mv[0].x : 0; q[5] = c.predFlag[0] ? c.mv[0].y : 0; q[6] = c.predFlag[1] ? c.mv[1].x : 0; q[7] = c.predFlag[1] ? c.mv[1].y : 0;
        }
    }
    return 0;
}

// amvp.hpp on recorded inputs: rows (int32 [n][4 + 5 * 9 + 3]): X, current POC, target POC, 0 | per neighbour A0, A1, B0, B1, B2: available, predFlag0, predFlag1, poc0,
// poc1, mv0.x, mv0.y, mv1.x, mv1.y | temporal candidate available, x, y.  out (int32 [n][4]): mvp[0].x, .y, mvp[1].x, .y
int client_amvp(const int32_t *rows, int n, int32_t *out)
{
    for (int i = 0; i < n; ++i)
    {
        const int32_t *r = rows + 52 * i;
        AmvpNeighbour nb[5];
        for (int k = 0; k < 5; ++k)
        {
            const int32_t *q = r + 4 + 9 * k;
            nb[k].available = q[0] != 0;
            nb[k].predFlag[0] = q[1] != 0;
            nb[k].predFlag[1] = q[2] != 0;
            nb[k].refPoc[0] = q[3];
            nb[k].refPoc[1] = q[4];
            nb[k].mv[0] = Mv(int16_t(q[5]), int16_t(q[6]));
            nb[k].mv[1] = Mv(int16_t(q[7]), int16_t(q[8]));
        }
        Mv mvp[2];
        deriveAmvp(r[0], r[1], r[2], nb, r[49] != 0, Mv(int16_t(r[50]), int16_t(r[51])), mvp);
        out[4 * i] = mvp[0].x; out[4 * i + 1] = mvp[0].y; out[4 * i + 2] = mvp[1].x; out[4 * i + 3] = mvp[1].y;
    }
    return 0;
}

// intra: per partition `count[i]` candidates in refinement order; cand rows (int64 [total][3]): mode, ssd, rate (Q16; -1 = the encoder did not measure it);
// rl[i] = reciprocalLambda (Q16).  out (int32 [n][2]): champion's mode, its index in the order
int client_intra_rd_decide(const int64_t *cand, const int32_t *count, const int32_t *rl, int n, int32_t *out)
{
    struct View
    {
        const int64_t *c;
        havoc_tu_outcome evaluate(int, int j)
        {
            havoc_tu_outcome o = havoc_tu_outcome();
            o.ssd = uint32_t(c[3 * j + 1]);
            return o;
        }
    };
    struct Rate
    {
        const int64_t *c;
        Cost operator()(int, int j, const havoc_tu_outcome &) const { return c[3 * j + 2] < 0 ? kCostMax : Cost(c[3 * j + 2]); }
    };
    int64_t at = 0;
    for (int i = 0; i < n; ++i)
    {
        havoc_search_intra_result order = havoc_search_intra_result();
        order.count = count[i];
        for (int j = 0; j < count[i]; ++j) order.order[j] = int32_t(cand[3 * (at + j)]);
        View view{cand + 3 * at};
        Lambda l;
        l.value = rl[i];
        const havoc_intra_rd_result r = decideIntraRd(view, order, l, Rate{cand + 3 * at});
        out[2 * i] = r.mode;
        out[2 * i + 1] = r.index;
        at += count[i];
    }
    return 0;
}

int client_intra35(int S, int bitDepth, int log2, const void *src, intptr_t ss, const void *nb, const int32_t *jobs, int n, int32_t *satd35)
{
    if (!g_open) return -1;
#ifndef SEARCH_ORACLE
    if (!g_intraReady)
    {
        g_i8.populate(g_code);
        g_i16.populate(g_code);
        g_intraReady = true;
    }
#endif
    if (S == 1) runIntra35<uint8_t>(bitDepth, log2, (const uint8_t *)src, ss, (const uint8_t *)nb, jobs, n, satd35);
    else runIntra35<uint16_t>(bitDepth, log2, (const uint16_t *)src, ss, (const uint16_t *)nb, jobs, n, satd35);
    return 0;
}

// a whole picture's uni-directional searches in dependency order (picture_order.hpp), one table call at a time: the expected values of
// havoc_search_picture_uni.  ref0 / ref1 = sample (0, 0) of the two reference pictures; out[2 * p + list]; field_out as there.
int client_picture_uni(int S, const void *src, intptr_t ss, const void *ref0, const void *ref1, intptr_t rs, const havoc_search_params *p,
                       const havoc_picture_pu *pus, const int32_t *ctu_first, int ctus_x, int ctus_y, const int64_t *mvp_rate, havoc_search_result *out,
                       int16_t *field_out, havoc_search_result *out_bi)
{
    if (!g_open) return -1;
    const SearchParams sp = paramsOf(*p);
    const Cost rate[2] = {mvp_rate[0], mvp_rate[1]};
    MotionField field;
    auto run = [&](auto sampleTag) {
        typedef decltype(sampleTag) Sample;
        const Sample *refs[2] = {(const Sample *)ref0, (const Sample *)ref1};
        auto search = [&](int pi, int list, const PuContext &pu) {
            (void)pi;
            TableView<Sample> view(tables<Sample>(), (const Sample *)src + intptr_t(pu.y0) * ss + pu.x0, ss, Plane<Sample>{refs[list], rs}, pu.x0, pu.y0, pu.w, pu.h,
                                   sp.bitDepth);
            MotionSearch<TableView<Sample>> ms(sp, pu, view);
            return ms.run();
        };
        // the bi-directional refinement one table call at a time (as runBi above): ideal predictor from the other list's vector, then searchMotionBi
        auto bi = [&](int pi, int list, const PuContext &pu, Mv other, Mv start) {
            (void)pi;
            HAVOC_ALIGN(32, Sample, ideal[64 * 64]);
            const LimitFullPelMv limit(pu, sp);
            makeIdealPredictor<Sample>(tables<Sample>(), ideal, (const Sample *)src + intptr_t(pu.y0) * ss + pu.x0, ss, Plane<Sample>{refs[1 - list], rs}, other, limit,
                                       pu.x0, pu.y0, pu.w, pu.h, sp.bitDepth);
            TableView<Sample> view(tables<Sample>(), ideal, 64, Plane<Sample>{refs[list], rs}, pu.x0, pu.y0, pu.w, pu.h, sp.bitDepth);
            return searchMotionBi(sp, pu, view, start);
        };
        walkPictureSequential(sp, pus, ctu_first, ctus_x, ctus_y, rate, search, out, field, bi, out_bi);
    };
    if (S == 1) run(uint8_t(0));
    else run(uint16_t(0));
    if (field_out)
        for (int l = 0; l < 2; ++l)
            for (size_t c = 0; c < size_t(field.cw) * field.ch; ++c)
            {
                const Mv v = MotionField::unpack(field.mv[l][c]);
                field_out[(size_t(l) * field.cw * field.ch + c) * 2 + 0] = field.valid[l][c] ? v.x : 0;
                field_out[(size_t(l) * field.cw * field.ch + c) * 2 + 1] = field.valid[l][c] ? v.y : 0;
            }
    return 0;
}

} // extern "C"

#if !defined(HAVOC_CLASSIC_EXT)
// ---- the residual-quadtree decisions (tu_decision.hpp) one block at a time: residual -> forward transform -> Rdoq::runQuantisation ->
// de-quantise -> inverse transform + add -> SSD, through the reference's tables and its own Rdoq.cpp (oracle/ref_shim_rdoq.cpp), or the
// CPU oracle with -DSEARCH_ORACLE.  The expected values of havoc_search_rqt.  rec receives the chosen candidates' reconstruction.
#ifndef SEARCH_ORACLE
extern "C" int ref_rdoq(int16_t *dst, const int16_t *src, int log2Size, int cIdx, int scanIdx, int isIntra, int sdh, int quantScale, int quantShift, int invScale,
                        int bitDepth, double lambda, const uint8_t *states);
namespace {
struct TuTables
{
    havoc::table_transform<8> t8;
    havoc::table_transform<10> t10;
    havoc::table_inverse_transform_add<uint8_t> ita8;
    havoc::table_inverse_transform_add<uint16_t> ita16;
    havoc_table_quantize_inverse qi;
    havoc_table_ssd<uint8_t> ssd8;
    havoc_table_ssd<uint16_t> ssd16;
    bool ready = false;
} g_tu;
}
#else
extern "C" {
int oracle_rdoq(int16_t *dst, const int16_t *src, int log2Size, int cIdx, int scanIdx, int isIntra, int sdh, int quantScale, int quantShift, int invScale,
                int bitDepth, int32_t lambdaQ16, int32_t sdhFactor, const uint8_t *states);
void oracle_rdoq_lambda(double lambda, int invQuantScale, int32_t *lambdaQ16, int32_t *sdhFactor);
}
#endif

template <typename Sample>
struct PerCallTuView
{
    int bitDepth, sdh;
    const Sample *src;      // sample (0, 0)
    intptr_t ss;
    const Sample *pred;
    intptr_t ps;
    Sample *rec;            // sample (0, 0) of the reconstruction picture: written by `commit`
    intptr_t rs;
    const uint8_t *states;
    const havoc_rqt_quant *quant;
    double lambda;
    int ctxIndex;
    // pieces of the candidates of the unit being decided: [depth][k]
    Sample piece[2][4][32 * 32];

    havoc_tu_outcome evaluate(int x0, int y0, int log2, int depth)
    {
        const int n = 1 << log2, k = depth ? kNext++ & 3 : 0;
        HAVOC_ALIGN(32, int16_t, res[32 * 32]);
        HAVOC_ALIGN(32, int16_t, coef[32 * 32]);
        HAVOC_ALIGN(32, int16_t, level[32 * 32]);
        HAVOC_ALIGN(32, int16_t, deq[32 * 32]);
        const Sample *s = src + intptr_t(y0) * ss + x0, *p = pred + intptr_t(y0) * ps + x0;
        for (int y = 0; y < n; ++y)
            for (int x = 0; x < n; ++x) res[y * n + x] = int16_t(s[y * ss + x] - p[y * ps + x]);      // Reconstruct.cpp:1274-1286
        const havoc_rqt_quant &q = quant[log2 - 2];
        havoc_tu_outcome o;
        Sample *out = piece[depth][k];
#ifndef SEARCH_ORACLE
        constexpr int tableDepth = 2 * sizeof(Sample) + 6;
        havoc::Transform *fwd = sizeof(Sample) == 1 ? *havoc::get_transform<8>(&g_tu.t8, 0, log2) : *havoc::get_transform<10>(&g_tu.t10, 0, log2);
        (void)tableDepth;
        fwd(coef, res, n);
        o.cbf = ref_rdoq(level, coef, log2, 0, 0, 0, sdh, q.quant_scale, q.quant_shift, q.inv_scale, bitDepth, lambda, states + 128 * ctxIndex);
        (*havoc_get_quantize_inverse(&g_tu.qi, q.inv_scale, q.inv_shift))(deq, level, q.inv_scale, q.inv_shift, n * n);
        itAdd(out, n, p, ps, deq, log2);
        o.ssd = uint32_t(ssdOf(s, ss, out, n, log2));
#else
        oracle_transform(coef, res, n, log2, 0, bitDepth);
        int32_t lq, sf;
        oracle_rdoq_lambda(lambda, q.inv_scale, &lq, &sf);
        o.cbf = oracle_rdoq(level, coef, log2, 0, 0, 0, sdh, q.quant_scale, q.quant_shift, q.inv_scale, bitDepth, lq, sf, states + 128 * ctxIndex);
        oracle_quantize_inverse(deq, level, q.inv_scale, q.inv_shift, n * n);
        oracle_inverse_transform_add(out, n, p, ps, deq, log2, 0, bitDepth, sizeof(Sample));
        o.ssd = oracle_ssd(s, ss, out, n, n, n, sizeof(Sample));
#endif
        o.nonzero = o.sum_abs = 0;
        for (int i = 0; i < n * n; ++i)
        {
            o.nonzero += level[i] != 0;
            o.sum_abs += level[i] < 0 ? -level[i] : level[i];
        }
        return o;
    }
#ifndef SEARCH_ORACLE
    void itAdd(uint8_t *dst, intptr_t sd, const uint8_t *p, intptr_t sp, const int16_t *c, int log2) { (*havoc::get_inverse_transform_add<uint8_t>(&g_tu.ita8, 0, log2))(dst, sd, p, sp, c, bitDepth); }
    void itAdd(uint16_t *dst, intptr_t sd, const uint16_t *p, intptr_t sp, const int16_t *c, int log2) { (*havoc::get_inverse_transform_add<uint16_t>(&g_tu.ita16, 0, log2))(dst, sd, p, sp, c, bitDepth); }
    int ssdOf(const uint8_t *a, intptr_t sa, const uint8_t *b, intptr_t sb, int log2) { return (*havoc_get_ssd<uint8_t>(&g_tu.ssd8, log2))(a, sa, b, sb, 1 << log2, 1 << log2); }
    int ssdOf(const uint16_t *a, intptr_t sa, const uint16_t *b, intptr_t sb, int log2) { return (*havoc_get_ssd<uint16_t>(&g_tu.ssd16, log2))(a, sa, b, sb, 1 << log2, 1 << log2); }
#endif
    int kNext = 0;

    void commit(const havoc_rqt_cu &cu, const havoc_rqt_result &r)
    {
        const int n = 1 << cu.log2_size, half = n / 2;
        if (r.depth == 0 && r.tried_zero)
            for (int y = 0; y < n; ++y) std::memcpy(rec + intptr_t(cu.y0 + y) * rs + cu.x0, piece[0][0] + y * n, n * sizeof(Sample));
        else      // split, or the uncoded short-cut (whose four pieces equal the prediction)
            for (int k = 0; k < 4; ++k)
                for (int y = 0; y < half; ++y)
                    std::memcpy(rec + intptr_t(cu.y0 + (k >> 1) * half + y) * rs + cu.x0 + (k & 1) * half, piece[1][k] + y * half, half * sizeof(Sample));
    }
};

// ---- the RD refinement of intra partitions (tu_decision.hpp: decideIntraRd) one candidate at a time through this back end's intra table, TU
// tables and Rdoq: the expected values of havoc_search_intra_rd.  jobs = havoc_mi355x_intra_search_job rows (8 int32); src / nb flat planes.
template <typename Sample>
struct PerCallIntraView
{
    int bitDepth, log2, sdh, ctxIndex;
    const Sample *src;      // the partition's source block
    intptr_t ss;
    const Sample *nbU, *nbF;
    uint64_t filt;
    int edge;
    const uint8_t *states;
    const havoc_rqt_quant *quant;
    double lambda;
    Sample rec[36][32 * 32];      // per candidate index

    havoc_tu_outcome evaluate(int mode, int index)
    {
        const int n = 1 << log2, tr = log2 == 2 ? 1 : 0;
        HAVOC_ALIGN(32, Sample, pred[32 * 32]);
        HAVOC_ALIGN(32, int16_t, res[32 * 32]);
        HAVOC_ALIGN(32, int16_t, coef[32 * 32]);
        HAVOC_ALIGN(32, int16_t, level[32 * 32]);
        HAVOC_ALIGN(32, int16_t, deq[32 * 32]);
        const Sample *nb = ((filt >> mode) & 1) ? nbF : nbU;
        const int scan = intraScanIdx(log2, mode);
        havoc_tu_outcome o;
        Sample *out = rec[index];
#ifndef SEARCH_ORACLE
        intraTable<Sample>().lookup(edge ? 0 : 1, bitDepth, log2, mode)(pred, n, nb, mode);
#else
        oracle_intra(pred, n, nb, log2, mode, (edge && log2 < 5) ? 1 : 0, bitDepth, sizeof(Sample));
#endif
        for (int y = 0; y < n; ++y)
            for (int x = 0; x < n; ++x) res[y * n + x] = int16_t(src[y * ss + x] - pred[y * n + x]);      // Reconstruct.cpp:258-260
#ifndef SEARCH_ORACLE
        havoc::Transform *fwd = sizeof(Sample) == 1 ? *havoc::get_transform<8>(&g_tu.t8, tr, log2) : *havoc::get_transform<10>(&g_tu.t10, tr, log2);
        fwd(coef, res, n);
        o.cbf = ref_rdoq(level, coef, log2, 0, scan, 1, sdh, quant->quant_scale, quant->quant_shift, quant->inv_scale, bitDepth, lambda, states + 128 * ctxIndex);
        (*havoc_get_quantize_inverse(&g_tu.qi, quant->inv_scale, quant->inv_shift))(deq, level, quant->inv_scale, quant->inv_shift, n * n);
        itAdd(out, n, pred, n, deq, tr);
        o.ssd = uint32_t(ssdOf(src, ss, out, n));
#else
        oracle_transform(coef, res, n, log2, tr, bitDepth);
        int32_t lq, sf;
        oracle_rdoq_lambda(lambda, quant->inv_scale, &lq, &sf);
        o.cbf = oracle_rdoq(level, coef, log2, 0, scan, 1, sdh, quant->quant_scale, quant->quant_shift, quant->inv_scale, bitDepth, lq, sf, states + 128 * ctxIndex);
        oracle_quantize_inverse(deq, level, quant->inv_scale, quant->inv_shift, n * n);
        oracle_inverse_transform_add(out, n, pred, n, deq, log2, tr, bitDepth, sizeof(Sample));
        o.ssd = oracle_ssd(src, ss, out, n, n, n, sizeof(Sample));
#endif
        o.nonzero = o.sum_abs = 0;
        for (int i = 0; i < n * n; ++i)
        {
            o.nonzero += level[i] != 0;
            o.sum_abs += level[i] < 0 ? -level[i] : level[i];
        }
        return o;
    }
#ifndef SEARCH_ORACLE
    void itAdd(uint8_t *dst, intptr_t sd, const uint8_t *p, intptr_t sp, const int16_t *c, int tr) { (*havoc::get_inverse_transform_add<uint8_t>(&g_tu.ita8, tr, log2))(dst, sd, p, sp, c, bitDepth); }
    void itAdd(uint16_t *dst, intptr_t sd, const uint16_t *p, intptr_t sp, const int16_t *c, int tr) { (*havoc::get_inverse_transform_add<uint16_t>(&g_tu.ita16, tr, log2))(dst, sd, p, sp, c, bitDepth); }
    int ssdOf(const uint8_t *a, intptr_t sa, const uint8_t *b, intptr_t sb) { return (*havoc_get_ssd<uint8_t>(&g_tu.ssd8, log2))(a, sa, b, sb, 1 << log2, 1 << log2); }
    int ssdOf(const uint16_t *a, intptr_t sa, const uint16_t *b, intptr_t sb) { return (*havoc_get_ssd<uint16_t>(&g_tu.ssd16, log2))(a, sa, b, sb, 1 << log2, 1 << log2); }
#endif
};

static void readyTuTables()
{
#ifndef SEARCH_ORACLE
    if (!g_tu.ready)
    {
        havoc::populate_transform<8>(&g_tu.t8, g_code);
        havoc::populate_transform<10>(&g_tu.t10, g_code);
        havoc::populate_inverse_transform_add<uint8_t>(&g_tu.ita8, g_code, 1);
        havoc::populate_inverse_transform_add<uint16_t>(&g_tu.ita16, g_code, 1);
        havoc_populate_quantize_inverse(&g_tu.qi, g_code);
        havoc_populate_ssd<uint8_t>(&g_tu.ssd8, g_code);
        havoc_populate_ssd<uint16_t>(&g_tu.ssd16, g_code);
        g_tu.ready = true;
    }
    if (!g_intraReady)
    {
        g_i8.populate(g_code);
        g_i16.populate(g_code);
        g_intraReady = true;
    }
#endif
}

extern "C" int client_intra_rd(int S, int bitDepth, int log2, const void *src, intptr_t ss, const void *nb, const int32_t *jobs, int n,
                               const havoc_search_intra_result *order, const havoc_search_intra_ctx *ictx, const int32_t *ctx_index, const uint8_t *states,
                               const havoc_rqt_quant *quant, double lambda, double reciprocal_lambda, int sdh, void *rec, havoc_intra_rd_result *out)
{
    if (!g_open) return -1;
    readyTuTables();
    Lambda rl;
    rl.set(reciprocal_lambda);
    auto run = [&](auto tag) {
        typedef decltype(tag) Sample;
        static PerCallIntraView<Sample> view;
        const int area = 1 << 2 * log2;
        for (int i = 0; i < n; ++i)
        {
            const int32_t *j = jobs + 8 * i;
            view.bitDepth = bitDepth; view.log2 = log2; view.sdh = sdh; view.ctxIndex = ctx_index[i];
            view.src = (const Sample *)src + j[0]; view.ss = ss;
            view.nbU = (const Sample *)nb + j[1]; view.nbF = (const Sample *)nb + j[2];
            view.filt = (uint64_t)(uint32_t)j[3] | ((uint64_t)(uint32_t)j[4] << 32);
            view.edge = j[5];
            view.states = states; view.quant = quant; view.lambda = lambda;
            out[i] = decideIntraRd(view, order[i], ictx[i], rl);
            std::memcpy((Sample *)rec + size_t(i) * area, view.rec[out[i].index < 0 ? 0 : out[i].index], area * sizeof(Sample));
        }
    };
    if (S == 1) run(uint8_t(0));
    else run(uint16_t(0));
    return 0;
}

extern "C" int client_rqt(int S, int bitDepth, const void *src, intptr_t ss, const void *pred, intptr_t ps, void *rec, intptr_t rs, const uint8_t *states,
               const havoc_rqt_quant *quant, double lambda, double reciprocal_lambda, int sdh, const havoc_rqt_cu *cus, int n, havoc_rqt_result *out)
{
    if (!g_open) return -1;
#ifndef SEARCH_ORACLE
    if (!g_tu.ready)
    {
        havoc::populate_transform<8>(&g_tu.t8, g_code);
        havoc::populate_transform<10>(&g_tu.t10, g_code);
        havoc::populate_inverse_transform_add<uint8_t>(&g_tu.ita8, g_code, 1);
        havoc::populate_inverse_transform_add<uint16_t>(&g_tu.ita16, g_code, 1);
        havoc_populate_quantize_inverse(&g_tu.qi, g_code);
        havoc_populate_ssd<uint8_t>(&g_tu.ssd8, g_code);
        havoc_populate_ssd<uint16_t>(&g_tu.ssd16, g_code);
        g_tu.ready = true;
    }
#endif
    Lambda rl;
    rl.set(reciprocal_lambda);
    auto run = [&](auto tag) {
        typedef decltype(tag) Sample;
        static PerCallTuView<Sample> view;
        view.bitDepth = bitDepth; view.sdh = sdh;
        view.src = (const Sample *)src; view.ss = ss;
        view.pred = (const Sample *)pred; view.ps = ps;
        view.rec = (Sample *)rec; view.rs = rs;
        view.states = states; view.quant = quant; view.lambda = lambda;
        for (int i = 0; i < n; ++i)
        {
            view.ctxIndex = cus[i].ctx_index;
            view.kNext = 0;
            out[i] = decideRqt(view, cus[i], rl);
            view.commit(cus[i], out[i]);
        }
    };
    if (S == 1) run(uint8_t(0));
    else run(uint16_t(0));
    return 0;
}
#endif

extern "C" {

// the 35-mode intra stage: costs and refinement order from the per-mode SATDs (Search.hpp:40-190)
int client_intra_order(const havoc_search_intra_ctx *ctx, double reciprocal_sqrt_lambda, const int32_t *satd35, int n, havoc_search_intra_result *out)
{
    for (int i = 0; i < n; ++i)
    {
        IntraContext ic;
        for (int k = 0; k < 3; ++k) ic.candModeList[k] = ctx[i].cand_mode_list[k];
        ic.neighbourModes = ctx[i].neighbour_modes;
        ic.maxRefine = ctx[i].max_refine;
        ic.rateAminusC = ctx[i].rate_a_minus_c;
        ic.rateBminusC = ctx[i].rate_b_minus_c;
        const IntraResult r = intraModeOrder(ic, reciprocal_sqrt_lambda, satd35 + 35 * i);
        std::memset(&out[i], 0, sizeof(out[i]));
        for (int m = 0; m < 35; ++m) out[i].costs[m] = r.costs[m];
        for (int m = 0; m < r.count; ++m) out[i].order[m] = r.order[m];
        out[i].count = r.count;
    }
    return 0;
}

#ifdef HAVOC_CLASSIC_EXT
// the MI355X library's extension beside the reference API: tell it which host planes are pictures
int client_register(const void *origin, intptr_t stride, int width, int height, int pad, int S, int bit_depth, int role)
{
    return havoc_classic_register_picture(g_code, origin, stride, width, height, pad, S, bit_depth, role);
}
int client_unregister(const void *origin) { return havoc_classic_unregister_picture(g_code, origin); }
void client_stats(int64_t out[8]) { havoc_classic_stats(g_code, out); }
#endif

} // extern "C"
