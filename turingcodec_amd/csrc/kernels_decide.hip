// Decisions that are data-parallel -- one lane per partition, no chain between partitions -- taken ON THE DEVICE between the launches they
// separate, so that neither the costs they read nor the job tables they produce cross the link (round 3):
//
//   k_intra_order   the 35 costs of a partition -> the order its modes are RD-refined in.  Restates intraModeOrder of
//                   turingcodec_amd/search/decision.hpp, i.e. turing/Search.hpp:55-98 (rate offsets of the most probable modes + lambda * SATD)
//                   and :143-190 (repeated selection of the cheapest mode, first index on ties; after `maxRefine` selections the most probable
//                   modes not yet taken are forced in).  Q16 costs in 64-bit integers, the host's arithmetic bit for bit.
//   k_intra_expand  every candidate (partition, mode of its order) -> the four job records of the chain that reconstructs it: intra prediction
//                   into a piece, residual + forward transform, Rdoq::runQuantisation (intra, scan by mode: turing/Global.h:1212-1227),
//                   reconstruction + SSD, level statistics.  Slots are handed out with an atomic counter (which slot a candidate gets does not
//                   matter to anything but addresses).
//   k_intra_decide  a partition's candidates' outcomes -> the champion (decideIntraRd of search/tu_decision.hpp; Search.hpp:143-255: the first
//                   candidate with the smallest mode rate + residual rate + ssd * reciprocal lambda) and the job that reconstructs it into the
//                   caller's buffer.
// The host forms of the same rules stay: they are what the per-call arms of the tests run over the reference's tables.
#include "common.h"
#include "../search/cand_mode_list.hpp"

namespace havoc_gpu {

namespace {

struct IntraCtx { int32_t cand[3], neighbourModes, maxRefine, reserved; int64_t rateA, rateB; };                    // havoc_search_intra_ctx
struct SearchJob { int32_t src_off, nb_off, nbf_off; uint32_t filt_lo, filt_hi; int32_t edge, reserved[2]; };       // havoc_mi355x_intra_search_job
struct IntraJob { int32_t dst_off, nb_off, log2, mode, edge, reserved[3]; };                                          // havoc_mi355x_intra_job
struct TuJob { int32_t coef_off, src_off, pred_off, rec_off; };                                                       // havoc_mi355x_tu_fused_job
struct RdoqJobRec { int32_t dst_off, src_off, quant_scale, quant_shift, inv_scale, lambda_q16, sdh_factor, ctx_index; uint8_t c_idx, scan_idx, is_intra, sdh; int32_t reserved[3]; };
struct RdResult { int32_t mode, index, evaluated, reserved; int64_t cost; int32_t cbf; uint32_t ssd; int32_t nonzero, sum_abs; };   // havoc_intra_rd_result
static_assert(sizeof(IntraCtx) == 40 && sizeof(SearchJob) == 32 && sizeof(IntraJob) == 32 && sizeof(TuJob) == 16 && sizeof(RdoqJobRec) == 48 && sizeof(RdResult) == 40,
              "record layouts");

constexpr int kMaxOrder = HAVOC_MI355X_INTRA_MAX_ORDER;      // maxRefine (<= 8) + 3 most probable modes, + 1 spare
constexpr int64_t kCostMax = 0x7fffffffffffffffll;

} // namespace

// order[i][0 .. count[i]) ; slot[i] = first candidate slot of partition i; total[0] = slots handed out, total[1] != 0: a partition wanted more
// than kMaxOrder candidates
__global__ __launch_bounds__(64) void k_intra_order(const int32_t *__restrict__ satd35, const IntraCtx *__restrict__ ictx, int n, int32_t lambdaQ16,
                                                    int32_t *__restrict__ order, int32_t *__restrict__ count, int32_t *__restrict__ slot, int32_t *__restrict__ total)
{
    __shared__ int64_t costs[35][64];      // [mode][lane]
    const int lane = threadIdx.x, i = blockIdx.x * 64 + lane;
    if (i >= n) return;
    IntraCtx c = ictx[i];
    // the records come from device memory: a mode outside 0..34 or more than three neighbour modes would index past costs[] / cand[] -- flagged
    // (total[1] bit 1) and brought into range, so nothing is written outside the partition's own slots
    const bool bad = (unsigned)c.cand[0] > 34u || (unsigned)c.cand[1] > 34u || (unsigned)c.cand[2] > 34u || (unsigned)c.neighbourModes > 3u || c.maxRefine < 1;
    if (bad)
    {
        atomicOr(total + 1, 2);
        for (int k = 0; k < 3; ++k) c.cand[k] = min(max(c.cand[k], 0), 34);
        c.neighbourModes = min(max(c.neighbourModes, 0), 3);
        c.maxRefine = max(c.maxRefine, 1);
    }
    for (int m = 0; m < 35; ++m) costs[m][lane] = 0;
    costs[c.cand[0]][lane] = c.rateA;
    costs[c.cand[1]][lane] = c.rateB;
    costs[c.cand[2]][lane] = c.rateB;
    for (int m = 0; m < 35; ++m) costs[m][lane] += (int64_t)lambdaQ16 * (int64_t)satd35[35 * (long)i + m];
    int cnt = 0, nMpm = 0;
    for (int j = 0; j < c.maxRefine + nMpm; ++j)
    {
        if (cnt == kMaxOrder)      // more candidates than a partition has slots: reported, nothing written past them
        {
            atomicOr(total + 1, 1);
            break;
        }
        int mode = 0;
        int64_t best = costs[0][lane];
        for (int m = 1; m < 35; ++m)
        {
            const int64_t v = costs[m][lane];
            if (v < best)
            {
                best = v;
                mode = m;
            }
        }
        costs[mode][lane] = kCostMax;
        if (j == c.maxRefine - 1)
            for (int k = 0; k < c.neighbourModes; ++k)
                if (costs[c.cand[k]][lane] != kCostMax)
                {
                    costs[c.cand[k]][lane] = 0;
                    ++nMpm;
                }
        order[kMaxOrder * (long)i + cnt++] = mode;
    }
    count[i] = cnt;
    slot[i] = atomicAdd(total, cnt);
}

__global__ __launch_bounds__(256) void k_intra_expand(const SearchJob *__restrict__ parts, const int32_t *__restrict__ order, const int32_t *__restrict__ count,
                                                      const int32_t *__restrict__ slot, const int32_t *__restrict__ ctxIndex, int n, int log2, int quantScale,
                                                      int quantShift, int invScale, int lambdaQ16, int sdhFactor, int sdh, IntraJob *__restrict__ ij,
                                                      TuJob *__restrict__ tj, RdoqJobRec *__restrict__ rj, int32_t *__restrict__ sj, int32_t *__restrict__ owner)
{
    const long t = blockIdx.x * 256L + threadIdx.x;
    const int i = (int)(t / kMaxOrder), k = (int)(t - (long)i * kMaxOrder);
    if (i >= n || k >= count[i]) return;
    const SearchJob p = parts[i];
    const int mode = order[kMaxOrder * (long)i + k], c = slot[i] + k, area = 1 << (2 * log2);
    const uint64_t mask = (uint64_t)p.filt_lo | ((uint64_t)p.filt_hi << 32);
    IntraJob a = {c * area, ((mask >> mode) & 1) ? p.nbf_off : p.nb_off, log2, mode, p.edge, {0, 0, 0}};
    ij[c] = a;
    TuJob b = {c * area, p.src_off, c * area, c * area};
    tj[c] = b;
    RdoqJobRec r = {c * area, c * area, quantScale, quantShift, invScale, lambdaQ16, sdhFactor, ctxIndex[i], 0,
                    (uint8_t)((log2 == 2 || log2 == 3) ? ((mode >= 6 && mode <= 14) ? 2 : ((mode >= 22 && mode <= 30) ? 1 : 0)) : 0), 1, (uint8_t)(sdh != 0), {0, 0, 0}};
    rj[c] = r;
    sj[2 * c] = c * area;
    sj[2 * c + 1] = area;
    owner[c] = i;
}

__global__ __launch_bounds__(256) void k_intra_decide(const IntraCtx *__restrict__ ictx, const int32_t *__restrict__ order, const int32_t *__restrict__ count,
                                                      const int32_t *__restrict__ slot, const int32_t *__restrict__ cbf, const uint32_t *__restrict__ ssd,
                                                      const int32_t *__restrict__ stats, const TuJob *__restrict__ tj, int n, int log2, int32_t reciprocalLambdaQ16,
                                                      RdResult *__restrict__ out, TuJob *__restrict__ fin)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const IntraCtx c = ictx[i];
    RdResult r = {-1, 0, 0, 0, kCostMax, 0, 0, 0, 0};
    const int base = slot[i];
    for (int j = 0; j < count[i]; ++j)
    {
        const int mode = order[kMaxOrder * (long)i + j], s = base + j;
        const int64_t modeRate = mode == c.cand[0] ? c.rateA : ((mode == c.cand[1] || mode == c.cand[2]) ? c.rateB : 0);
        const int64_t tuRate = (int64_t)(1 + (cbf[s] ? 2 * stats[2 * s] + stats[2 * s + 1] : 0)) << 16;      // search/tu_decision.hpp: tuRate (stand-in)
        const int64_t cost = modeRate + tuRate + (int64_t)reciprocalLambdaQ16 * (int64_t)(int32_t)ssd[s];
        ++r.evaluated;
        if (cost < r.cost)
        {
            r.mode = mode;
            r.index = j;
            r.cost = cost;
            r.cbf = cbf[s];
            r.ssd = ssd[s];
            r.nonzero = stats[2 * s];
            r.sum_abs = stats[2 * s + 1];
        }
    }
    out[i] = r;
    // a partition without candidates (count 0: max_refine < 1, which the host wrapper refuses) has no champion: out[i].mode stays -1 and its final job reads the
    // first slots of the buffers instead of a slot that belongs to another partition or lies past the allocation
    TuJob f = count[i] > 0 ? tj[base + max(0, r.index)] : TuJob{0, 0, 0, 0};
    f.rec_off = i << (2 * log2);
    fin[i] = f;
}

// ---- the transform-tree decision of inter units and the picture's block structure, ON THE DEVICE (round 4) -----------------------------------------------
// search/tu_decision.hpp's decideRqt (turing/Reconstruct.cpp:1296-1428) restated lane per unit over the outcomes of its five candidate blocks (depth 0: one block,
// depth 1: four, evaluated by the chain tu_forward -> rdoq -> tu_reconstruct -> level_stats), so that a picture's launches after its searches need no host in
// between: the decision, the job records that reconstruct EVERY candidate -- the chosen ones into the picture, the others into a dump area (a fixed number of jobs per
// size: the sequence can be recorded into a HIP graph) -- and the 4x4 cells havoc_mi355x_derive_bs reads.  The host forms stay: tests hold these against them.
struct RqtUnit { int32_t x0, y0, log2, ctxIndex; };                                                                          // havoc_rqt_cu
struct TuOutcome { int32_t cbf; uint32_t ssd; int32_t nonzero, sumAbs; };                                                    // havoc_tu_outcome
struct RqtResult { int32_t depth, triedZero; TuOutcome zero, one[4]; int64_t costZero, costOne; };                            // havoc_rqt_result
struct RqtSize { const int32_t *cbf; const uint32_t *ssd; const int32_t *stats; const TuJob *jobs; TuJob *fin; };             // one transform size: havoc_mi355x_rqt_size
struct RqtSizes { RqtSize s[4]; };
struct Cell { int16_t mv[2][2]; int8_t dpb[2]; uint8_t flags; int8_t qpY; uint8_t tuLog2; uint8_t reserved[3]; };            // havoc_mi355x_cell
static_assert(sizeof(RqtUnit) == 16 && sizeof(TuOutcome) == 16 && sizeof(RqtResult) == 104 && sizeof(RqtSize) == 40 && sizeof(Cell) == 16, "record layouts");

__device__ __forceinline__ TuOutcome outcomeOf(const RqtSize &z, int j) { return TuOutcome{z.cbf[j], z.ssd[j], z.stats[2 * j], z.stats[2 * j + 1]}; }
__device__ __forceinline__ int64_t tuRateOf(const TuOutcome &t) { return (int64_t)(1 + (t.cbf ? 2 * t.nonzero + t.sumAbs : 0)) << 16; }      // tu_decision.hpp: tuRate (stand-in)

__global__ __launch_bounds__(256) void k_rqt_decide(const RqtUnit *__restrict__ units, int n, const int32_t *__restrict__ zeroAt, const int32_t *__restrict__ oneAt,
                                                    const RqtSizes z, long recOrigin, int recStride, int dumpOff, int32_t reciprocalLambdaQ16, RqtResult *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const RqtUnit u = units[i];
    if (u.log2 < 3 || u.log2 > 5)      // a unit has a depth-0 block of 8 .. 32 and four depth-1 blocks of half that: anything else has no size table
    {
        RqtResult bad = RqtResult();
        bad.depth = -1;
        out[i] = bad;
        return;
    }
    const RqtSize &s0 = z.s[u.log2 - 2], &s1 = z.s[u.log2 - 3];
    const int j0 = zeroAt[i], j1 = oneAt[i], half = 1 << (u.log2 - 1);
    RqtResult r = RqtResult();
    int32_t ssdOne = 0;      // (int32 as the reference's StateEncodeSubstream::ssd: four 16x16 blocks of <= 255^2 * 256 each, or 10-bit SSDs >> 4, stay far below 2^31)
    bool coded = false;
    int64_t rateOne = 0;
    for (int k = 0; k < 4; ++k)      // rqtdepth = 1 first (Reconstruct.cpp:1325-1326), blocks in z-order
    {
        r.one[k] = outcomeOf(s1, j1 + k);
        ssdOne += (int32_t)r.one[k].ssd;
        coded |= r.one[k].cbf != 0;
        rateOne += tuRateOf(r.one[k]);
    }
    r.costOne = rateOne + (int64_t)reciprocalLambdaQ16 * (int64_t)ssdOne;
    if (coded)
    {
        r.triedZero = 1;
        r.zero = outcomeOf(s0, j0);
        r.costZero = tuRateOf(r.zero) + (int64_t)reciprocalLambdaQ16 * (int64_t)(int32_t)r.zero.ssd;
        r.depth = r.costZero < r.costOne ? 0 : 1;      // Reconstruct.cpp:1389
    }
    out[i] = r;
    // every candidate is reconstructed once more: the chosen tree into the picture -- a unit left without residual through its four depth-1 blocks, whose levels are
    // all zero (= the prediction) -- the rest into the dump area
    const bool zeroWins = r.depth == 0 && r.triedZero;
    TuJob f = s0.jobs[j0];
    f.rec_off = zeroWins ? (int32_t)(recOrigin + (long)u.y0 * recStride + u.x0) : dumpOff;
    s0.fin[j0] = f;
    for (int k = 0; k < 4; ++k)
    {
        TuJob g = s1.jobs[j1 + k];
        g.rec_off = zeroWins ? dumpOff : (int32_t)(recOrigin + (long)(u.y0 + (k >> 1) * half) * recStride + u.x0 + (k & 1) * half);
        s1.fin[j1 + k] = g;
    }
}

// havoc_search_block_cells on the device: every unit one inter 2Nx2N prediction unit from list 0 at the vector the field holds at its origin, its transform tree as decided
__global__ __launch_bounds__(256) void k_block_cells_blank(Cell *__restrict__ cells, int count, int qp, int dpb0)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    Cell c = Cell();
    c.dpb[0] = (int8_t)dpb0;
    c.dpb[1] = -1;
    c.qpY = (int8_t)qp;
    c.tuLog2 = 2;
    cells[i] = c;
}

__global__ __launch_bounds__(64) void k_block_cells(const RqtUnit *__restrict__ units, const RqtResult *__restrict__ dec, int n, const int32_t *__restrict__ field, int cw,
                                                    int qp, int dpb0, Cell *__restrict__ cells)
{
    const int i = blockIdx.x;
    if (i >= n) return;
    const RqtUnit u = units[i];
    const int x4 = u.x0 >> 2, y4 = u.y0 >> 2, n4 = (1 << u.log2) >> 2, half = n4 >> 1;
    const int32_t packed = field[(long)y4 * cw + x4];
    const RqtResult &d = dec[i];
    const bool split = d.depth == 1, coded0 = d.triedZero == 1 && d.zero.cbf != 0;
    for (int t = threadIdx.x; t < n4 * n4; t += 64)
    {
        const int y = t / n4, x = t - y * n4;
        Cell c = Cell();
        c.mv[0][0] = (int16_t)(packed & 0xffff);
        c.mv[0][1] = (int16_t)(packed >> 16);
        c.dpb[0] = (int8_t)dpb0;
        c.dpb[1] = -1;
        c.qpY = (int8_t)qp;
        const bool coded = split ? d.one[(y >= half) * 2 + (x >= half)].cbf != 0 : coded0;
        c.flags = (uint8_t)((coded ? 2 : 0) | (x == 0 ? 8 : 0) | (y == 0 ? 16 : 0));      // HAVOC_CELL_CODED, _PU_LEFT, _PU_TOP
        c.tuLog2 = (uint8_t)(u.log2 - (split ? 1 : 0));
        cells[(long)(y4 + y) * cw + x4 + x] = c;
    }
}

hipError_t launch_rqt_decide(hipStream_t st, const void *units, int n, const int32_t *zeroAt, const int32_t *oneAt, const void *sizes, long recOrigin, int recStride, int dumpOff,
                             int32_t rlQ16, void *out)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_rqt_decide, dim3((n + 255) / 256), dim3(256), 0, st, (const RqtUnit *)units, n, zeroAt, oneAt, *static_cast<const RqtSizes *>(sizes), recOrigin, recStride,
                       dumpOff, rlQ16, (RqtResult *)out);
    return hipGetLastError();
}

hipError_t launch_block_cells(hipStream_t st, int width, int height, int qp, int dpb0, const int16_t *field, const void *units, const void *dec, int n, void *cells, bool blank)
{
    const int cw = width >> 2, ch = height >> 2;
    if (blank)
        hipLaunchKernelGGL(k_block_cells_blank, dim3((cw * ch + 255) / 256), dim3(256), 0, st, (Cell *)cells, cw * ch, qp, dpb0);
    if (n > 0)
        hipLaunchKernelGGL(k_block_cells, dim3(n), dim3(64), 0, st, (const RqtUnit *)units, (const RqtResult *)dec, n, reinterpret_cast<const int32_t *>(field), cw, qp, dpb0, (Cell *)cells);
    return hipGetLastError();
}

// ---- an intra picture's partitions with their REAL dependencies (round 4; turing/Reconstruct.cpp:609-615, CandModeList.h:33-95) ----------------------------
// A partition predicts from the RECONSTRUCTION of what precedes it and takes its most probable modes from its neighbours' champions, so the partitions of
// a picture form a dependency graph; the host cuts it into levels (partitions whose neighbours are all final) and runs the batch chain level by level.
// What a level needs from the picture's running state is made here, on the device:
//   k_intra_gather   per partition: the 4n + 1 reference samples from the reconstruction picture with the substitution process of HEVC 8.4.4.2.2 for the
//                    ones not yet coded / outside the picture (availability = "the 4x4 cell's owner precedes me in coding order"), their filtered copy ([1 2 1], or the bi-linear
//                    strong smoothing of a flat 32x32 block's edges: IntraReferenceSamples.h:373-421), and candModeList from the modes decided left of and above it (CandModeList.h)
//   k_intra_commit   per partition: the champion's reconstruction into the picture, its mode into the mode map
struct ChainPart { int32_t x0, y0, log2, index; };                                                  // havoc_mi355x_intra_chain_part
struct ChainLayout { int32_t picWidth, picHeight, stride, pad, cellsPerRow, bitDepth, ctbLog2, strongIntraSmoothing; };      // havoc_mi355x_intra_chain_layout
static_assert(sizeof(ChainPart) == 16 && sizeof(ChainLayout) == 32, "record layouts");

template <int S>
__global__ __launch_bounds__(64) void k_intra_gather(const ChainLayout L, const char *__restrict__ recv, const int32_t *__restrict__ owner, const uint8_t *__restrict__ modes,
                                                     const ChainPart *__restrict__ parts, int n, const SearchJob *__restrict__ jobs, char *__restrict__ nbv,
                                                     IntraCtx *__restrict__ ictx)
{
    typedef typename Sample<S>::T T;
    __shared__ int32_t val[132];
    __shared__ uint8_t have[132];
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= n) return;
    const ChainPart p = parts[i];
    const SearchJob job = jobs[i];
    const T *rec = reinterpret_cast<const T *>(recv);
    T *nb = reinterpret_cast<T *>(nbv);
    const int nn = 1 << p.log2, len = 4 * nn + 1;
    auto available = [&](int x, int y) {
        return x >= 0 && y >= 0 && x < L.picWidth && y < L.picHeight && owner[(y >> 2) * L.cellsPerRow + (x >> 2)] < p.index;
    };
    for (int k = lane; k < len; k += 64)
    {   // k < 2n: the left column from the bottom; 2n: the corner; beyond: the row above
        const int x = k <= 2 * nn ? p.x0 - 1 : p.x0 + k - 2 * nn - 1;
        const int y = k < 2 * nn ? p.y0 + 2 * nn - 1 - k : p.y0 - 1;
        const bool a = available(x, y);
        have[k] = a;
        val[k] = a ? (int)rec[(long)(y + L.pad) * L.stride + x + L.pad] : 0;
    }
    __syncthreads();
    if (lane == 0)
    {   // 8.4.4.2.2: nothing there -> mid grey; else the first sample takes the first one there is, every other missing one its predecessor
        int first = 0;
        while (first < len && !have[first]) ++first;
        if (first == len)
            for (int k = 0; k < len; ++k) val[k] = 1 << (L.bitDepth - 1);
        else
        {
            if (first) val[0] = val[first];
            for (int k = 1; k < len; ++k)
                if (!have[k]) val[k] = val[k - 1];
        }
    }
    __syncthreads();
    const long base = job.nb_off - (2 * nn + 1), basef = job.nbf_off - (2 * nn + 1);
    // IntraReferenceSamples.h:382-402 (HEVC 8.4.4.2.3): a 32x32 block whose two edges are nearly linear takes the bi-linear interpolation between the corner
    // and the ends instead of the [1 2 1] filter (strong_intra_smoothing_enabled_flag, on by default: Encoder.cpp:688)
    const int corner = val[2 * nn], thr = 1 << (L.bitDepth - 5);
    const bool strong = L.strongIntraSmoothing == 1 && nn == 32 && abs(corner + val[4 * nn] - 2 * val[3 * nn]) < thr && abs(corner + val[0] - 2 * val[nn]) < thr;
    for (int k = lane; k < len; k += 64)
    {
        nb[base + k] = (T)val[k];
        int f;
        if (strong)      // k <= 64: the left column from the bottom (k = 63 - y), corner at 64; beyond: the row above (x = k - 65)
            f = k < 64 ? (k * corner + (64 - k) * val[0] + 32) >> 6 : k == 64 ? corner : ((128 - k) * corner + (k - 64) * val[128] + 32) >> 6;
        else
            f = (k == 0 || k == len - 1) ? val[k] : (val[k - 1] + 2 * val[k] + val[k + 1] + 2) >> 2;
        nb[basef + k] = (T)f;
    }
    if (lane == 0)
    {   // CandModeList.h:33-95: A = left, B = above (DC when not there, or above in another CTU row)
        const int a = available(p.x0 - 1, p.y0) ? modes[(p.y0 >> 2) * L.cellsPerRow + ((p.x0 - 1) >> 2)] : 1;
        const int b = available(p.x0, p.y0 - 1) && (p.y0 - 1) >= ((p.y0 >> L.ctbLog2) << L.ctbLog2) ? modes[((p.y0 - 1) >> 2) * L.cellsPerRow + (p.x0 >> 2)] : 1;
        IntraCtx c = ictx[i];
        int cand[3];
        c.neighbourModes = havoc_search::candModeListOf(a, b, cand);      // (search/cand_mode_list.hpp: pinned against the encoder's own lists)
        c.cand[0] = cand[0]; c.cand[1] = cand[1]; c.cand[2] = cand[2];
        ictx[i] = c;
    }
}

template <int S>
__global__ __launch_bounds__(256) void k_intra_commit(const ChainLayout L, char *__restrict__ recv, uint8_t *__restrict__ modes, const ChainPart *__restrict__ parts, int n,
                                                      const char *__restrict__ blocksv, const int32_t *__restrict__ choice, int choiceStride)
{
    typedef typename Sample<S>::T T;
    const int i = blockIdx.x;
    if (i >= n) return;
    const ChainPart p = parts[i];
    const int nn = 1 << p.log2, area = nn * nn;
    const T *blk = reinterpret_cast<const T *>(blocksv) + (long)i * area;
    T *rec = reinterpret_cast<T *>(recv);
    for (int t = threadIdx.x; t < area; t += 256)
    {
        const int y = t >> p.log2, x = t & (nn - 1);
        rec[(long)(p.y0 + y + L.pad) * L.stride + p.x0 + x + L.pad] = blk[t];
    }
    const int cw = nn >> 2, mode = choice[(long)i * choiceStride];      // a plain array of modes (stride 1) or the choice records (stride 10: `mode` leads)
    for (int t = threadIdx.x; t < cw * cw; t += 256)
        modes[((p.y0 >> 2) + t / cw) * L.cellsPerRow + (p.x0 >> 2) + t % cw] = (uint8_t)mode;
}

// the candidate slots [total[0], capacity) nobody was given: copies of slot 0's records that write into their OWN slots, so that the chain can be launched over
// `capacity` jobs without the host first learning how many candidates there are (an intra picture's levels: thousands of small batches, no wait between them)
__global__ __launch_bounds__(256) void k_intra_fill_spare(const int32_t *__restrict__ total, int capacity, int log2, IntraJob *__restrict__ ij, TuJob *__restrict__ tj,
                                                          RdoqJobRec *__restrict__ rj, int32_t *__restrict__ sj, int32_t *__restrict__ owner)
{
    const int c = total[0] + blockIdx.x * 256 + threadIdx.x;
    if (c >= capacity || total[0] <= 0) return;
    const int area = 1 << (2 * log2);
    IntraJob a = ij[0];
    a.dst_off = c * area;
    ij[c] = a;
    const TuJob b = {c * area, tj[0].src_off, c * area, c * area};
    tj[c] = b;
    RdoqJobRec r = rj[0];
    r.dst_off = r.src_off = c * area;
    rj[c] = r;
    sj[2 * c] = c * area;
    sj[2 * c + 1] = area;
    owner[c] = owner[0];
}

hipError_t launch_intra_fill_spare(hipStream_t st, const int32_t *total, int capacity, int log2, void *ij, void *tj, void *rj, int32_t *sj, int32_t *owner)
{
    if (capacity <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_intra_fill_spare, dim3((capacity + 255) / 256), dim3(256), 0, st, total, capacity, log2, (IntraJob *)ij, (TuJob *)tj, (RdoqJobRec *)rj, sj, owner);
    return hipGetLastError();
}

hipError_t launch_intra_gather(hipStream_t st, int S, const void *layout, const void *rec, const int32_t *owner, const uint8_t *modes, const void *parts, int n, const void *jobs,
                               void *nb, void *ictx)
{
    if (n <= 0) return hipSuccess;
    const ChainLayout L = *static_cast<const ChainLayout *>(layout);
    if (S == 1) hipLaunchKernelGGL((k_intra_gather<1>), dim3(n), dim3(64), 0, st, L, (const char *)rec, owner, modes, (const ChainPart *)parts, n, (const SearchJob *)jobs, (char *)nb, (IntraCtx *)ictx);
    else hipLaunchKernelGGL((k_intra_gather<2>), dim3(n), dim3(64), 0, st, L, (const char *)rec, owner, modes, (const ChainPart *)parts, n, (const SearchJob *)jobs, (char *)nb, (IntraCtx *)ictx);
    return hipGetLastError();
}

hipError_t launch_intra_commit(hipStream_t st, int S, const void *layout, void *rec, uint8_t *modes, const void *parts, int n, const void *blocks, const void *choice, int choiceStride)
{
    if (n <= 0) return hipSuccess;
    const ChainLayout L = *static_cast<const ChainLayout *>(layout);
    if (S == 1) hipLaunchKernelGGL((k_intra_commit<1>), dim3(n), dim3(256), 0, st, L, (char *)rec, modes, (const ChainPart *)parts, n, (const char *)blocks, (const int32_t *)choice, choiceStride);
    else hipLaunchKernelGGL((k_intra_commit<2>), dim3(n), dim3(256), 0, st, L, (char *)rec, modes, (const ChainPart *)parts, n, (const char *)blocks, (const int32_t *)choice, choiceStride);
    return hipGetLastError();
}

// ---- job tables made ON THE DEVICE from the decided motion field (round 4): what a host would otherwise build per picture and upload -----------------
// The prediction jobs of the steps that follow the searches -- every unit at its decided vector (luma 8-tap, chroma 4-tap), every unit's spatial merge
// candidates bi-directionally in three planes -- depend on the field the search kernel has just left in device memory; building them there keeps the
// picture's step free of host work between its launches.  Layout of the pictures: planes of one component at multiples of `planeElems` (luma: source, list 0,
// list 1; chroma: Cb source, list 0, list 1, then Cr source, list 0, list 1), `pad` samples of border, `stride` samples per row.
struct MergeLayout { int32_t picWidth, picHeight, range, fieldCw, lumaStride, lumaPad, lumaPlaneElems, chromaStride, chromaPad, chromaPlaneElems, reserved[2]; };   // havoc_mi355x_field_layout
struct PredUniJob { int32_t dst_off, ref_off, w, h, xFrac, yFrac, reserved[2]; };
struct PredBiJob { int32_t dst_off, ref0_off, ref1_off, w, h, xFrac0, yFrac0, xFrac1, yFrac1, reserved[3]; };
static_assert(sizeof(MergeLayout) == 48 && sizeof(PredUniJob) == 32 && sizeof(PredBiJob) == 48, "record layouts");

__device__ __forceinline__ void clampVector(const MergeLayout &L, int x0, int y0, int n, int &mx, int &my)
{   // as LimitFullPelMv (Search.hpp:1366-1407) keeps the searches: the block stays within `range` samples of the picture (quarter-sample units)
    mx = min(max(mx, (-L.range - x0) * 4), (L.picWidth + L.range - x0 - n) * 4);
    my = min(max(my, (-L.range - y0) * 4), (L.picHeight + L.range - y0 - n) * 4);
}

// thread = (unit, candidate k of 5): the vectors of both lists at spatial merge position k (HEVC 8.5.3.2.3: A1, B1, B0, A0, B2; outside the picture: zero)
__global__ __launch_bounds__(256) void k_merge_jobs(const MergeLayout L, const int32_t *__restrict__ field, const int32_t *__restrict__ ux, const int32_t *__restrict__ uy, int n,
                                                    int log2, PredBiJob *__restrict__ jl, PredBiJob *__restrict__ jcb, PredBiJob *__restrict__ jcr, int32_t *__restrict__ vectors)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= 5 * n) return;
    const int i = t / 5, k = t - 5 * i, nn = 1 << log2, x0 = ux[i], y0 = uy[i];
    const int px = k == 0 || k >= 3 ? x0 - 1 : (k == 1 ? x0 + nn - 1 : x0 + nn);
    const int py = k == 0 ? y0 + nn - 1 : (k == 3 ? y0 + nn : y0 - 1);
    const bool inside = px >= 0 && py >= 0 && px < L.picWidth && py < L.picHeight;
    const int ch = (L.picHeight + 3) >> 2;
    int mv[2][2];
    for (int lst = 0; lst < 2; ++lst)
    {
        const int32_t packed = inside ? field[((long)lst * ch + (py >> 2)) * L.fieldCw + (px >> 2)] : 0;
        mv[lst][0] = (int16_t)(packed & 0xffff);
        mv[lst][1] = (int16_t)(packed >> 16);
        clampVector(L, x0, y0, nn, mv[lst][0], mv[lst][1]);
        vectors[2 * t + lst] = (mv[lst][0] & 0xffff) | (mv[lst][1] << 16);
    }
    {
        PredBiJob j = PredBiJob();
        j.dst_off = t * nn * nn;
        j.w = j.h = nn;
        const int here = (y0 + L.lumaPad) * L.lumaStride + x0 + L.lumaPad;
        j.ref0_off = 1 * L.lumaPlaneElems + here + (mv[0][1] >> 2) * L.lumaStride + (mv[0][0] >> 2);
        j.ref1_off = 2 * L.lumaPlaneElems + here + (mv[1][1] >> 2) * L.lumaStride + (mv[1][0] >> 2);
        j.xFrac0 = mv[0][0] & 3; j.yFrac0 = mv[0][1] & 3; j.xFrac1 = mv[1][0] & 3; j.yFrac1 = mv[1][1] & 3;
        jl[t] = j;
    }
    for (int comp = 0; comp < 2; ++comp)
    {
        PredBiJob j = PredBiJob();
        const int cn = nn >> 1;
        j.dst_off = t * cn * cn;
        j.w = j.h = cn;
        const int here = ((y0 >> 1) + L.chromaPad) * L.chromaStride + (x0 >> 1) + L.chromaPad;
        j.ref0_off = (3 * comp + 1) * L.chromaPlaneElems + here + (mv[0][1] >> 3) * L.chromaStride + (mv[0][0] >> 3);
        j.ref1_off = (3 * comp + 2) * L.chromaPlaneElems + here + (mv[1][1] >> 3) * L.chromaStride + (mv[1][0] >> 3);
        j.xFrac0 = mv[0][0] & 7; j.yFrac0 = mv[0][1] & 7; j.xFrac1 = mv[1][0] & 7; j.yFrac1 = mv[1][1] & 7;
        (comp ? jcr : jcb)[t] = j;
    }
}

// thread = unit: its HavocPredUni job at the vector decided for list `list` at its origin; plane 0 = luma (8-tap phases), 1 / 2 = Cb / Cr (eighth-sample phases)
__global__ __launch_bounds__(256) void k_pred_jobs(const MergeLayout L, const int32_t *__restrict__ field, int list, const int32_t *__restrict__ ux, const int32_t *__restrict__ uy,
                                                   int n, int log2, int plane, const int32_t *__restrict__ dstOff, PredUniJob *__restrict__ jobs)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int x0 = ux[i], y0 = uy[i], ch = (L.picHeight + 3) >> 2;
    const int32_t packed = field[((long)list * ch + (y0 >> 2)) * L.fieldCw + (x0 >> 2)];
    const int mx = (int16_t)(packed & 0xffff), my = (int16_t)(packed >> 16);
    PredUniJob j = PredUniJob();
    j.dst_off = dstOff[i];
    if (plane == 0)
    {
        j.w = j.h = 1 << log2;
        j.ref_off = (1 + list) * L.lumaPlaneElems + (y0 + (my >> 2) + L.lumaPad) * L.lumaStride + x0 + (mx >> 2) + L.lumaPad;
        j.xFrac = mx & 3; j.yFrac = my & 3;
    }
    else
    {
        j.w = j.h = 1 << (log2 - 1);
        j.ref_off = (3 * (plane - 1) + 1 + list) * L.chromaPlaneElems + ((y0 >> 1) + (my >> 3) + L.chromaPad) * L.chromaStride + (x0 >> 1) + (mx >> 3) + L.chromaPad;
        j.xFrac = mx & 7; j.yFrac = my & 7;
    }
    jobs[i] = j;
}

// thread = unit: cost of its five candidates = stand-in rate of the merge index (k + 1 bits, 4 at most) + (satdY + satdCb + satdCr) * reciprocalSqrtLambda (Q16: measurePuCost,
// Search.hpp:1659-1706), best = the first of the cheapest (`cost < bestCost`, searchMergeModes :1754-1768)
__global__ __launch_bounds__(256) void k_merge_decide(const int32_t *__restrict__ sy, const int32_t *__restrict__ scb, const int32_t *__restrict__ scr, int n, int64_t lamQ16,
                                                      int64_t *__restrict__ cost, int32_t *__restrict__ best)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int64_t bestCost = 0;
    int b = 0;
    for (int k = 0; k < 5; ++k)
    {
        const int t = 5 * i + k;
        const int64_t c = ((int64_t)min(k + 1, 4) << 16) + ((int64_t)sy[t] + scb[t] + scr[t]) * lamQ16;
        cost[t] = c;
        if (k == 0 || c < bestCost) { bestCost = c; b = k; }
    }
    best[i] = b;
}

hipError_t launch_merge_decide(hipStream_t st, const int32_t *sy, const int32_t *scb, const int32_t *scr, int n, int64_t lamQ16, int64_t *cost, int32_t *best)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_merge_decide, dim3((n + 255) / 256), dim3(256), 0, st, sy, scb, scr, n, lamQ16, cost, best);
    return hipGetLastError();
}

hipError_t launch_merge_jobs(hipStream_t st, const void *layout, const int16_t *field, const int32_t *ux, const int32_t *uy, int n, int log2, void *jl, void *jcb, void *jcr,
                             int16_t *vectors)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_merge_jobs, dim3((5 * n + 255) / 256), dim3(256), 0, st, *static_cast<const MergeLayout *>(layout), reinterpret_cast<const int32_t *>(field), ux, uy, n, log2,
                       (PredBiJob *)jl, (PredBiJob *)jcb, (PredBiJob *)jcr, reinterpret_cast<int32_t *>(vectors));
    return hipGetLastError();
}

hipError_t launch_pred_jobs(hipStream_t st, const void *layout, const int16_t *field, int list, const int32_t *ux, const int32_t *uy, int n, int log2, int plane,
                            const int32_t *dstOff, void *jobs)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_pred_jobs, dim3((n + 255) / 256), dim3(256), 0, st, *static_cast<const MergeLayout *>(layout), reinterpret_cast<const int32_t *>(field), list, ux, uy, n, log2,
                       plane, dstOff, (PredUniJob *)jobs);
    return hipGetLastError();
}

hipError_t launch_intra_order(hipStream_t st, const int32_t *satd35, const void *ictx, int n, int32_t lambdaQ16, int32_t *order, int32_t *count, int32_t *slot,
                              int32_t *total)
{
    if (n <= 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(total, 0, 8, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_intra_order, dim3((n + 63) / 64), dim3(64), 0, st, satd35, (const IntraCtx *)ictx, n, lambdaQ16, order, count, slot, total);
    return hipGetLastError();
}

hipError_t launch_intra_expand(hipStream_t st, const void *parts, const int32_t *order, const int32_t *count, const int32_t *slot, const int32_t *ctxIndex, int n, int log2,
                               int quantScale, int quantShift, int invScale, int lambdaQ16, int sdhFactor, int sdh, void *ij, void *tj, void *rj, int32_t *sj, int32_t *owner)
{
    if (n <= 0) return hipSuccess;
    const long threads = (long)n * kMaxOrder;
    hipLaunchKernelGGL(k_intra_expand, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, (const SearchJob *)parts, order, count, slot, ctxIndex, n, log2, quantScale,
                       quantShift, invScale, lambdaQ16, sdhFactor, sdh, (IntraJob *)ij, (TuJob *)tj, (RdoqJobRec *)rj, sj, owner);
    return hipGetLastError();
}

hipError_t launch_intra_decide(hipStream_t st, const void *ictx, const int32_t *order, const int32_t *count, const int32_t *slot, const int32_t *cbf, const uint32_t *ssd,
                               const int32_t *stats, const void *tj, int n, int log2, int32_t reciprocalLambdaQ16, void *out, void *fin)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_intra_decide, dim3((n + 255) / 256), dim3(256), 0, st, (const IntraCtx *)ictx, order, count, slot, cbf, ssd, stats, (const TuJob *)tj, n, log2,
                       reciprocalLambdaQ16, (RdResult *)out, (TuJob *)fin);
    return hipGetLastError();
}

} // namespace havoc_gpu
