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
