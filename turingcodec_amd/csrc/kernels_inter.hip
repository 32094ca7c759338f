// Inter prediction written to memory: HEVC 8-tap (luma) / 4-tap (chroma) fractional-sample interpolation, uni and bi
// (havoc/pred_inter.h:35,63; C reference havoc/pred_inter.cpp:76-202, 1207-1252), and the bi-search helper SubtractBi.
//
// Same two-phase structure as the sub-pel candidate kernel (kernels_subpel.hip):
//   1. horizontal pass straight from HBM/L2 with dot4 / dot2 (interp.h, hfilter4), transposed into LDS (tmp[x][y]);
//      bi-prediction does it for both references;
//   2. one column of 8 outputs per lane: two ds_read_b128 fetch the column's intermediates, the vertical filter is
//      four v_dot2_i32_i16 per output; uni: round, shift, clip; bi: the two 14-bit intermediates are averaged with
//      rounding and clipped (havoc_pred_bi_mean_c_ref).  The column goes to an LDS output tile;
//   3. the output tile is written row-wise, 4 samples per lane (coalesced, exactly w x h samples: the reference JIT's
//      licence to write to the right of the block, havoc/pred_inter.h:27, is not used).
// All phase combinations take the two-pass route (zero phase = the {..,64,..} filter), bit-identical to the
// reference's copy / one-pass forms for bit depths 8..10.  A launch is uniform in a size class (max_w x max_h), as
// the reference's table is indexed by width class: G = 8 / 32 / 128 / 256 lanes per block, 256 / G blocks per
// workgroup.
#include "common.h"
#include "interp.h"

namespace havoc_gpu {

// LDS of one workgroup of a size class: intermediates [JPW][NREF][MAXW * TH + 8] int16, then the output tiles [JPW][MAXH * OS + 2] uint16
template <int MAXW, int MAXH, int G, bool BI>
struct PredLds
{
    static constexpr int JPW = 256 / G, NREF = BI ? 2 : 1, TH = MAXH + 8, OS = MAXW + 2;
    static constexpr int tmpElems = MAXW * TH + 8, outElems = MAXH * OS + 2;
    static constexpr int tmpBytes = ((JPW * NREF * tmpElems * 2) + 15) & ~15;
    static constexpr int bytes = tmpBytes + ((JPW * outElems * 2 + 15) & ~15);
};

// one workgroup's share of a size class: workgroup `wg` of `wgs`, job table of `njobs` jobs of that class
template <int S, int TAPS, int MAXW, int MAXH, int G, bool BI>
__device__ __forceinline__ void pred_workgroup(char *lds, char *__restrict__ dst, long stride_dst, const char *__restrict__ ref, long stride_ref,
                                               const int32_t *__restrict__ jobs, int njobs, int bitDepth, int wg, int wgs)
{
    typedef typename Sample<S>::T T;
    typedef PredLds<MAXW, MAXH, G, BI> L;
    constexpr int JPW = 256 / G;
    constexpr int NREF = BI ? 2 : 1;
    constexpr int AB = TAPS / 2 - 1;
    constexpr int TH = MAXH + 8;     // intermediate column: h + TAPS - 1 <= MAXH + 7, rounded up to 8
    constexpr int OS = MAXW + 2;     // output tile row stride (samples), skewed against bank conflicts
    int16_t (*s_tmp)[NREF][L::tmpElems] = reinterpret_cast<int16_t (*)[NREF][L::tmpElems]>(lds);
    uint16_t (*s_out)[L::outElems] = reinterpret_cast<uint16_t (*)[L::outElems]>(lds + L::tmpBytes);

    const int sub = threadIdx.x / G, l = threadIdx.x - sub * G;
    const int job = xcd_block(wg, wgs) * JPW + sub;
    const bool live = job < njobs;
    // havoc_mi355x_pred_uni_job: dst, ref, w, h, xFrac, yFrac | havoc_mi355x_pred_bi_job: dst, ref0, ref1, w, h, 4 fracs
    const int32_t *j = jobs + (long)(live ? job : 0) * (BI ? 12 : 8);
    const int w = BI ? j[3] : j[2], h = BI ? j[4] : j[3];
    const long dsb = stride_dst * S, rsb = stride_ref * S;
    const int maxv = (1 << bitDepth) - 1;
    const int shift1 = min(4, bitDepth - 8);
    const int shift3 = max(2, 14 - bitDepth);
    const int qpr = (w + 3) >> 2, wh = h + TAPS - 1;
    const FastDiv fq(qpr);

    // ---- phase 1: horizontal pass(es)
#pragma unroll
    for (int r = 0; r < NREF; ++r)
    {
        const char *r0 = ref + (long)j[1 + r] * S;
        int cx[TAPS];
        taps_of<TAPS>(BI ? j[5 + 2 * r] : j[4], cx);
        int16_t *tmp = s_tmp[sub][r];
        for (int i = l; i < wh * qpr; i += G)
        {
            const int y = fq.div(i), x0 = (i - y * qpr) * 4;
            int a[4];
            hfilter4<S, TAPS>(r0 + (y - AB) * rsb + (x0 - AB) * S, cx, a);
#pragma unroll
            for (int o = 0; o < 4; ++o)
                if (x0 + o < w) tmp[(x0 + o) * TH + y] = (int16_t)(a[o] >> shift1);
        }
    }
    __syncthreads();

    // ---- phase 2: vertical pass, one column of 8 rows per lane
    {
        uint32_t cp[NREF][TAPS / 2];
#pragma unroll
        for (int r = 0; r < NREF; ++r)
        {
            int cy[TAPS];
            taps_of<TAPS>(BI ? j[6 + 2 * r] : j[5], cy);
#pragma unroll
            for (int k = 0; k < TAPS / 2; ++k) cp[r][k] = pack_i16(cy[2 * k], cy[2 * k + 1]);
        }
        const int groups = (h + 7) >> 3;
        const FastDiv fw(w);
        uint16_t *ot = s_out[sub];
        for (int it = l; it < (live ? w * groups : 0); it += G)
        {
            const int gy = fw.div(it), x = it - gy * w;
            const int y0 = gy * 8;
            int acc[8];
#pragma unroll
            for (int r = 0; r < NREF; ++r)
            {
                const int16_t *col = &s_tmp[sub][r][x * TH + y0];
                const u32x4 q0 = *reinterpret_cast<const u32x4 *>(col), q1 = *reinterpret_cast<const u32x4 *>(col + 8);
                const uint32_t e[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
                uint32_t od[7];
#pragma unroll
                for (int k = 0; k < 7; ++k) od[k] = __builtin_amdgcn_alignbit(e[k + 1], e[k], 16);
#pragma unroll
                for (int jj = 0; jj < 8; ++jj)
                {
                    int a = 0;
#pragma unroll
                    for (int k = 0; k < TAPS / 2; ++k) a = sdot2((jj & 1) ? od[(jj >> 1) + k] : e[(jj >> 1) + k], cp[r][k], a);
                    if (!BI) acc[jj] = a;
                    else if (r == 0) acc[jj] = a >> 6;
                    else acc[jj] += a >> 6;
                }
            }
#pragma unroll
            for (int jj = 0; jj < 8; ++jj)
            {
                const int v = BI ? (acc[jj] + (1 << shift3)) >> (shift3 + 1) : (acc[jj] + (1 << (5 + shift3))) >> (6 + shift3);
                if (y0 + jj < h) ot[(y0 + jj) * OS + x] = (uint16_t)clip3(0, maxv, v);
            }
        }
    }
    __syncthreads();

    // ---- phase 3: coalesced row-wise write of exactly w x h samples
    if (!live) return;
    char *d = dst + (long)j[0] * S;
    const uint16_t *ot = s_out[sub];
    for (int i = l; i < h * qpr; i += G)
    {
        const int y = fq.div(i), x0 = (i - y * qpr) * 4;
        const uint16_t *p = ot + y * OS + x0;
        char *q = d + y * dsb + x0 * S;
        if (x0 + 4 <= w)
        {
            if (S == 1) st4(q, (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24));
            else st8(q, u32x2{(uint32_t)p[0] | ((uint32_t)p[1] << 16), (uint32_t)p[2] | ((uint32_t)p[3] << 16)});
        }
        else
            for (int o = 0; x0 + o < w; ++o) reinterpret_cast<T *>(q)[o] = (T)p[o];
    }
}

template <int S, int TAPS, int MAXW, int MAXH, int G, bool BI>
__global__ __launch_bounds__(256) void k_pred(char *__restrict__ dst, long stride_dst, const char *__restrict__ ref, long stride_ref,
                                              const int32_t *__restrict__ jobs, int njobs, int bitDepth)
{
    __shared__ __attribute__((aligned(16))) char lds[PredLds<MAXW, MAXH, G, BI>::bytes];
    pred_workgroup<S, TAPS, MAXW, MAXH, G, BI>(lds, dst, stride_dst, ref, stride_ref, jobs, njobs, bitDepth, blockIdx.x, gridDim.x);
}

// ALL FOUR size classes of a job table in one launch (the width class stays a property of the job, the launch no longer is):
// the table is sorted by class -- count[c] jobs whose larger side is <= 8, 16, 32, 64 -- and a workgroup finds its class from the prefix of
// workgroups per class.  LDS is the largest class's (27 KB for bi-prediction), which still leaves 5 workgroups per CU.
struct PredClasses { int count[4], firstWg[5]; };

template <int S, int TAPS, bool BI>
__global__ __launch_bounds__(256) void k_pred_classes(char *__restrict__ dst, long stride_dst, const char *__restrict__ ref, long stride_ref,
                                                      const int32_t *__restrict__ jobs, PredClasses pc, int bitDepth)
{
    __shared__ __attribute__((aligned(16))) char lds[PredLds<64, 64, 256, BI>::bytes];
    constexpr int words = BI ? 12 : 8;
    const int wg = blockIdx.x;
    if (wg < pc.firstWg[1])
        pred_workgroup<S, TAPS, 8, 8, 8, BI>(lds, dst, stride_dst, ref, stride_ref, jobs, pc.count[0], bitDepth, wg, pc.firstWg[1]);
    else if (wg < pc.firstWg[2])
        pred_workgroup<S, TAPS, 16, 16, 32, BI>(lds, dst, stride_dst, ref, stride_ref, jobs + (long)pc.count[0] * words, pc.count[1], bitDepth, wg - pc.firstWg[1],
                                                pc.firstWg[2] - pc.firstWg[1]);
    else if (wg < pc.firstWg[3])
        pred_workgroup<S, TAPS, 32, 32, 128, BI>(lds, dst, stride_dst, ref, stride_ref, jobs + (long)(pc.count[0] + pc.count[1]) * words, pc.count[2], bitDepth,
                                                 wg - pc.firstWg[2], pc.firstWg[3] - pc.firstWg[2]);
    else
        pred_workgroup<S, TAPS, 64, 64, 256, BI>(lds, dst, stride_dst, ref, stride_ref, jobs + (long)(pc.count[0] + pc.count[1] + pc.count[2]) * words, pc.count[3],
                                                 bitDepth, wg - pc.firstWg[3], pc.firstWg[4] - pc.firstWg[3]);
}

// havoc::SubtractBi (havoc/pred_inter.h:87; havoc/pred_inter.cpp:2063-2080): dst = clip(2*src - pred)
template <int S>
__global__ __launch_bounds__(256) void k_subtract_bi(char *__restrict__ dst, long stride_dst, const char *__restrict__ pred, long stride_pred,
                                                     const char *__restrict__ src, long stride_src, const int32_t *__restrict__ jobs, int njobs,
                                                     int bitDepth)
{
    typedef typename Sample<S>::T T;
    const int job = xcd_block(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (job >= njobs) return;
    const int32_t *j = jobs + job * 8;   // havoc_mi355x_subtract_bi_job
    const int w = j[3], h = j[4];
    T *d = reinterpret_cast<T *>(dst) + j[0];
    const T *p = reinterpret_cast<const T *>(pred) + j[1];
    const T *s = reinterpret_cast<const T *>(src) + j[2];
    const int maxv = (1 << bitDepth) - 1;
    const FastDiv fd(w);
    for (int i = lane; i < w * h; i += kWave)
    {
        const int y = fd.div(i), x = i - y * w;
        d[y * stride_dst + x] = (T)clip3(0, maxv, 2 * (int)s[y * stride_src + x] - (int)p[y * stride_pred + x]);
    }
}

template <int S, int TAPS, bool BI>
static hipError_t launch_pred_st(hipStream_t st, int bd, int maxw, int maxh, char *dst, long sd, const char *ref, long sr, const int32_t *jobs, int n)
{
    const dim3 b(256);
    if (maxw <= 8 && maxh <= 8)
        hipLaunchKernelGGL((k_pred<S, TAPS, 8, 8, 8, BI>), dim3((n + 31) / 32), b, 0, st, dst, sd, ref, sr, jobs, n, bd);
    else if (maxw <= 16 && maxh <= 16)
        hipLaunchKernelGGL((k_pred<S, TAPS, 16, 16, 32, BI>), dim3((n + 7) / 8), b, 0, st, dst, sd, ref, sr, jobs, n, bd);
    else if (maxw <= 32 && maxh <= 32)
        hipLaunchKernelGGL((k_pred<S, TAPS, 32, 32, 128, BI>), dim3((n + 1) / 2), b, 0, st, dst, sd, ref, sr, jobs, n, bd);
    else
        hipLaunchKernelGGL((k_pred<S, TAPS, 64, 64, 256, BI>), dim3(n), b, 0, st, dst, sd, ref, sr, jobs, n, bd);
    return hipGetLastError();
}

template <bool BI>
static hipError_t launch_pred(hipStream_t st, int S, int taps, int bd, int maxw, int maxh, void *dst, long sd, const void *ref, long sr,
                              const void *jobs, int n)
{
    if (n <= 0) return hipSuccess;
    char *d = (char *)dst;
    const char *r = (const char *)ref;
    const int32_t *j = (const int32_t *)jobs;
    if (S == 1 && taps == 8) return launch_pred_st<1, 8, BI>(st, bd, maxw, maxh, d, sd, r, sr, j, n);
    if (S == 1 && taps == 4) return launch_pred_st<1, 4, BI>(st, bd, maxw, maxh, d, sd, r, sr, j, n);
    if (S == 2 && taps == 8) return launch_pred_st<2, 8, BI>(st, bd, maxw, maxh, d, sd, r, sr, j, n);
    if (S == 2 && taps == 4) return launch_pred_st<2, 4, BI>(st, bd, maxw, maxh, d, sd, r, sr, j, n);
    return hipErrorInvalidValue;
}

hipError_t launch_pred_uni(hipStream_t st, int S, int taps, int bd, int maxw, int maxh, void *dst, long sd, const void *ref, long sr,
                           const void *jobs, int n)
{
    return launch_pred<false>(st, S, taps, bd, maxw, maxh, dst, sd, ref, sr, jobs, n);
}

hipError_t launch_pred_bi(hipStream_t st, int S, int taps, int bd, int maxw, int maxh, void *dst, long sd, const void *ref, long sr, const void *jobs,
                          int n)
{
    return launch_pred<true>(st, S, taps, bd, maxw, maxh, dst, sd, ref, sr, jobs, n);
}

template <bool BI>
static hipError_t launch_pred_classes_t(hipStream_t st, int S, int taps, int bd, void *dst, long sd, const void *ref, long sr, const void *jobs, const int count[4])
{
    PredClasses pc;
    static const int jpw[4] = {32, 8, 2, 1};
    pc.firstWg[0] = 0;
    for (int c = 0; c < 4; ++c)
    {
        if (count[c] < 0) return hipErrorInvalidValue;
        pc.count[c] = count[c];
        pc.firstWg[c + 1] = pc.firstWg[c] + (count[c] + jpw[c] - 1) / jpw[c];
    }
    if (pc.firstWg[4] == 0) return hipSuccess;
    char *d = (char *)dst;
    const char *r = (const char *)ref;
    const int32_t *j = (const int32_t *)jobs;
    const dim3 g(pc.firstWg[4]), b(256);
    if (S == 1 && taps == 8) hipLaunchKernelGGL((k_pred_classes<1, 8, BI>), g, b, 0, st, d, sd, r, sr, j, pc, bd);
    else if (S == 1 && taps == 4) hipLaunchKernelGGL((k_pred_classes<1, 4, BI>), g, b, 0, st, d, sd, r, sr, j, pc, bd);
    else if (S == 2 && taps == 8) hipLaunchKernelGGL((k_pred_classes<2, 8, BI>), g, b, 0, st, d, sd, r, sr, j, pc, bd);
    else if (S == 2 && taps == 4) hipLaunchKernelGGL((k_pred_classes<2, 4, BI>), g, b, 0, st, d, sd, r, sr, j, pc, bd);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_pred_classes(hipStream_t st, int bi, int S, int taps, int bd, void *dst, long sd, const void *ref, long sr, const void *jobs, const int count[4])
{
    return bi ? launch_pred_classes_t<true>(st, S, taps, bd, dst, sd, ref, sr, jobs, count) : launch_pred_classes_t<false>(st, S, taps, bd, dst, sd, ref, sr, jobs, count);
}

hipError_t launch_subtract_bi(hipStream_t st, int S, int bitDepth, void *dst, long sd, const void *pred, long sp, const void *src, long ss,
                              const void *jobs, int n)
{
    if (n <= 0) return hipSuccess;
    if (S == 1)
        hipLaunchKernelGGL((k_subtract_bi<1>), dim3((n + 3) / 4), dim3(256), 0, st, (char *)dst, sd, (const char *)pred, sp, (const char *)src, ss,
                           (const int32_t *)jobs, n, bitDepth);
    else
        hipLaunchKernelGGL((k_subtract_bi<2>), dim3((n + 3) / 4), dim3(256), 0, st, (char *)dst, sd, (const char *)pred, sp, (const char *)src, ss,
                           (const int32_t *)jobs, n, bitDepth);
    return hipGetLastError();
}

} // namespace havoc_gpu
