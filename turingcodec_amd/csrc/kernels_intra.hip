// Intra prediction: planar, DC and the 33 angular modes for 4x4 .. 32x32 blocks
// (reference: havoc/pred_intra.cpp:20282-20401; neighbour layout havoc/pred_intra.cpp:43-51).
//
// Work mapping: a launch is uniform in block size (as the reference's table is indexed by log2TrafoSize,
// havoc/pred_intra.h:39-52); each job is owned by a group of LANES = min(64, n*n/4) lanes (4 samples each), so 4x4 blocks run 16 to a
// wavefront and 32x32 blocks give each lane 16 samples.  The 4n+1 neighbour samples and the projected angular
// reference array live in LDS; every predicted sample is a two-tap blend read from there.
#include "common.h"

namespace havoc_gpu {

// SPL = samples per lane: more samples per lane = more jobs per wavefront; the kernel is bound by the per-wavefront
// latency chain (job -> neighbours -> LDS -> reference -> stores), not by arithmetic
template <int S, int LOG2, int SPL>
__global__ __launch_bounds__(64) void k_intra(char *__restrict__ dst, long stride_dst, const char *__restrict__ neighbours,
                                              const int32_t *__restrict__ jobs, int njobs, int bitDepth)
{
    typedef typename Sample<S>::T T;
    constexpr int N = 1 << LOG2;
    constexpr int LANES = N * N / SPL < 1 ? 1 : (N * N / SPL < 64 ? N * N / SPL : 64);   // lanes per job
    constexpr int JPW = 64 / LANES;                  // jobs per wavefront
    constexpr int NBLEN = 4 * N + 1;
    // per job: nb[0 .. 4N]: index i <-> neighbours[i - 2N - 1]  (so nb[2N] = corner, nb[2N+1+x] = p(x,-1),
    //          nb[2N-1-y] = p(-1,y));  ref[-N .. 2N] stored at offset N
    __shared__ uint16_t s_nb[JPW][NBLEN + 3];
    __shared__ uint16_t s_ref[JPW][3 * N + 1 + 3];

    const int sub = threadIdx.x / LANES;
    const int l = threadIdx.x - sub * LANES;
    const int job = blockIdx.x * JPW + sub;
    const bool live = job < njobs;
    const int32_t *j = jobs + (live ? job : 0) * 8;   // havoc_mi355x_intra_job
    const int mode = j[3];
    const bool edge = j[4] != 0 && LOG2 < 5;
    const T *nbp = reinterpret_cast<const T *>(neighbours) + j[1];
    T *d = reinterpret_cast<T *>(dst) + j[0];
    uint16_t *nb = s_nb[sub];
    uint16_t *ref = s_ref[sub] + N;
    const int maxv = (1 << bitDepth) - 1;

    for (int i = l; i < NBLEN; i += LANES) nb[i] = nbp[i - 2 * N - 1];
    __syncthreads();
#define P_TOP(x) ((int)nb[2 * N + 1 + (x)])   /* p(x, -1), x = -1 .. 2N-1 */
#define P_LEFT(y) ((int)nb[2 * N - 1 - (y)])  /* p(-1, y), y = -1 .. 2N-1 */

    if (mode >= 2)
    {
        const int angle = angle_of(mode);
        const bool vertical = mode >= 18;
        // main reference: ref[i] = p(-1+i, -1) (vertical modes) or p(-1, -1+i) (horizontal modes), i = 0..N
        for (int i = l; i <= 2 * N; i += LANES)
            if (i <= N || angle >= 0) ref[i] = vertical ? P_TOP(i - 1) : P_LEFT(i - 1);
        if (angle < 0)
        {
            const int last = (N * angle) >> 5;
            const int inv = inv_angle_of(mode);
            for (int i = -1 - l; i >= last; i -= LANES)
                if (last < -1)
                {
                    const int k = -1 + ((i * inv + 128) >> 8);
                    ref[i] = vertical ? P_LEFT(k) : P_TOP(k);
                }
        }
    }
    int dc = 0;
    if (mode == 1)
    {
        // every lane of the group computes the same DC value (N <= 32 terms each side)
        int s = N;
        for (int i = 0; i < N; ++i) s += P_TOP(i) + P_LEFT(i);
        dc = s >> (LOG2 + 1);
    }
    __syncthreads();
    if (!live) return;

    for (int q = l; q < N * N / 4; q += LANES)
    {
        const int y = (4 * q) >> LOG2, x0 = (4 * q) & (N - 1);
        int v4[4];
#pragma unroll
        for (int o = 0; o < 4; ++o)
        {
            const int x = x0 + o;
            int v;
            if (mode == 0)
                v = ((N - 1 - x) * P_LEFT(y) + (x + 1) * P_TOP(N) + (N - 1 - y) * P_TOP(x) + (y + 1) * P_LEFT(N) + N) >> (LOG2 + 1);
            else if (mode == 1)
            {
                v = dc;
                if (edge)
                {
                    if (x == 0 && y == 0) v = (P_LEFT(0) + 2 * dc + P_TOP(0) + 2) >> 2;
                    else if (y == 0) v = (P_TOP(x) + 3 * dc + 2) >> 2;
                    else if (x == 0) v = (P_LEFT(y) + 3 * dc + 2) >> 2;
                }
            }
            else
            {
                const int angle = angle_of(mode);
                const bool vertical = mode >= 18;
                const int major = vertical ? y : x, minor = vertical ? x : y;
                const int t = (major + 1) * angle;
                const int idx = t >> 5, fact = t & 31;
                const int r0 = ref[minor + idx + 1];
                if (fact == 0) v = r0;
                else v = ((32 - fact) * r0 + fact * (int)ref[minor + idx + 2] + 16) >> 5;
                if (edge && mode == 26 && x == 0) v = clip3(0, maxv, P_TOP(0) + ((P_LEFT(y) - P_TOP(-1)) >> 1));
                if (edge && mode == 10 && y == 0) v = clip3(0, maxv, P_LEFT(0) + ((P_TOP(x) - P_TOP(-1)) >> 1));
            }
            v4[o] = v;
        }
        T *o4 = d + y * stride_dst + x0;
        if (S == 1) st4(o4, (uint32_t)v4[0] | ((uint32_t)v4[1] << 8) | ((uint32_t)v4[2] << 16) | ((uint32_t)v4[3] << 24));
        else st8(o4, u32x2{(uint32_t)v4[0] | ((uint32_t)v4[1] << 16), (uint32_t)v4[2] | ((uint32_t)v4[3] << 16)});
    }
#undef P_TOP
#undef P_LEFT
}

template <int S>
static hipError_t launch_intra_s(hipStream_t st, int log2, int bitDepth, void *dst, long sd, const void *nb, const void *jobs, int n)
{
    char *d = (char *)dst;
    const char *p = (const char *)nb;
    const int32_t *j = (const int32_t *)jobs;
    switch (log2)
    {
#define LAUNCH(l2, spl) hipLaunchKernelGGL((k_intra<S, l2, spl>), dim3((n + 64 / ((1 << (2 * l2)) / spl) - 1) / (64 / ((1 << (2 * l2)) / spl))), dim3(64), 0, st, d, sd, p, j, n, bitDepth)
    case 2: LAUNCH(2, 4); break;     // 16 jobs per wavefront
    case 3: LAUNCH(3, 8); break;     // 8
    case 4: LAUNCH(4, 16); break;    // 4
    case 5: LAUNCH(5, 16); break;    // 1
#undef LAUNCH
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_intra(hipStream_t st, int S, int log2, int bitDepth, void *dst, long sd, const void *nb, const void *jobs, int n)
{
    if (n <= 0) return hipSuccess;
    return S == 1 ? launch_intra_s<1>(st, log2, bitDepth, dst, sd, nb, jobs, n) : launch_intra_s<2>(st, log2, bitDepth, dst, sd, nb, jobs, n);
}

} // namespace havoc_gpu
