// Intra prediction: planar, DC and the 33 angular modes for 4x4 .. 32x32 blocks
// (reference: havoc/pred_intra.cpp:20282-20401; neighbour layout havoc/pred_intra.cpp:43-51).
//
// Work mapping: a launch is uniform in block size (as the reference's table is indexed by log2TrafoSize,
// havoc/pred_intra.h:39-52); a job is owned by n*n/16 lanes (8x8: 4, 16x16: 16, 32x32: 64; 4x4 blocks: 4 lanes of 4
// samples), each producing 4x4 sub-blocks, so a wavefront carries 16 / 16 / 4 / 1 jobs.  The 4n+1 neighbour samples
// and the projected angular reference array live in LDS; a line of four angular samples is two packed 16-bit lerps.
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
    const int job = xcd_block(blockIdx.x, gridDim.x) * JPW + sub;
    const bool live = job < njobs;
    const int32_t *j = jobs + (live ? job : 0) * 8;   // havoc_mi355x_intra_job
    const int mode = j[3];
    const bool edge = j[4] != 0 && LOG2 < 5;
    const T *nbp = reinterpret_cast<const T *>(neighbours) + j[1];
    T *d = reinterpret_cast<T *>(dst) + j[0];
    uint16_t *nb = s_nb[sub];
    uint16_t *ref = s_ref[sub] + N;
    const int maxv = (1 << bitDepth) - 1;

    // 4N+1 neighbour samples, four per load (the array sits at an arbitrary sample offset: unaligned dword / qword loads)
    {
        const T *n0 = nbp - 2 * N - 1;
        for (int i = 4 * l; i < NBLEN; i += 4 * LANES)
        {
            if (i + 4 <= NBLEN)
            {
                if (S == 1)
                {
                    const uint32_t v = ld4(n0 + i);
                    nb[i] = v & 0xff; nb[i + 1] = (v >> 8) & 0xff; nb[i + 2] = (v >> 16) & 0xff; nb[i + 3] = v >> 24;
                }
                else
                {
                    const u32x2 v = ld8(n0 + i);
                    nb[i] = v.x & 0xffff; nb[i + 1] = v.x >> 16; nb[i + 2] = v.y & 0xffff; nb[i + 3] = v.y >> 16;
                }
            }
            else
                for (int k = i; k < NBLEN; ++k) nb[k] = n0[k];
        }
    }
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

    if constexpr (N >= 8 && SPL == 16)
    {
        // one 4x4 sub-block per lane and step.  Angular modes: along the minor axis the samples of one major index are
        // consecutive reference entries with one weight, so a line of four is two packed 16-bit lerps (pk_lerp) from one
        // unaligned LDS vector read; horizontal modes are produced as columns and transposed in registers.
        T *const d0 = d;
        for (int sb = l; sb < N * N / 16; sb += LANES)
        {
            const int by = (sb / (N / 4)) * 4, bx = (sb - (sb / (N / 4)) * (N / 4)) * 4;
            uint32_t rows[4][2];   // rows[i] = samples (bx .. bx+3, by + i) as two (lo, hi) 16-bit pairs
            if (mode >= 2)
            {
                const int angle = angle_of(mode);
                const bool vertical = mode >= 18;
                const int maj0 = vertical ? by : bx, min0 = vertical ? bx : by;
                const bool efilt = edge && min0 == 0 && angle == 0;
                uint32_t q[4][2];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                {
                    const int t = (maj0 + j + 1) * angle;
                    const int idx = t >> 5, fact = t & 31;
                    const uint16_t *r = ref + min0 + idx + 1;
                    uint32_t ra[2], rb[2];
                    ld_pairs<4>(r, ra);
                    rb[0] = __builtin_amdgcn_alignbit(ra[1], ra[0], 16);
                    rb[1] = __builtin_amdgcn_alignbit((uint32_t)r[4], ra[1], 16);
                    const uint32_t w1 = (uint32_t)fact * 0x00010001u, w0 = 0x00200020u - w1;
                    q[j][0] = pk_lerp(ra[0], rb[0], w0, w1);
                    q[j][1] = pk_lerp(ra[1], rb[1], w0, w1);
                    if (efilt)
                    {   // pred_intra.cpp:20355-20360 / :20394-20399: first column (row) of the pure vertical (horizontal) mode
                        const int side = vertical ? P_LEFT(maj0 + j) : P_TOP(maj0 + j);
                        q[j][0] = (q[j][0] & 0xffff0000u) | (uint32_t)clip3(0, maxv, (int)ref[1] + ((side - (int)ref[0]) >> 1));
                    }
                }
                if (vertical)
                {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { rows[i][0] = q[i][0]; rows[i][1] = q[i][1]; }
                }
                else
                {   // q[j] is column bx + j: 4x4 transpose of 16-bit values
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                    {
                        const uint32_t sel = (i & 1) ? 0x07060302u : 0x05040100u;
                        rows[i][0] = __builtin_amdgcn_perm(q[1][i >> 1], q[0][i >> 1], sel);
                        rows[i][1] = __builtin_amdgcn_perm(q[3][i >> 1], q[2][i >> 1], sel);
                    }
                }
            }
            else
            {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                {
                    const int y = by + i;
                    int v[4];
#pragma unroll
                    for (int o = 0; o < 4; ++o)
                    {
                        const int x = bx + o;
                        if (mode == 0)
                            v[o] = ((N - 1 - x) * P_LEFT(y) + (x + 1) * P_TOP(N) + (N - 1 - y) * P_TOP(x) + (y + 1) * P_LEFT(N) + N) >> (LOG2 + 1);
                        else
                        {
                            v[o] = dc;
                            if (edge)
                            {
                                if (x == 0 && y == 0) v[o] = (P_LEFT(0) + 2 * dc + P_TOP(0) + 2) >> 2;
                                else if (y == 0) v[o] = (P_TOP(x) + 3 * dc + 2) >> 2;
                                else if (x == 0) v[o] = (P_LEFT(y) + 3 * dc + 2) >> 2;
                            }
                        }
                    }
                    rows[i][0] = (uint32_t)v[0] | ((uint32_t)v[1] << 16);
                    rows[i][1] = (uint32_t)v[2] | ((uint32_t)v[3] << 16);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
            {
                T *o4 = d0 + (by + i) * stride_dst + bx;
                if (S == 1) st4(o4, __builtin_amdgcn_perm(rows[i][1], rows[i][0], 0x06040200u));   // low byte of each 16-bit sample
                else st8(o4, u32x2{rows[i][0], rows[i][1]});
            }
        }
        return;
    }

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
    case 3: LAUNCH(3, 16); break;    // 16
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
