// Fused search kernels: the batches the encoder's decision loops issue as (predict, measure) PAIRS, evaluated
// without ever writing the prediction to HBM.
//
//   k_intra_satd35 : the 35-mode luma SATD stage of searchIntraPartition (turing/Search.hpp:113-142 calling
//                    PredictIntraLumaBlock, turing/Reconstruct.cpp:630-701): for one partition, all 35 intra predictions
//                    (havoc/pred_intra.cpp:20282-20401) each followed by the Hadamard SATD against the source block
//                    (8x8 tiles, or one 4x4 for 4x4 blocks; havoc/hadamard.cpp:58-98).
//
// Mapping: one wavefront owns P partitions of one size; source block, both neighbour arrays (unfiltered / filtered)
// and the 33 projected angular reference arrays live in LDS.  A work item is (partition, mode, SATD tile): one LANE
// predicts its tile straight into registers, subtracts the source tile, runs the 2-D Hadamard in registers and adds
// its tile cost into the partition's 35 LDS accumulators.  Horizontal modes are evaluated in the transposed domain
// (prediction^T against source^T): SATD is invariant under transposition, and it makes the horizontal and vertical
// modes one code path.  Items are ordered mode-major so that after the first step every wavefront step is
// divergence-free angular work.
#include "common.h"

namespace havoc_gpu {

__constant__ int8_t c_angle35[35] = {0,  0,   32,  26,  21,  17,  13, 9,  5,  2,  0, -2, -5, -9, -13, -17, -21, -26,
                                     -32, -26, -21, -17, -13, -9, -5, -2, 0,  2,  5, 9,  13, 17,  21,  26,  32};
__constant__ int16_t c_invAngle35[26] = {0,    0,    0,    0,    0,    0,    0,    0,    0,    0,    0,     -4096, -1638,
                                         -910, -630, -482, -390, -315, -256, -315, -390, -482, -630, -910, -1638, -4096};

// normalised SATD of a TS x TS difference tile held in registers (compute_satd_c_ref<TS>)
template <int S, int TS>
__device__ __forceinline__ int satd_regs(int (&d)[TS][TS])
{
#pragma unroll
    for (int y = 0; y < TS; ++y) wht_inplace<TS>(d[y]);
    int sum = TS / 4;
#pragma unroll
    for (int x = 0; x < TS; ++x)
    {
        int col[TS];
#pragma unroll
        for (int y = 0; y < TS; ++y) col[y] = d[y][x];
        wht_inplace<TS>(col);
#pragma unroll
        for (int y = 0; y < TS; ++y) sum += abs(col[y]);
    }
    sum /= TS / 2;
    return S == 2 ? sum >> 2 : sum;
}

// 8-bit content: the same transform on packed 16-bit pairs (|coefficient| <= 64*255 fits int16): v_pk_add/sub_i16 for
// the butterflies between registers, rotate + v_pk_mad for the butterfly inside a register, v_sad_u16 against a bias
// for the sum of absolute values.  Roughly half the instructions of the 32-bit version.
template <int TS>
__device__ __forceinline__ int satd_regs_pk(uint32_t (&p)[TS][TS / 2])
{
#pragma unroll
    for (int y = 0; y < TS; ++y)
    {
#pragma unroll
        for (int k = 0; k < TS / 2; ++k) p[y][k] = pk_bfly(p[y][k]);          // x pairs (0,1), (2,3), ...
#pragma unroll
        for (int len = 1; len < TS / 2; len <<= 1)                            // between registers of the row
#pragma unroll
            for (int i = 0; i < TS / 2; i += len << 1)
#pragma unroll
                for (int k = i; k < i + len; ++k)
                {
                    const uint32_t a = p[y][k], b = p[y][k + len];
                    p[y][k] = pk_add(a, b);
                    p[y][k + len] = pk_sub(a, b);
                }
    }
#pragma unroll
    for (int len = 1; len < TS; len <<= 1)                                    // between rows
#pragma unroll
        for (int i = 0; i < TS; i += len << 1)
#pragma unroll
            for (int y = i; y < i + len; ++y)
#pragma unroll
                for (int k = 0; k < TS / 2; ++k)
                {
                    const uint32_t a = p[y][k], b = p[y + len][k];
                    p[y][k] = pk_add(a, b);
                    p[y + len][k] = pk_sub(a, b);
                }
    uint32_t sum = TS / 4;
#pragma unroll
    for (int y = 0; y < TS; ++y)
#pragma unroll
        for (int k = 0; k < TS / 2; ++k) sum = pk_abs_acc(p[y][k], sum);
    return (int)(sum / (TS / 2));
}

// the difference tile of one work item: packed pairs for 8-bit samples, 32-bit for 16-bit samples
template <int S, int TS>
struct TileDiff
{
    int d[S == 1 ? 1 : TS][S == 1 ? 1 : TS];
    uint32_t p[S == 1 ? TS : 1][S == 1 ? TS / 2 : 1];
    // row j = source row (TS samples at `src`, 4-byte aligned) minus prediction v[]
    __device__ __forceinline__ void set_row(int j, const uint16_t *src, const int (&v)[TS])
    {
        if constexpr (S == 1)
        {
#pragma unroll
            for (int k = 0; k < TS / 2; ++k)
                p[j][k] = pk_sub(*reinterpret_cast<const uint32_t *>(src + 2 * k), (uint32_t)v[2 * k] | ((uint32_t)v[2 * k + 1] << 16));
        }
        else
        {
#pragma unroll
            for (int i = 0; i < TS; ++i) d[j][i] = (int)src[i] - v[i];
        }
    }
    __device__ __forceinline__ int satd()
    {
        if constexpr (S == 1) return satd_regs_pk<TS>(p);
        else return satd_regs<S, TS>(d);
    }
};

// job: havoc_mi355x_intra_search_job = { src_off, nb_off, nbf_off, filt_lo, filt_hi, edge, reserved[2] }
template <int S, int LOG2, int P, int THREADS>
__global__ __launch_bounds__(THREADS) void k_intra_satd35(const char *__restrict__ src, long stride_src, const char *__restrict__ neighbours,
                                                     const int32_t *__restrict__ jobs, int njobs, int bitDepth, int32_t *__restrict__ cost)
{
    typedef typename Sample<S>::T T;
    constexpr int N = 1 << LOG2;
    constexpr int TS = N >= 8 ? 8 : 4;       // SATD tile (Reconstruct.cpp:684-701)
    constexpr int TPR = N / TS, NT = TPR * TPR;
    constexpr int NB = 4 * N + 1;
    constexpr int RL = 3 * N + 2;            // projected reference: indices -N .. 2N (+1 slack)
    constexpr int PT = P * NT;               // (partition, tile) pairs per workgroup

    // row stride N+2 and a per-partition skew keep tiles of different rows / partitions on different LDS banks
    // (with dense N*N blocks every partition and every 8-row band started on bank 0: 10-way conflicts on 16x16)
    constexpr int NS = N + 2;
    constexpr int SB = N * NS + 6;
    __shared__ uint16_t s_src[P][SB];        // row-major source block
    __shared__ uint16_t s_srcT[P][SB];       // transposed source block
    __shared__ uint16_t s_nb[P][2][NB + 1];  // [0] unfiltered, [1] filtered; index i <-> neighbours[i - 2N - 1]
    __shared__ uint16_t s_ref[P][33][RL];    // angular modes 2..34; ref[i] at [i + N]
    __shared__ int s_dc[P][2];
    __shared__ int s_cost[P][36];
    __shared__ int s_job[P][8];

    const int lane = threadIdx.x;
    const int job0 = blockIdx.x * P;
    const int maxv = (1 << bitDepth) - 1;

    for (int i = lane; i < P * 8; i += THREADS)
    {
        const int p = i >> 3;
        s_job[p][i & 7] = jobs[(long)min(job0 + p, njobs - 1) * 8 + (i & 7)];
    }
    for (int i = lane; i < P * 36; i += THREADS) s_cost[i / 36][i % 36] = 0;
    __syncthreads();

    // ---- phase 0: source block (both orientations) and the two neighbour arrays into LDS
    const long ssb = stride_src * S;
    for (int i = lane; i < P * N * N / 4; i += THREADS)
    {
        const int p = i / (N * N / 4), r = i - p * (N * N / 4);
        const int y = r / (N / 4), x = (r - y * (N / 4)) * 4;
        const char *q = src + (long)s_job[p][0] * S + y * ssb + x * S;
        uint16_t v[4];
        if (S == 1)
        {
            const uint32_t w = ld4(q);
            v[0] = w & 0xff; v[1] = (w >> 8) & 0xff; v[2] = (w >> 16) & 0xff; v[3] = w >> 24;
        }
        else
        {
            const u32x2 w = ld8(q);
            v[0] = w.x & 0xffff; v[1] = w.x >> 16; v[2] = w.y & 0xffff; v[3] = w.y >> 16;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
        {
            s_src[p][y * NS + x + k] = v[k];
            s_srcT[p][(x + k) * NS + y] = v[k];
        }
    }
    for (int i = lane; i < P * 2 * NB; i += THREADS)
    {
        const int p = i / (2 * NB), r = i - p * 2 * NB;
        const int f = r / NB, k = r - f * NB;
        s_nb[p][f][k] = reinterpret_cast<const T *>(neighbours)[s_job[p][1 + f] + k - 2 * N - 1];
    }
    __syncthreads();

    // ---- phase 1: projected reference arrays for the 33 angular modes, DC values
    // only the entries a mode can read are produced: indices 0 .. 2N for the non-negative angles, last .. N for the
    // negative ones (last = (N * angle) >> 5 >= -N): at most 2N + 1 per mode
    for (int i = lane; i < P * 33 * (2 * N + 1); i += THREADS)
    {
        const int p = i / (33 * (2 * N + 1)), r = i - p * 33 * (2 * N + 1);
        const int mi = r / (2 * N + 1), e = r - mi * (2 * N + 1);
        const int mode = mi + 2;
        const int angle = c_angle35[mode];
        const bool vertical = mode >= 18;
        const int last = angle < 0 ? (N * angle) >> 5 : 0;
        const int idx = (last < -1 ? last : 0) + e;
        if (idx > (angle < 0 ? N : 2 * N)) continue;
        const uint32_t fbits = mode < 32 ? (uint32_t)s_job[p][3] >> mode : (uint32_t)s_job[p][4] >> (mode - 32);
        const uint16_t *nb = s_nb[p][fbits & 1];
        int v;
        if (idx >= 0) v = vertical ? nb[2 * N + idx] : nb[2 * N - idx];   // p(-1+idx,-1) / p(-1,-1+idx)
        else
        {
            const int k = -1 + ((idx * (int)c_invAngle35[mode] + 128) >> 8);
            v = vertical ? nb[2 * N - 1 - k] : nb[2 * N + 1 + k];          // p(-1,k) / p(k,-1)
        }
        s_ref[p][mi][idx + N] = (uint16_t)v;
    }
    if (lane < 2 * P)
    {
        const uint16_t *nb = s_nb[lane >> 1][lane & 1];
        int s = N;
        for (int k = 0; k < N; ++k) s += nb[2 * N + 1 + k] + nb[2 * N - 1 - k];
        s_dc[lane >> 1][lane & 1] = s >> (LOG2 + 1);
    }
    __syncthreads();

    // ---- phase 2: one (mode, partition, tile) item per lane per step; mode-major ordering
    for (int it = lane; it < 35 * PT; it += THREADS)
    {
        const int mode = it / PT, pt = it - mode * PT;
        const int p = pt / NT, tile = pt - p * NT;
        const int ty = tile / TPR, tx = tile - ty * TPR;
        const uint32_t fbits = mode < 32 ? (uint32_t)s_job[p][3] >> mode : (uint32_t)s_job[p][4] >> (mode - 32);
        const uint16_t *nb = s_nb[p][fbits & 1];
        const bool edge = s_job[p][5] != 0 && LOG2 < 5;
        TileDiff<S, TS> td;
        if (mode >= 2)
        {
            const int angle = c_angle35[mode];
            const bool vertical = mode >= 18;
            const uint16_t *ref = &s_ref[p][mode - 2][N];
            const uint16_t *sb = vertical ? s_src[p] : s_srcT[p];
            const int maj0 = (vertical ? ty : tx) * TS, min0 = (vertical ? tx : ty) * TS;
            const bool efilt = edge && min0 == 0 && (mode == 26 || mode == 10);
#pragma unroll
            for (int j = 0; j < TS; ++j)
            {
                const int t = (maj0 + j + 1) * angle;
                const int idx = t >> 5, fact = t & 31;
                const uint16_t *r = ref + min0 + idx + 1;
                int rv[TS + 1], v[TS];
#pragma unroll
                for (int i = 0; i <= TS; ++i) rv[i] = r[i];
#pragma unroll
                for (int i = 0; i < TS; ++i) v[i] = fact ? ((32 - fact) * rv[i] + fact * rv[i + 1] + 16) >> 5 : rv[i];
                if (efilt)
                {   // pred_intra.cpp:20355-20360 / :20394-20399: first column (row) of vertical (horizontal) prediction
                    const int side = vertical ? nb[2 * N - 1 - (maj0 + j)] : nb[2 * N + 1 + (maj0 + j)];
                    v[0] = clip3(0, maxv, (int)ref[1] + ((side - (int)ref[0]) >> 1));
                }
                td.set_row(j, sb + (maj0 + j) * NS + min0, v);
            }
        }
        else if (mode == 1)
        {
            const int dc = s_dc[p][fbits & 1];
#pragma unroll
            for (int j = 0; j < TS; ++j)
            {
                const int y = ty * TS + j;
                int v[TS];
#pragma unroll
                for (int i = 0; i < TS; ++i)
                {
                    const int x = tx * TS + i;
                    v[i] = dc;
                    if (edge)
                    {
                        if (x == 0 && y == 0) v[i] = ((int)nb[2 * N - 1] + 2 * dc + (int)nb[2 * N + 1] + 2) >> 2;
                        else if (y == 0) v[i] = ((int)nb[2 * N + 1 + x] + 3 * dc + 2) >> 2;
                        else if (x == 0) v[i] = ((int)nb[2 * N - 1 - y] + 3 * dc + 2) >> 2;
                    }
                }
                td.set_row(j, s_src[p] + y * NS + tx * TS, v);
            }
        }
        else
        {
            const int topR = nb[2 * N + 1 + N], botL = nb[2 * N - 1 - N];   // p(N,-1), p(-1,N)
#pragma unroll
            for (int j = 0; j < TS; ++j)
            {
                const int y = ty * TS + j;
                const int left = nb[2 * N - 1 - y];
                int v[TS];
#pragma unroll
                for (int i = 0; i < TS; ++i)
                {
                    const int x = tx * TS + i;
                    v[i] = ((N - 1 - x) * left + (x + 1) * topR + (N - 1 - y) * (int)nb[2 * N + 1 + x] + (y + 1) * botL + N) >> (LOG2 + 1);
                }
                td.set_row(j, s_src[p] + y * NS + tx * TS, v);
            }
        }
        const int c = td.satd();
        if (NT == 1) s_cost[p][mode] = c;
        else atomicAdd(&s_cost[p][mode], c);
    }
    __syncthreads();
    for (int i = lane; i < P * 35; i += THREADS)
    {
        const int p = i / 35, m = i - p * 35;
        if (job0 + p < njobs) cost[(long)(job0 + p) * 35 + m] = s_cost[p][m];
    }
}

template <int S>
static hipError_t launch_intra_satd35_s(hipStream_t st, int log2, int bitDepth, const void *src, long ss, const void *nb, const void *jobs, int n,
                                        int32_t *cost)
{
    const char *s = (const char *)src, *q = (const char *)nb;
    const int32_t *j = (const int32_t *)jobs;
    switch (log2)
    {
    // (partitions per workgroup, threads): items = 35 * P * tiles -> 245 / 245 / 700 / 560: 1, 1, 3, 3 full steps
    case 2: hipLaunchKernelGGL((k_intra_satd35<S, 2, 7, 256>), dim3((n + 6) / 7), dim3(256), 0, st, s, ss, q, j, n, bitDepth, cost); break;
    case 3: hipLaunchKernelGGL((k_intra_satd35<S, 3, 7, 256>), dim3((n + 6) / 7), dim3(256), 0, st, s, ss, q, j, n, bitDepth, cost); break;
    case 4: hipLaunchKernelGGL((k_intra_satd35<S, 4, 5, 256>), dim3((n + 4) / 5), dim3(256), 0, st, s, ss, q, j, n, bitDepth, cost); break;
    case 5: hipLaunchKernelGGL((k_intra_satd35<S, 5, 1, 192>), dim3(n), dim3(192), 0, st, s, ss, q, j, n, bitDepth, cost); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_intra_satd35(hipStream_t st, int S, int log2, int bitDepth, const void *src, long ss, const void *nb, const void *jobs, int n,
                               int32_t *cost)
{
    if (n <= 0) return hipSuccess;
    return S == 1 ? launch_intra_satd35_s<1>(st, log2, bitDepth, src, ss, nb, jobs, n, cost)
                  : launch_intra_satd35_s<2>(st, log2, bitDepth, src, ss, nb, jobs, n, cost);
}

} // namespace havoc_gpu
