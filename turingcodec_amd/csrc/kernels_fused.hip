// Fused search kernels: the batches the encoder's decision loops issue as (predict, measure) PAIRS, evaluated
// without ever writing the prediction to HBM.
//
//   k_intra_satd35 : the 35-mode luma SATD stage of searchIntraPartition (turing/Search.hpp:113-142 calling
//                    PredictIntraLumaBlock, turing/Reconstruct.cpp:630-701): for one partition, all 35 intra predictions
//                    (havoc/pred_intra.cpp:20282-20401) each followed by the Hadamard SATD against the source block
//                    (8x8 tiles, or one 4x4 for 4x4 blocks; havoc/hadamard.cpp:58-98).
//
// Mapping: one workgroup owns P partitions of one size; source block (both orientations), both neighbour arrays
// (unfiltered / filtered, plus the left column reversed) and the projected reference arrays of the 15 negative-angle
// modes live in LDS.  A work item is (group of 5 modes, partition, SATD tile): one LANE keeps its source tile in
// registers, and per mode predicts the tile into registers two samples per instruction (packed 16-bit lerp), subtracts,
// runs the 2-D Hadamard in registers and adds the tile cost into the partition's 35 LDS accumulators.  Horizontal
// modes are evaluated in the transposed domain (prediction^T against source^T): SATD is invariant under
// transposition, and it makes the horizontal and vertical modes one code path.  The kernel is bound by LDS accesses
// (scattered 16-bit reference reads conflict on banks) more than by VALU, hence the register-resident source tile,
// the one-vector reference read per row and the rotation instead of a second read.
#include "common.h"

namespace havoc_gpu {

// the projected (index < 0) entries of the negative-angle modes' reference arrays (pred_intra.cpp:20330-20345, 20370-20384):
// entry = { mode - 11, index + N, k + 1 } with ref[index] = p(-1, k) (vertical modes) or p(k, -1) (horizontal modes)
template <int N>
struct ProjTable
{
    int n;
    uint16_t e[15 * N];
    constexpr ProjTable() : n(0), e{}
    {
        for (int mode = 11; mode <= 25; ++mode)
        {
            const int angle = angle_of(mode), inv = inv_angle_of(mode);
            int last = (N * angle) / 32;
            if (last * 32 > N * angle) --last;   // arithmetic shift: floor
            if (last >= -1) continue;
            for (int idx = last; idx <= -1; ++idx)
            {
                int q = idx * inv + 128;         // > 0
                const int k = -1 + q / 256;
                e[n++] = (uint16_t)((mode - 11) | ((idx + N) << 4) | ((k + 1) << 10));
            }
        }
    }
};
template <int N> __constant__ ProjTable<N> c_proj = ProjTable<N>();

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
    // row j = source row s2 (packed pairs, held in registers across the item's modes) minus prediction row v2
    __device__ __forceinline__ void set_row_pk(int j, const uint32_t (&s2)[TS / 2], const uint32_t (&v2)[TS / 2])
    {
#pragma unroll
        for (int k = 0; k < TS / 2; ++k)
        {
            if constexpr (S == 1) p[j][k] = pk_sub(s2[k], v2[k]);
            else
            {
                d[j][2 * k] = (int)(s2[k] & 0xffffu) - (int)(v2[k] & 0xffffu);
                d[j][2 * k + 1] = (int)(s2[k] >> 16) - (int)(v2[k] >> 16);
            }
        }
    }
    __device__ __forceinline__ int satd()
    {
        if constexpr (S == 1) return satd_regs_pk<TS>(p);
        else return satd_regs<S, TS>(d);
    }
};

// a lane evaluates MPI (modes per item: 5, or 1 for the single-tile sizes) consecutive entries of this list for one tile: planar, DC and the vertical family
// (source tile as stored), then the horizontal family (transposed source tile) -- one orientation switch in the list
// job: havoc_mi355x_intra_search_job = { src_off, nb_off, nbf_off, filt_lo, filt_hi, edge, reserved[2] }
template <int S, int LOG2, int P, int THREADS, int MPI>
__global__ __launch_bounds__(THREADS) void k_intra_satd35(const char *__restrict__ src, long stride_src, const char *__restrict__ neighbours,
                                                     const int32_t *__restrict__ jobs, int njobs, int bitDepth, int32_t *__restrict__ cost)
{
    typedef typename Sample<S>::T T;
    constexpr int N = 1 << LOG2;
    constexpr int TS = N >= 8 ? 8 : 4;       // SATD tile (Reconstruct.cpp:684-701)
    constexpr int TPR = N / TS, NT = TPR * TPR;
    constexpr int NB = 4 * N + 1;
    constexpr int NL = 2 * N + 2;            // top / left run: corner + 2N samples (+1 slack)
    constexpr int RL = 2 * N + 2;            // projected reference of a negative-angle mode: indices -N .. N (+1 slack)
    constexpr int PT = P * NT;               // (partition, tile) pairs per workgroup
    constexpr int kModeGroups = 35 / MPI;
    static_assert(kModeGroups * MPI == 35 && kModeGroups * PT <= THREADS, "one item per thread");

    // row stride N+2 and a per-partition skew keep tiles of different rows / partitions on different LDS banks
    // (with dense N*N blocks every partition and every 8-row band started on bank 0: 10-way conflicts on 16x16)
    constexpr int NS = N + 2;
    constexpr int SB = N * NS + 6;
    __shared__ __attribute__((aligned(16))) uint16_t s_src[P][SB];        // row-major source block
    __shared__ __attribute__((aligned(16))) uint16_t s_srcT[P][SB];       // transposed source block
    __shared__ __attribute__((aligned(16))) uint16_t s_nb[P][2][NB + 1];  // [0] unfiltered, [1] filtered; index i <-> neighbours[i - 2N - 1]
    __shared__ __attribute__((aligned(16))) uint16_t s_left[P][2][NL];    // the left column read downwards: [i] = p(-1, -1+i) = nb[2N - i]
    __shared__ __attribute__((aligned(16))) uint16_t s_ref[P][15][RL];    // negative-angle modes 11..25; ref[i] at [i + N]
    __shared__ int s_dc[P][2];
    __shared__ int s_cost[P][36];
    __shared__ int s_job[P][8];

    const int lane = threadIdx.x;
    const int job0 = xcd_block(blockIdx.x, gridDim.x) * P;
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
    // neighbour arrays, four samples per load (NB = 4N+1: N quads + the last sample)
    constexpr int NQ = N + 1;
    for (int i = lane; i < P * 2 * NQ; i += THREADS)
    {
        const int p = i / (2 * NQ), r = i - p * 2 * NQ;
        const int f = r / NQ, k = (r - f * NQ) * 4;
        const T *n0 = reinterpret_cast<const T *>(neighbours) + s_job[p][1 + f] - 2 * N - 1;
        uint16_t v[4];
        int cnt = 4;
        if (k + 4 <= NB)
        {
            if (S == 1)
            {
                const uint32_t w = ld4(n0 + k);
                v[0] = w & 0xff; v[1] = (w >> 8) & 0xff; v[2] = (w >> 16) & 0xff; v[3] = w >> 24;
            }
            else
            {
                const u32x2 w = ld8(n0 + k);
                v[0] = w.x & 0xffff; v[1] = w.x >> 16; v[2] = w.y & 0xffff; v[3] = w.y >> 16;
            }
        }
        else
        {
            cnt = NB - k;   // 1
            v[0] = n0[k]; v[1] = v[2] = v[3] = 0;
        }
        for (int q = 0; q < cnt; ++q)
        {
            s_nb[p][f][k + q] = v[q];
            if (k + q <= 2 * N) s_left[p][f][2 * N - k - q] = v[q];
        }
    }
    __syncthreads();

    // ---- phase 1: reference arrays of the 15 negative-angle modes (the others read the top row / the left column as
    // they are): entries 0 .. N are a dword copy of the top row / left column, the negative ones come from the table
    constexpr int CD = N / 2 + 1;            // dwords holding entries 0 .. N (+1)
    for (int i = lane; i < P * 15 * CD; i += THREADS)
    {
        const int p = i / (15 * CD), r = i - p * 15 * CD;
        const int mi = r / CD, c = r - mi * CD;
        const int mode = mi + 11;
        const int f = ((uint32_t)s_job[p][3] >> mode) & 1;
        const uint32_t *from = reinterpret_cast<const uint32_t *>(mode >= 18 ? &s_nb[p][f][2 * N] : &s_left[p][f][0]);
        reinterpret_cast<uint32_t *>(&s_ref[p][mi][N])[c] = from[c];
    }
    for (int i = lane; i < P * c_proj<N>.n; i += THREADS)
    {
        const int p = i / c_proj<N>.n, r = i - p * c_proj<N>.n;
        const int e = c_proj<N>.e[r];
        const int mi = e & 15, at = (e >> 4) & 63, k = (e >> 10) - 1;
        const int mode = mi + 11;
        const uint16_t *nb = s_nb[p][((uint32_t)s_job[p][3] >> mode) & 1];
        s_ref[p][mi][at] = mode >= 18 ? nb[2 * N - 1 - k] : nb[2 * N + 1 + k];   // p(-1,k) / p(k,-1)
    }
    if (lane < 2 * P)
    {
        const uint16_t *nb = s_nb[lane >> 1][lane & 1];
        int s = N;
        for (int k = 0; k < N; ++k) s += nb[2 * N + 1 + k] + nb[2 * N - 1 - k];
        s_dc[lane >> 1][lane & 1] = s >> (LOG2 + 1);
    }
    __syncthreads();

    // ---- phase 2: one item per lane = (mode group, partition, tile); the source tile stays in registers across the
    // group's modes.  Group-major ordering: the lanes of a wavefront mostly run the same modes.
    if (lane < kModeGroups * PT)
    {
        const int grp = lane / PT, pt = lane - grp * PT;
        const int p = pt / NT, tile = pt - p * NT;
        const int ty = tile / TPR, tx = tile - ty * TPR;
        const bool edge = s_job[p][5] != 0 && LOG2 < 5;
        uint32_t sp[TS][TS / 2];
        int have = -1;   // orientation of the tile in sp: 0 as stored, 1 transposed
#pragma unroll 1
        for (int m = 0; m < MPI; ++m)
        {
            const int o = grp * MPI + m;
            const int mode = o < 2 ? o : (o < 19 ? o + 16 : o - 17);   // planar, DC, 18..34, 2..17
            const bool vertical = mode >= 18;
            const int orient = (mode >= 2 && !vertical) ? 1 : 0;
            const int maj0 = (orient ? tx : ty) * TS, min0 = (orient ? ty : tx) * TS;
            if (orient != have)
            {
                const uint16_t *sb = (orient ? s_srcT[p] : s_src[p]) + maj0 * NS + min0;
#pragma unroll
                for (int j = 0; j < TS; ++j) ld_pairs<TS>(sb + j * NS, sp[j]);
                have = orient;
            }
            const uint32_t fbits = mode < 32 ? (uint32_t)s_job[p][3] >> mode : (uint32_t)s_job[p][4] >> (mode - 32);
            const int f = fbits & 1;
            const uint16_t *nb = s_nb[p][f];
            TileDiff<S, TS> td;
            if (mode >= 2)
            {
                const int angle = angle_of(mode);
                // ref[i], i >= 0: p(-1+i,-1) (vertical) or p(-1,-1+i) (horizontal); negative-angle modes add projected entries
                const uint16_t *ref = angle < 0 ? &s_ref[p][mode - 11][N] : (vertical ? nb + 2 * N : s_left[p][f]);
                const bool efilt = edge && min0 == 0 && angle == 0;
#pragma unroll
                for (int j = 0; j < TS; ++j)
                {
                    const int t = (maj0 + j + 1) * angle;
                    const int idx = t >> 5, fact = t & 31;
                    const uint16_t *r = ref + min0 + idx + 1;
                    uint32_t ra[TS / 2], rb[TS / 2], v2[TS / 2];
                    ld_pairs<TS>(r, ra);          // (ref[i], ref[i+1]) pairs; the pairs one entry further by rotation, not re-read
                    const uint32_t rl = r[TS];
#pragma unroll
                    for (int k = 0; k < TS / 2; ++k) rb[k] = __builtin_amdgcn_alignbit(k + 1 < TS / 2 ? ra[k + 1] : rl, ra[k], 16);
                    const uint32_t w1 = (uint32_t)fact * 0x00010001u, w0 = 0x00200020u - w1;
#pragma unroll
                    for (int k = 0; k < TS / 2; ++k) v2[k] = pk_lerp(ra[k], rb[k], w0, w1);
                    if (efilt)
                    {   // pred_intra.cpp:20355-20360 / :20394-20399: first column (row) of vertical (horizontal) prediction
                        const int side = vertical ? nb[2 * N - 1 - (maj0 + j)] : nb[2 * N + 1 + (maj0 + j)];
                        v2[0] = (v2[0] & 0xffff0000u) | (uint32_t)clip3(0, maxv, (int)ref[1] + ((side - (int)ref[0]) >> 1));
                    }
                    td.set_row_pk(j, sp[j], v2);
                }
            }
            else if (mode == 1)
            {
                const int dc = s_dc[p][f];
#pragma unroll
                for (int j = 0; j < TS; ++j)
                {
                    const int y = ty * TS + j;
                    uint32_t v2[TS / 2];
#pragma unroll
                    for (int k = 0; k < TS / 2; ++k) v2[k] = (uint32_t)dc * 0x00010001u;
                    if (edge && (y == 0 || tx == 0))
                    {
                        int v[TS];
#pragma unroll
                        for (int i = 0; i < TS; ++i)
                        {
                            const int x = tx * TS + i;
                            v[i] = dc;
                            if (x == 0 && y == 0) v[i] = ((int)nb[2 * N - 1] + 2 * dc + (int)nb[2 * N + 1] + 2) >> 2;
                            else if (y == 0) v[i] = ((int)nb[2 * N + 1 + x] + 3 * dc + 2) >> 2;
                            else if (x == 0) v[i] = ((int)nb[2 * N - 1 - y] + 3 * dc + 2) >> 2;
                        }
#pragma unroll
                        for (int k = 0; k < TS / 2; ++k) v2[k] = (uint32_t)v[2 * k] | ((uint32_t)v[2 * k + 1] << 16);
                    }
                    td.set_row_pk(j, sp[j], v2);
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
                    uint32_t v2[TS / 2];
#pragma unroll
                    for (int k = 0; k < TS / 2; ++k)
                    {
                        int v[2];
#pragma unroll
                        for (int e = 0; e < 2; ++e)
                        {
                            const int x = tx * TS + 2 * k + e;
                            v[e] = ((N - 1 - x) * left + (x + 1) * topR + (N - 1 - y) * (int)nb[2 * N + 1 + x] + (y + 1) * botL + N) >> (LOG2 + 1);
                        }
                        v2[k] = (uint32_t)v[0] | ((uint32_t)v[1] << 16);
                    }
                    td.set_row_pk(j, sp[j], v2);
                }
            }
            const int c = td.satd();
            if (NT == 1) s_cost[p][mode] = c;
            else atomicAdd(&s_cost[p][mode], c);
        }
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
    // (partitions per workgroup, threads, modes per item).  8x8 / 4x4 partitions are one tile: their set-up phases weigh as
    // much as the 35 predictions, so they keep 35 lanes per partition (245 items of 256 lanes); the multi-tile sizes run 5
    // modes per lane on a register-resident source tile (252 / 224 items of 256 lanes)
    case 2: hipLaunchKernelGGL((k_intra_satd35<S, 2, 7, 256, 1>), dim3((n + 6) / 7), dim3(256), 0, st, s, ss, q, j, n, bitDepth, cost); break;
    case 3: hipLaunchKernelGGL((k_intra_satd35<S, 3, 7, 256, 1>), dim3((n + 6) / 7), dim3(256), 0, st, s, ss, q, j, n, bitDepth, cost); break;
    case 4: hipLaunchKernelGGL((k_intra_satd35<S, 4, 9, 256, 5>), dim3((n + 8) / 9), dim3(256), 0, st, s, ss, q, j, n, bitDepth, cost); break;
    case 5: hipLaunchKernelGGL((k_intra_satd35<S, 5, 2, 256, 5>), dim3((n + 1) / 2), dim3(256), 0, st, s, ss, q, j, n, bitDepth, cost); break;
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
