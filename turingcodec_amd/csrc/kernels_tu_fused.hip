// The transform-unit chain of the reference's reconstruction (turing/Reconstruct.cpp:258-353 intra, :766-856 inter),
// fused on either side of the host's quantiser decision (RDOQ sits between the two halves):
//
//   k_tu_forward     : residual = source - prediction  (Reconstruct.cpp:258-260)  ->  forward DCT/DST
//                      (havoc/transform.cpp:3087-3397).  The residual is formed in registers from the two sample rows.
//   k_tu_reconstruct : de-quantise the levels (havoc/quantize.cpp:37-46)  ->  inverse DCT/DST + add prediction + clip
//                      (havoc/transform.cpp:50-401, transform.h:104-114)  ->  SSD of the reconstruction against the
//                      source (havoc/ssd.cpp:28-43).  De-quantised coefficients and the residual never leave the CU.
//
// Same mapping as kernels_tu.hip: one lane per TU row, 64/N TUs per wavefront, basis pairs through SGPRs into
// v_dot2c_i32_i16, padded LDS transpose between the passes.  Results are bit-identical to running the separate
// kernels (tests: tu_fused groups).
#include "common.h"

#include <cstdlib>
#include "transform_basis.h"
#include "rdoq_work.h"

namespace havoc_gpu {

// N samples of one row as N/2 packed 16-bit pairs (x, x+1)
template <int S, int N>
__device__ __forceinline__ void load_sample_row(const char *p, uint32_t (&row)[N / 2])
{
    if (S == 1)
    {
        uint32_t b[N / 4];
        if (N == 4) b[0] = ld4(p);
        else if (N == 8)
        {
            const u32x2 v = ld8(p);
            b[0] = v.x; b[1] = v.y;
        }
        else
        {
#pragma unroll
            for (int q = 0; q < N / 16; ++q)
            {
                const u32x4 v = ld16(p + 16 * q);
                b[4 * q] = v.x; b[4 * q + 1] = v.y; b[4 * q + 2] = v.z; b[4 * q + 3] = v.w;
            }
        }
#pragma unroll
        for (int q = 0; q < N / 4; ++q)
        {
            row[2 * q] = __builtin_amdgcn_perm(0u, b[q], 0x0c010c00u);       // (byte0, byte1) zero-extended
            row[2 * q + 1] = __builtin_amdgcn_perm(0u, b[q], 0x0c030c02u);   // (byte2, byte3)
        }
    }
    else
    {
        if (N == 4)
        {
            const u32x2 v = ld8(p);
            row[0] = v.x; row[1] = v.y;
        }
        else
        {
#pragma unroll
            for (int q = 0; q < N / 8; ++q)
            {
                const u32x4 v = ld16(p + 16 * q);
                row[4 * q] = v.x; row[4 * q + 1] = v.y; row[4 * q + 2] = v.z; row[4 * q + 3] = v.w;
            }
        }
    }
}

// job: havoc_mi355x_tu_fused_job { coef_off, src_off, pred_off, rec_off }
// SCAN (16x16 / 32x32 DCT): the first pass of Rdoq::runQuantisation's device form is done HERE, where the coefficients are in registers --
// which 4x4 groups hold a rounded level > 0 / > 1 / > 2 and the block's energy go to the RDOQ workspace (RdoqInfo of block `job`), and the
// level block RDOQ will write is zeroed -- instead of a separate kernel reading the coefficients back (k_rdoq_scan: 2 x 47 MB and 47 us
// per 1080p picture).  rjobs[job] = the havoc_mi355x_rdoq_job of the same block.
template <int S, int LOG2, int TR, bool SCAN = false>
__global__ __launch_bounds__(64) void k_tu_forward(int16_t *__restrict__ coeffs, const char *__restrict__ src, long stride_src,
                                                   const char *__restrict__ pred, long stride_pred, const int32_t *__restrict__ jobs, int njobs,
                                                   int bitDepth, const RdoqJob *__restrict__ rjobs = nullptr, int16_t *__restrict__ levels = nullptr,
                                                   RdoqWork *__restrict__ work = nullptr)
{
    constexpr int N = 1 << LOG2, TPW = 64 / N, LS = N + 2;
    __shared__ int16_t lds[TPW][N * LS];
    const int t = threadIdx.x / N, r = threadIdx.x % N;
    const int job = xcd_block(blockIdx.x, gridDim.x) * TPW + t;
    const bool live = job < njobs;
    const int32_t *j = jobs + (long)(live ? job : 0) * 4;
    const int shift1 = LOG2 - 1 + bitDepth - 8, shift2 = LOG2 + 6;

    uint32_t row[N / 2], prow[N / 2];
    load_sample_row<S, N>(src + ((long)j[1] + (long)r * stride_src) * S, row);
    load_sample_row<S, N>(pred + ((long)j[2] + (long)r * stride_pred) * S, prow);
#pragma unroll
    for (int p = 0; p < N / 2; ++p) row[p] = pk_sub(row[p], prow[p]);   // residual (|.| <= 1023: no 16-bit overflow)
    int o[N];
    basis_times_row<N, TR, false, true>(row, 1 << (shift1 - 1), o);   // residual of <= 10-bit samples: the fold is exact
#pragma unroll
    for (int k = 0; k < N; ++k) lds[t][k * LS + r] = (int16_t)(o[k] >> shift1);
    __syncthreads();
#pragma unroll
    for (int p = 0; p < N / 2; ++p) row[p] = *reinterpret_cast<const uint32_t *>(&lds[t][r * LS + 2 * p]);
    basis_times_row<N, TR, false>(row, 1 << (shift2 - 1), o);
    if (SCAN)
    {
        // lane r holds column r of the block: o[k] >> shift2 = coefficient (row k, column r).  Group (gx = r / 4, gy): the largest magnitude of
        // its 16 coefficients = max over this lane's rows 4 gy .. 4 gy + 3, then over the four lanes of the quad
        constexpr int GW = N / 4;
        const RdoqJob rj = rjobs[live ? job : 0];
        uint32_t thr[3];
        rdoqThresholds(rj.quant_scale, rj.quant_shift, thr);
        uint32_t lo = 0, hi = 0;
        uint64_t mask[3] = {0, 0, 0};
#pragma unroll
        for (int gy = 0; gy < GW; ++gy)
        {
            uint32_t big = 0;
#pragma unroll
            for (int k = 4 * gy; k < 4 * gy + 4; ++k)
            {
                const int c = (int16_t)(o[k] >> shift2);
                big = max(big, (uint32_t)abs(c));        // -32768 -> 32768
                const uint32_t sq = (uint32_t)(c * c);    // <= 2^30
                lo += sq & 0xffffu;
                hi += sq >> 16;
            }
            big = max(big, (uint32_t)__shfl_xor((int)big, 1, kWave));
            big = max(big, (uint32_t)__shfl_xor((int)big, 2, kWave));
#pragma unroll
            for (int m = 0; m < 3; ++m)
            {
                // one bit per quad (its first lane), the block's N lanes are a field of the ballot: compress bits 4 gx -> gx
                const uint64_t b = __ballot(live && (r & 3) == 0 && big >= thr[m]) >> (t * N);
                uint32_t f = 0;
#pragma unroll
                for (int gx = 0; gx < GW; ++gx) f |= (uint32_t)((b >> (4 * gx)) & 1) << gx;
                mask[m] |= (uint64_t)f << (gy * GW);
            }
        }
        const int slo = group_sum<N>((int)lo), shi = group_sum<N>((int)hi);
        if (live)
        {
            // the level block, zeroed row by row (RDOQ writes only the levels it keeps)
            int16_t *z = levels + rj.dst_off + r * N;      // offsets are multiples of 4 levels: 8-byte stores
#pragma unroll
            for (int q = 0; q < N / 4; ++q) st8(z + 4 * q, u32x2{0, 0});
            if (r == 0)
            {
                RdoqInfo *info = reinterpret_cast<RdoqInfo *>(reinterpret_cast<char *>(work) + rdoqInfoOffset());
                RdoqInfo v;
                v.mask = mask[0];
                v.mask2 = mask[1];
                v.mask3 = mask[2];
                v.sumSq = ((int64_t)shi << 16) + slo;
                info[job] = v;
            }
        }
    }
    if (!live) return;
    int16_t *c = coeffs + j[0] + r;
#pragma unroll
    for (int k = 0; k < N; ++k) c[k * N] = (int16_t)(o[k] >> shift2);
}

template <int S, int LOG2, int TR>
__global__ __launch_bounds__(64) void k_tu_reconstruct(char *rec, long stride_rec, const char *pred, long stride_pred, const char *__restrict__ src,
                                                       long stride_src, const int16_t *__restrict__ levels, const int32_t *__restrict__ jobs,
                                                       int njobs, int bitDepth, int scale, int shift, uint32_t *__restrict__ ssd)
{
    typedef typename Sample<S>::T T;
    constexpr int N = 1 << LOG2, TPW = 64 / N, LS = N + 2;
    __shared__ int16_t lds[TPW][N * LS];
    const int t = threadIdx.x / N, r = threadIdx.x % N;
    const int job = xcd_block(blockIdx.x, gridDim.x) * TPW + t;
    const bool live = job < njobs;
    const int32_t *j = jobs + (long)(live ? job : 0) * 4;
    const int shift2 = 20 - bitDepth;
    const int dqadd = 1 << (shift - 1);

    // column r of the level block, de-quantised on the fly, packed in row pairs
    const int16_t *c = levels + j[0] + r;
    uint32_t col[N / 2];
#pragma unroll
    for (int p = 0; p < N / 2; ++p)
    {
        const int lo = clip3(-32768, 32767, ((int)c[(2 * p) * N] * scale + dqadd) >> shift);
        const int hi = clip3(-32768, 32767, ((int)c[(2 * p + 1) * N] * scale + dqadd) >> shift);
        col[p] = ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16);
    }
    int o[N];
    basis_times_row<N, TR, true>(col, 1 << 6, o);
#pragma unroll
    for (int k = 0; k < N; ++k) lds[t][k * LS + r] = (int16_t)clip3(-32768, 32767, o[k] >> 7);
    __syncthreads();
#pragma unroll
    for (int p = 0; p < N / 2; ++p) col[p] = *reinterpret_cast<const uint32_t *>(&lds[t][r * LS + 2 * p]);
    basis_times_row<N, TR, true>(col, 1 << (shift2 - 1), o);

    const int maxv = (1 << bitDepth) - 1;
    uint32_t prow[N / 2], srow[N / 2];
    load_sample_row<S, N>(pred + ((long)j[2] + (long)r * stride_pred) * S, prow);   // whole row read before any write: pred may alias rec
    load_sample_row<S, N>(src + ((long)j[1] + (long)r * stride_src) * S, srow);
    uint32_t acc = 0;
    int v[N];
#pragma unroll
    for (int k = 0; k < N; ++k)
    {
        const int res = clip3(-32768, 32767, o[k] >> shift2);
        const int pk = (k & 1) ? (int)(prow[k >> 1] >> 16) : (int)(prow[k >> 1] & 0xffff);
        const int sk = (k & 1) ? (int)(srow[k >> 1] >> 16) : (int)(srow[k >> 1] & 0xffff);
        v[k] = clip3(0, maxv, pk + res);
        const int d = sk - v[k];
        acc += (uint32_t)(d * d);
    }
    if (live)
    {
        T *q = reinterpret_cast<T *>(rec) + j[3] + (long)r * stride_rec;
        if (S == 1)
        {
#pragma unroll
            for (int k = 0; k < N; k += 4) st4(q + k, (uint32_t)v[k] | ((uint32_t)v[k + 1] << 8) | ((uint32_t)v[k + 2] << 16) | ((uint32_t)v[k + 3] << 24));
        }
        else
        {
#pragma unroll
            for (int k = 0; k < N; k += 2) st4(q + k, (uint32_t)v[k] | ((uint32_t)v[k + 1] << 16));
        }
    }
    // SSD of the TU: sum over its N lanes (uint32 accumulation as havoc_ssd_c_ref; 16-bit result >> 4)
    uint32_t tot = (uint32_t)group_sum<N>((int)acc);
    if (S == 2) tot >>= 4;
    if (live && r == 0) ssd[job] = tot;
}

// The 35-mode stage of ONE intra partition in one launch (round 6; the per-block table API's serve layer, csrc/classic.cpp): job i = mode slot i of the partition --
// its prediction block (pred_off, row stride stride_pred) against the partition's source block (src_off) -- and per job
//   satd[i * tiles + t]   havoc_hadamard_satd of tile t (8x8 tiles; 4x4 for a 4x4 partition), raster order          = havoc_mi355x_satd on those tiles
//   coeffs[coef_off ..]   forward transform of source - prediction, DST-VII for 4x4 / DCT above (Reconstruct.cpp:263)  = havoc_mi355x_tu_forward
//   coeffsDct[coef_off ..] 4x4 only: the DCT of the same residual (a chroma block's type)                              = havoc_mi355x_tu_forward, trType 0
//   rec0[rec_off ..], ssd0[i]  the reconstruction from a block of ZERO levels and its SSD against the source           = havoc_mi355x_tu_reconstruct on zero levels
// (de-quantised zeros are zeros and their inverse transform rounds to zero, so the reconstruction is the clipped prediction; the SSD as havoc_ssd: 16-bit >> 4).
// Same mapping as k_tu_forward -- a lane per row, 64 / N blocks per wavefront -- whose residual rows are exactly what the tile SATDs and the SSD need.
template <int S, int LOG2>
__global__ __launch_bounds__(64) void k_intra_measure(int16_t *__restrict__ coeffs, int16_t *__restrict__ coeffsDct, int32_t *__restrict__ satd, char *__restrict__ rec0,
                                                      uint32_t *__restrict__ ssd0, const char *__restrict__ src, long stride_src, const char *__restrict__ pred,
                                                      long stride_pred, const int32_t *__restrict__ jobs, int njobs, int bitDepth, int withSatd)
{
    typedef typename Sample<S>::T T;
    constexpr int N = 1 << LOG2, TPW = 64 / N, LS = N + 2, TS = N >= 8 ? 8 : 4, TX = N / TS, TILES = TX * TX;
    __shared__ int16_t lds[TPW][N * LS];
    const int t = threadIdx.x / N, r = threadIdx.x % N;
    const int job = blockIdx.x * TPW + t;
    const bool live = job < njobs;
    const int32_t *j = jobs + (long)(live ? job : 0) * 4;
    const int shift1 = LOG2 - 1 + bitDepth - 8, shift2 = LOG2 + 6;

    uint32_t row[N / 2], prow[N / 2];
    load_sample_row<S, N>(src + ((long)j[1] + (long)r * stride_src) * S, row);
    load_sample_row<S, N>(pred + ((long)j[2] + (long)r * stride_pred) * S, prow);
    // zero-level reconstruction = the prediction, clipped like inverse_transform_add's output
    if (live)
    {
        const int maxv = (1 << bitDepth) - 1;
        T *q = reinterpret_cast<T *>(rec0) + j[3] + (long)r * N;
#pragma unroll
        for (int p = 0; p < N / 2; ++p)
        {
            q[2 * p] = (T)clip3(0, maxv, (int)(prow[p] & 0xffff));
            q[2 * p + 1] = (T)clip3(0, maxv, (int)(prow[p] >> 16));
        }
    }
#pragma unroll
    for (int p = 0; p < N / 2; ++p) row[p] = pk_sub(row[p], prow[p]);   // residual (|.| <= 1023: no 16-bit overflow)
    // SSD(source, zero-level reconstruction): the prediction is inside the sample range, so the difference is the residual
    {
        uint32_t acc = 0;
#pragma unroll
        for (int p = 0; p < N / 2; ++p)
        {
            const int lo = (int16_t)(row[p] & 0xffff), hi = (int16_t)(row[p] >> 16);
            acc += (uint32_t)(lo * lo) + (uint32_t)(hi * hi);
        }
        uint32_t tot = (uint32_t)group_sum<N>((int)acc);
        if (S == 2) tot >>= 4;
        if (live && r == 0) ssd0[job] = tot;
    }
    if (withSatd)      // (uniform)
    {
#pragma unroll
        for (int tx = 0; tx < TX; ++tx)
        {
            int c;
            if (S == 1)
            {
                uint32_t p[TS / 2];
#pragma unroll
                for (int k = 0; k < TS / 2; ++k) p[k] = row[tx * (TS / 2) + k];
                c = satd_rows_pk<TS>(p, r & (TS - 1));
            }
            else
            {
                int d[TS];
#pragma unroll
                for (int k = 0; k < TS / 2; ++k)
                {
                    d[2 * k] = (int16_t)(row[tx * (TS / 2) + k] & 0xffff);
                    d[2 * k + 1] = (int16_t)(row[tx * (TS / 2) + k] >> 16);
                }
                c = satd_rows<S, TS>(d, r & (TS - 1));
            }
            if (live && (r & (TS - 1)) == 0) satd[(long)job * TILES + (r / TS) * TX + tx] = c;
        }
    }
    uint32_t keep[N / 2];
#pragma unroll
    for (int p = 0; p < N / 2; ++p) keep[p] = row[p];
    int o[N];
    constexpr int TR = LOG2 == 2 ? 1 : 0;
    basis_times_row<N, TR, false, true>(row, 1 << (shift1 - 1), o);
#pragma unroll
    for (int k = 0; k < N; ++k) lds[t][k * LS + r] = (int16_t)(o[k] >> shift1);
    __syncthreads();
#pragma unroll
    for (int p = 0; p < N / 2; ++p) row[p] = *reinterpret_cast<const uint32_t *>(&lds[t][r * LS + 2 * p]);
    basis_times_row<N, TR, false>(row, 1 << (shift2 - 1), o);
    if (live)
    {
        int16_t *c = coeffs + j[0] + r;
#pragma unroll
        for (int k = 0; k < N; ++k) c[k * N] = (int16_t)(o[k] >> shift2);
    }
    if (LOG2 == 2)
    {
        __syncthreads();
        basis_times_row<N, 0, false, true>(keep, 1 << (shift1 - 1), o);
#pragma unroll
        for (int k = 0; k < N; ++k) lds[t][k * LS + r] = (int16_t)(o[k] >> shift1);
        __syncthreads();
#pragma unroll
        for (int p = 0; p < N / 2; ++p) row[p] = *reinterpret_cast<const uint32_t *>(&lds[t][r * LS + 2 * p]);
        basis_times_row<N, 0, false>(row, 1 << (shift2 - 1), o);
        if (live)
        {
            int16_t *c = coeffsDct + j[0] + r;
#pragma unroll
            for (int k = 0; k < N; ++k) c[k * N] = (int16_t)(o[k] >> shift2);
        }
    }
}

hipError_t launch_intra_measure(hipStream_t st, int S, int bd, int log2, int16_t *coeffs, int16_t *coeffsDct, int32_t *satd, void *rec0, uint32_t *ssd0, const void *src,
                                long ss, const void *pred, long sp, const void *jobs, int n, int withSatd)
{
    if (n <= 0) return hipSuccess;
    if (log2 < 2 || log2 > 5) return hipErrorInvalidValue;
    const int tpw = 64 >> log2;
    const dim3 g((n + tpw - 1) / tpw), b(64);
    char *r0 = (char *)rec0;
    const char *s8 = (const char *)src, *p8 = (const char *)pred;
    const int32_t *j = (const int32_t *)jobs;
#define MEASURE_GO(SS, LL) hipLaunchKernelGGL((k_intra_measure<SS, LL>), g, b, 0, st, coeffs, coeffsDct, satd, r0, ssd0, s8, ss, p8, sp, j, n, bd, withSatd)
    if (S == 1) { if (log2 == 2) MEASURE_GO(1, 2); else if (log2 == 3) MEASURE_GO(1, 3); else if (log2 == 4) MEASURE_GO(1, 4); else MEASURE_GO(1, 5); }
    else { if (log2 == 2) MEASURE_GO(2, 2); else if (log2 == 3) MEASURE_GO(2, 3); else if (log2 == 4) MEASURE_GO(2, 4); else MEASURE_GO(2, 5); }
#undef MEASURE_GO
    return hipGetLastError();
}


// ---- The 32x32 forward DCT of 8-bit content on the MATRIX cores (round 6).  The transform is two matrix products with the constant basis M (|M| <= 90: int8):
//   T = (M * R^T + 8) >> 4 (wrapped to int16),   C = (M * T^T + 1024) >> 11   -- exactly what k_tu_forward<1, 5> computes with 768 v_dot2 per lane.
// v_mfma_i32_32x32x32_i8 (gfx950) multiplies a 32 x 32 int8 A by a 32 x 32 int8 B in ONE instruction: lane l holds row / column (l & 31) and the 16 K values
// 16 (l >> 5) .. + 15; D[row][col] lands in register reg of lane l with col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5).  The second operand is 16 bits
// wide (9 in the first pass), so it goes in as TWO byte planes -- v = 256 vh + (vl' + 128) with vh = v >> 8 (the high byte as it stands) and vl' = the low byte with
// its top bit flipped -- and M v = 256 (M vh) + M vl' + 128 rowsum(M): two MFMAs per pass, the constant in the accumulator's initial value, everything exact in int32.
// A wavefront per block (lane = row y, half of its 32 samples): ~350 vector instructions per block where the dot2 form issues ~940 (the step is bound by
// instruction issue).  SCAN as in k_tu_forward.
typedef int i32x16v __attribute__((ext_vector_type(16)));
typedef int i32x4v __attribute__((ext_vector_type(4)));

struct Dct32I8 { int8_t m[32][32]; int32_t rowsum128[32]; };
constexpr Dct32I8 make_dct32_i8()
{
    Dct32I8 t{};
    for (int k = 0; k < 32; ++k)
    {
        int sum = 0;
        for (int c = 0; c < 32; ++c)
        {
            t.m[k][c] = (int8_t)basis(32, 0, k, c);
            sum += basis(32, 0, k, c);
        }
        t.rowsum128[k] = 128 * sum;
    }
    return t;
}
static __constant__ Dct32I8 c_dct32_i8 = make_dct32_i8();

// acc[reg] = sum_j M[row(reg)][j] * v[j][col] + add for the 32 x 32 int16 operand whose column (lane & 31), K values 16 (lane >> 5) .. + 15 this lane holds as eight packed pairs
__device__ __forceinline__ void mfma_dct32(const i32x4v a, const uint32_t (&v)[8], int add, int half, int (&out)[16])
{
    i32x4v bh, bl;
#pragma unroll
    for (int q = 0; q < 4; ++q)
    {
        bh[q] = (int)__builtin_amdgcn_perm(v[2 * q + 1], v[2 * q], 0x07050301u);                    // the high bytes of four values
        bl[q] = (int)(__builtin_amdgcn_perm(v[2 * q + 1], v[2 * q], 0x06040200u) ^ 0x80808080u);    // the low bytes - 128
    }
    i32x16v ch, cl;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg)
    {
        ch[reg] = 0;
        cl[reg] = add + c_dct32_i8.rowsum128[(reg & 3) + 8 * (reg >> 2) + 4 * half];
    }
    ch = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bh, ch, 0, 0, 0);
    cl = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bl, cl, 0, 0, 0);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) out[reg] = (int)(((uint32_t)ch[reg] << 8) + (uint32_t)cl[reg]);
}

template <bool SCAN>
__global__ __launch_bounds__(64) void k_tu_forward_mfma32(int16_t *__restrict__ coeffs, const char *__restrict__ src, long stride_src, const char *__restrict__ pred,
                                                          long stride_pred, const int32_t *__restrict__ jobs, int njobs, const RdoqJob *__restrict__ rjobs = nullptr,
                                                          int16_t *__restrict__ levels = nullptr, RdoqWork *__restrict__ work = nullptr)
{
    constexpr int LS = 40;      // int16 per row of the transposed intermediate: 80 bytes, so that a lane's 32 bytes are 16-byte aligned
    __shared__ __attribute__((aligned(16))) int16_t lds[32 * LS];
    const int job = xcd_block(blockIdx.x, gridDim.x);
    if (job >= njobs) return;      // (uniform: a wavefront is a block)
    const int32_t *j = jobs + (long)job * 4;
    const int lane = threadIdx.x, y = lane & 31, half = lane >> 5;
    // the basis rows this lane feeds as A: row y, K = 16 half .. + 15
    const i32x4v a = *reinterpret_cast<const i32x4v *>(&c_dct32_i8.m[y][16 * half]);
    // residual of row y, samples 16 half .. + 15, as packed pairs
    uint32_t r[8];
    {
        const u32x4 s4 = ld16(src + (long)j[1] + (long)y * stride_src + 16 * half), p4 = ld16(pred + (long)j[2] + (long)y * stride_pred + 16 * half);
        const uint32_t sw[4] = {s4.x, s4.y, s4.z, s4.w}, pw[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
        for (int q = 0; q < 4; ++q)
        {
            r[2 * q] = pk_sub(__builtin_amdgcn_perm(0u, sw[q], 0x0c010c00u), __builtin_amdgcn_perm(0u, pw[q], 0x0c010c00u));
            r[2 * q + 1] = pk_sub(__builtin_amdgcn_perm(0u, sw[q], 0x0c030c02u), __builtin_amdgcn_perm(0u, pw[q], 0x0c030c02u));
        }
    }
    int t[16];
    mfma_dct32(a, r, 8, half, t);      // T[k][y], k = (reg & 3) + 8 (reg >> 2) + 4 half
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) lds[((reg & 3) + 8 * (reg >> 2) + 4 * half) * LS + y] = (int16_t)(t[reg] >> 4);
    __syncthreads();
    // second pass: this lane's column r' = y of T^T is row y of T: K = 16 half .. + 15
    uint32_t v[8];
    {
        const u32x4 q0 = *reinterpret_cast<const u32x4 *>(&lds[y * LS + 16 * half]), q1 = *reinterpret_cast<const u32x4 *>(&lds[y * LS + 16 * half + 8]);
        v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w; v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
    }
    int o[16];
    mfma_dct32(a, v, 1024, half, o);   // C[k][y] << 11
    int16_t *c = coeffs + j[0] + y;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) c[((reg & 3) + 8 * (reg >> 2) + 4 * half) * 32] = (int16_t)(o[reg] >> 11);
    if (SCAN)
    {
        // the lane holds column y, rows 4 gy .. 4 gy + 3 of group row gy = 2 q + half in registers 4 q .. 4 q + 3: a group's largest magnitude is the maximum over those
        // four and over the four lanes of the quad
        const RdoqJob rj = rjobs[job];
        uint32_t thr[3];
        rdoqThresholds(rj.quant_scale, rj.quant_shift, thr);
        uint32_t lo = 0, hi = 0;
        uint64_t mask[3] = {0, 0, 0};
#pragma unroll
        for (int q = 0; q < 4; ++q)
        {
            uint32_t big = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
            {
                const int cf = (int16_t)(o[4 * q + k] >> 11);
                big = max(big, (uint32_t)abs(cf));
                const uint32_t sq = (uint32_t)(cf * cf);
                lo += sq & 0xffffu;
                hi += sq >> 16;
            }
            big = max(big, (uint32_t)__shfl_xor((int)big, 1, kWave));
            big = max(big, (uint32_t)__shfl_xor((int)big, 2, kWave));
#pragma unroll
            for (int m = 0; m < 3; ++m)
            {
                const uint64_t b = __ballot((y & 3) == 0 && big >= thr[m]);      // bit 4 gx + 32 half: group (gx, 2 q + half)
                uint32_t f0 = 0, f1 = 0;
#pragma unroll
                for (int gx = 0; gx < 8; ++gx)
                {
                    f0 |= (uint32_t)((b >> (4 * gx)) & 1) << gx;
                    f1 |= (uint32_t)((b >> (32 + 4 * gx)) & 1) << gx;
                }
                mask[m] |= (uint64_t)f0 << (16 * q) | (uint64_t)f1 << (16 * q + 8);
            }
        }
        const int slo = wave_sum((int)lo), shi = wave_sum((int)hi);
        // the level block, zeroed (RDOQ writes only the levels it keeps): 2 048 bytes, 32 per lane
        int16_t *z = levels + rj.dst_off + lane * 16;
        st8(z, u32x2{0, 0}); st8(z + 4, u32x2{0, 0}); st8(z + 8, u32x2{0, 0}); st8(z + 12, u32x2{0, 0});
        if (lane == 0)
        {
            RdoqInfo *info = reinterpret_cast<RdoqInfo *>(reinterpret_cast<char *>(work) + rdoqInfoOffset());
            RdoqInfo w;
            w.mask = mask[0];
            w.mask2 = mask[1];
            w.mask3 = mask[2];
            w.sumSq = ((int64_t)shi << 16) + slo;
            info[job] = w;
        }
    }
}

// diagnostic A/B switch (profiles/): HAVOC_TU_MFMA=0 keeps the 32x32 forward transform of 8-bit content on the vector units
static bool mfmaForward()
{
    static const bool on = !(getenv("HAVOC_TU_MFMA") && atoi(getenv("HAVOC_TU_MFMA")) == 0);
    return on;
}

template <int S, int LOG2, int TR>
static void go_fwd(hipStream_t st, int16_t *co, const char *src, long ss, const char *pred, long sp, const int32_t *j, int n, int bd)
{
    constexpr int TPW = 64 >> LOG2;
    hipLaunchKernelGGL((k_tu_forward<S, LOG2, TR>), dim3((n + TPW - 1) / TPW), dim3(64), 0, st, co, src, ss, pred, sp, j, n, bd);
}

template <int S>
static hipError_t launch_fwd_s(hipStream_t st, int bd, int log2, int tr, int16_t *co, const char *src, long ss, const char *pred, long sp,
                               const int32_t *j, int n)
{
    if (tr)
    {
        if (log2 != 2) return hipErrorInvalidValue;
        go_fwd<S, 2, 1>(st, co, src, ss, pred, sp, j, n, bd);
    }
    else
        switch (log2)
        {
        case 2: go_fwd<S, 2, 0>(st, co, src, ss, pred, sp, j, n, bd); break;
        case 3: go_fwd<S, 3, 0>(st, co, src, ss, pred, sp, j, n, bd); break;
        case 4: go_fwd<S, 4, 0>(st, co, src, ss, pred, sp, j, n, bd); break;
        case 5:
            if (S == 1 && bd == 8 && mfmaForward()) hipLaunchKernelGGL((k_tu_forward_mfma32<false>), dim3(n), dim3(64), 0, st, co, src, ss, pred, sp, j, n);
            else go_fwd<S, 5, 0>(st, co, src, ss, pred, sp, j, n, bd);
            break;
        default: return hipErrorInvalidValue;
        }
    return hipGetLastError();
}

// tu_forward with the RDOQ scan folded in (16x16 / 32x32 DCT blocks)
hipError_t launch_tu_forward_scan(hipStream_t st, int S, int bd, int log2, int16_t *coeffs, const void *src, long ss, const void *pred, long sp, const void *jobs,
                                  int n, const void *rdoq_jobs, int16_t *levels, void *workspace)
{
    if (n <= 0) return hipSuccess;
    if (log2 != 4 && log2 != 5) return hipErrorInvalidValue;
    const char *s8 = (const char *)src, *p8 = (const char *)pred;
    const int32_t *j = (const int32_t *)jobs;
    const RdoqJob *rj = (const RdoqJob *)rdoq_jobs;
    RdoqWork *w = (RdoqWork *)workspace;
    const int tpw = 64 >> log2;
    const dim3 g((n + tpw - 1) / tpw), b(64);
    if (S == 1 && log2 == 4) hipLaunchKernelGGL((k_tu_forward<1, 4, 0, true>), g, b, 0, st, coeffs, s8, ss, p8, sp, j, n, bd, rj, levels, w);
    else if (S == 1 && bd == 8 && mfmaForward()) hipLaunchKernelGGL((k_tu_forward_mfma32<true>), dim3(n), b, 0, st, coeffs, s8, ss, p8, sp, j, n, rj, levels, w);
    else if (S == 1) hipLaunchKernelGGL((k_tu_forward<1, 5, 0, true>), g, b, 0, st, coeffs, s8, ss, p8, sp, j, n, bd, rj, levels, w);
    else if (log2 == 4) hipLaunchKernelGGL((k_tu_forward<2, 4, 0, true>), g, b, 0, st, coeffs, s8, ss, p8, sp, j, n, bd, rj, levels, w);
    else hipLaunchKernelGGL((k_tu_forward<2, 5, 0, true>), g, b, 0, st, coeffs, s8, ss, p8, sp, j, n, bd, rj, levels, w);
    return hipGetLastError();
}

hipError_t launch_tu_forward(hipStream_t st, int S, int bd, int log2, int tr, int16_t *coeffs, const void *src, long ss, const void *pred, long sp,
                             const void *jobs, int n)
{
    if (n <= 0) return hipSuccess;
    return S == 1 ? launch_fwd_s<1>(st, bd, log2, tr, coeffs, (const char *)src, ss, (const char *)pred, sp, (const int32_t *)jobs, n)
                  : launch_fwd_s<2>(st, bd, log2, tr, coeffs, (const char *)src, ss, (const char *)pred, sp, (const int32_t *)jobs, n);
}

template <int S, int LOG2, int TR>
static void go_rec(hipStream_t st, char *rec, long sr, const char *pred, long sp, const char *src, long ss, const int16_t *lv, const int32_t *j, int n,
                   int bd, int scale, int shift, uint32_t *ssd)
{
    constexpr int TPW = 64 >> LOG2;
    hipLaunchKernelGGL((k_tu_reconstruct<S, LOG2, TR>), dim3((n + TPW - 1) / TPW), dim3(64), 0, st, rec, sr, pred, sp, src, ss, lv, j, n, bd, scale,
                       shift, ssd);
}

template <int S>
static hipError_t launch_rec_s(hipStream_t st, int bd, int log2, int tr, int scale, int shift, char *rec, long sr, const char *pred, long sp,
                               const char *src, long ss, const int16_t *lv, const int32_t *j, int n, uint32_t *ssd)
{
    if (tr)
    {
        if (log2 != 2) return hipErrorInvalidValue;
        go_rec<S, 2, 1>(st, rec, sr, pred, sp, src, ss, lv, j, n, bd, scale, shift, ssd);
    }
    else
        switch (log2)
        {
        case 2: go_rec<S, 2, 0>(st, rec, sr, pred, sp, src, ss, lv, j, n, bd, scale, shift, ssd); break;
        case 3: go_rec<S, 3, 0>(st, rec, sr, pred, sp, src, ss, lv, j, n, bd, scale, shift, ssd); break;
        case 4: go_rec<S, 4, 0>(st, rec, sr, pred, sp, src, ss, lv, j, n, bd, scale, shift, ssd); break;
        case 5: go_rec<S, 5, 0>(st, rec, sr, pred, sp, src, ss, lv, j, n, bd, scale, shift, ssd); break;
        default: return hipErrorInvalidValue;
        }
    return hipGetLastError();
}

hipError_t launch_tu_reconstruct(hipStream_t st, int S, int bd, int log2, int tr, int scale, int shift, void *rec, long sr, const void *pred, long sp,
                                 const void *src, long ss, const int16_t *levels, const void *jobs, int n, uint32_t *ssd)
{
    if (n <= 0) return hipSuccess;
    return S == 1 ? launch_rec_s<1>(st, bd, log2, tr, scale, shift, (char *)rec, sr, (const char *)pred, sp, (const char *)src, ss, levels,
                                    (const int32_t *)jobs, n, ssd)
                  : launch_rec_s<2>(st, bd, log2, tr, scale, shift, (char *)rec, sr, (const char *)pred, sp, (const char *)src, ss, levels,
                                    (const int32_t *)jobs, n, ssd);
}

} // namespace havoc_gpu
