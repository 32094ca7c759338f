// Transform-unit primitives: residual, forward DCT/DST, inverse DCT/DST (+ add), quantise, de-quantise,
// quantise-reconstruct.
//
// Transforms: a launch is uniform in (log2TrafoSize, trType) -- the reference selects one function per size through
// get_transform / get_inverse_transform_add (havoc/transform.h:72-84, :128-140).  One lane owns one ROW of a TU and 64/N
// TUs share a wavefront, so a 4x4 launch runs 16 TUs per wave and a 32x32 launch two.  Both 1-D passes are dot
// products of the lane's packed int16 row with a basis row that is identical for every lane: the basis pairs come
// from __constant__ memory through the scalar unit and feed v_dot2c_i32_i16 directly.  The transpose between the two
// passes goes through LDS with a padded row stride (N+2 int16) that keeps both the column writes and the row reads
// bank-conflict free.  Integer arithmetic throughout; the wrap (forward) / saturate (inverse) quirks of the reference
// are reproduced exactly.
#include "common.h"
#include "transform_basis.h"

namespace havoc_gpu {

// havoc::Transform (havoc/transform.h:117; C reference havoc/transform.cpp:3087-3397)
template <int LOG2, int TR>
__global__ __launch_bounds__(64) void k_transform(int16_t *__restrict__ coeffs, const int16_t *__restrict__ res, long stride_res,
                                                  const int32_t *__restrict__ jobs, int njobs, int bitDepth)
{
    constexpr int N = 1 << LOG2, TPW = 64 / N, LS = N + 2;
    __shared__ int16_t lds[TPW][N * LS];
    const int t = threadIdx.x / N, r = threadIdx.x % N;
    const int job = xcd_block(blockIdx.x, gridDim.x) * TPW + t;
    const bool live = job < njobs;
    const int32_t *j = jobs + (live ? job : 0) * 4;   // havoc_mi355x_tu_job
    const int shift1 = LOG2 - 1 + bitDepth - 8, shift2 = LOG2 + 6;

    uint32_t row[N / 2];
    load_row16<N>(res + j[1] + (long)r * stride_res, row);
    int o[N];
    basis_times_row<N, TR, false>(row, 1 << (shift1 - 1), o);
#pragma unroll
    for (int k = 0; k < N; ++k) lds[t][k * LS + r] = (int16_t)(o[k] >> shift1);   // wraps (havoc/transform.cpp:3071-3084)
    __syncthreads();
#pragma unroll
    for (int p = 0; p < N / 2; ++p) row[p] = *reinterpret_cast<const uint32_t *>(&lds[t][r * LS + 2 * p]);
    basis_times_row<N, TR, false>(row, 1 << (shift2 - 1), o);
    if (!live) return;
    int16_t *c = coeffs + j[0] + r;
#pragma unroll
    for (int k = 0; k < N; ++k) c[k * N] = (int16_t)(o[k] >> shift2);
}

// MODE 0: havoc::inverse_transform (int16 residual out, havoc/transform.h:33);
// MODE 1/2: havoc::inverse_transform_add<uint8_t / uint16_t> (havoc/transform.h:61; transform.cpp:358-401)
template <int LOG2, int TR, int MODE>
__global__ __launch_bounds__(64) void k_inverse_transform(char *dst, long stride_dst, const char *pred, long stride_pred,
                                                          int16_t *__restrict__ resout, const int16_t *__restrict__ coeffs,
                                                          const int32_t *__restrict__ jobs, int njobs, int bitDepth)
{
    constexpr int N = 1 << LOG2, TPW = 64 / N, LS = N + 2;
    __shared__ int16_t lds[TPW][N * LS];
    const int t = threadIdx.x / N, r = threadIdx.x % N;
    const int job = xcd_block(blockIdx.x, gridDim.x) * TPW + t;
    const bool live = job < njobs;
    const int32_t *j = jobs + (live ? job : 0) * 4;
    const int shift2 = 20 - bitDepth;

    // column r of the coefficient block, packed in row pairs
    const int16_t *c = coeffs + j[0] + r;
    uint32_t col[N / 2];
#pragma unroll
    for (int p = 0; p < N / 2; ++p)
        col[p] = (uint32_t)(uint16_t)c[(2 * p) * N] | ((uint32_t)(uint16_t)c[(2 * p + 1) * N] << 16);
    int o[N];
    basis_times_row<N, TR, true>(col, 1 << 6, o);
#pragma unroll
    for (int k = 0; k < N; ++k) lds[t][k * LS + r] = (int16_t)clip3(-32768, 32767, o[k] >> 7);
    __syncthreads();
#pragma unroll
    for (int p = 0; p < N / 2; ++p) col[p] = *reinterpret_cast<const uint32_t *>(&lds[t][r * LS + 2 * p]);
    basis_times_row<N, TR, true>(col, 1 << (shift2 - 1), o);
    if (!live) return;
#pragma unroll
    for (int k = 0; k < N; ++k) o[k] = clip3(-32768, 32767, o[k] >> shift2);   // residual row r

    if (MODE == 0)
    {
        int16_t *q = resout + j[1] + r * N;
#pragma unroll
        for (int k = 0; k < N; ++k) q[k] = (int16_t)o[k];
        return;
    }
    const int maxv = (1 << bitDepth) - 1;
    if (MODE == 1)
    {
        const uint8_t *p = reinterpret_cast<const uint8_t *>(pred) + j[2] + (long)r * stride_pred;
        uint8_t *d = reinterpret_cast<uint8_t *>(dst) + j[3] + (long)r * stride_dst;
        uint32_t pk[N / 4];
#pragma unroll
        for (int q = 0; q < N / 4; ++q) pk[q] = ld4(p + 4 * q);     // read the whole row first: pred may alias dst
#pragma unroll
        for (int q = 0; q < N / 4; ++q)
        {
            uint32_t v = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) v |= (uint32_t)clip3(0, maxv, (int)((pk[q] >> (8 * b)) & 0xff) + o[4 * q + b]) << (8 * b);
            st4(d + 4 * q, v);
        }
    }
    else
    {
        const uint16_t *p = reinterpret_cast<const uint16_t *>(pred) + j[2] + (long)r * stride_pred;
        uint16_t *d = reinterpret_cast<uint16_t *>(dst) + j[3] + (long)r * stride_dst;
        uint32_t pk[N / 2];
#pragma unroll
        for (int q = 0; q < N / 2; ++q) pk[q] = ld4(p + 2 * q);
#pragma unroll
        for (int q = 0; q < N / 2; ++q)
        {
            const uint32_t lo = (uint32_t)clip3(0, maxv, (int)(pk[q] & 0xffff) + o[2 * q]);
            const uint32_t hi = (uint32_t)clip3(0, maxv, (int)(pk[q] >> 16) + o[2 * q + 1]);
            st4(d + 2 * q, lo | (hi << 16));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// quantise / de-quantise: one wavefront per job, two int16 per lane per step
// ---------------------------------------------------------------------------------------------------------

// havoc_quantize (havoc/quantize.h:63; C reference havoc/quantize.cpp:278-304)
__global__ __launch_bounds__(256) void k_quantize(int16_t *__restrict__ dst, const int16_t *__restrict__ src, const int32_t *__restrict__ jobs,
                                                  int njobs, int32_t *__restrict__ cbf)
{
    const int job = xcd_block(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (job >= njobs) return;
    const int32_t *j = jobs + job * 8;   // havoc_mi355x_quant_job
    const int n = j[2], scale = j[3], shift = j[4];
    const int offset = j[5] << (shift - 16);
    const int16_t *s = src + j[1];
    int16_t *d = dst + j[0];
    int any = 0;
    for (int i = lane; i < n; i += kWave)
    {
        int x = s[i];
        const int a = (abs(x) * scale + offset) >> shift;
        x = clip3(-32768, 32767, x < 0 ? -a : a);
        any |= x;
        d[i] = (int16_t)x;
    }
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) any |= __shfl_xor(any, o, kWave);
    if (lane == 0) cbf[job] = any;
}

// What a rate estimate reads of a block of quantised levels without downloading them: out[2 * job] = number of non-zero levels,
// out[2 * job + 1] = sum of |level| (the encoder's EstimateRate<residual_coding> walks the levels themselves on the host; a batch client
// that decides between transform-tree candidates wants these per candidate).  jobs: (offset, n) pairs, n a multiple of 2.
__global__ __launch_bounds__(256) void k_level_stats(const int16_t *__restrict__ levels, const int32_t *__restrict__ jobs, int njobs, int32_t *__restrict__ out)
{
    const int job = xcd_block(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (job >= njobs) return;
    const int16_t *s = levels + jobs[2 * job];
    const int n = jobs[2 * job + 1];
    int nz = 0, sum = 0;
    for (int i = 2 * lane; i < n; i += 2 * kWave)
    {
        const uint32_t v = *reinterpret_cast<const uint32_t *>(s + i);
        const int a = (int16_t)(v & 0xffff), b = (int16_t)(v >> 16);
        nz += (a != 0) + (b != 0);
        sum += abs(a) + abs(b);
    }
    nz = wave_sum(nz);
    sum = wave_sum(sum);
    if (lane == 0)
    {
        out[2 * job] = nz;
        out[2 * job + 1] = sum;
    }
}

// havoc_quantize_inverse (havoc/quantize.h:42; C reference havoc/quantize.cpp:37-46)
__global__ __launch_bounds__(256) void k_quantize_inverse(int16_t *__restrict__ dst, const int16_t *__restrict__ src,
                                                          const int32_t *__restrict__ jobs, int njobs)
{
    const int job = xcd_block(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (job >= njobs) return;
    const int32_t *j = jobs + job * 8;
    const int n = j[2], scale = j[3], shift = j[4];
    const int16_t *s = src + j[1];
    int16_t *d = dst + j[0];
    const int add = 1 << (shift - 1);
    for (int i = lane; i < n; i += kWave) d[i] = (int16_t)clip3(-32768, 32767, ((int)s[i] * scale + add) >> shift);
}

// havoc_quantize_reconstruct (havoc/quantize.h:84; C reference havoc/quantize.cpp:538-549), 8-bit
__global__ __launch_bounds__(256) void k_quantize_reconstruct(uint8_t *__restrict__ rec, long stride_rec, const uint8_t *__restrict__ pred,
                                                              long stride_pred, const int16_t *__restrict__ res,
                                                              const int32_t *__restrict__ jobs, int njobs, int log2)
{
    const int job = xcd_block(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (job >= njobs) return;
    const int32_t *j = jobs + job * 4;   // tu_job: [1] res_off, [2] pred_off, [3] dst_off
    const int n = 1 << log2;
    for (int i = lane; i < n * n; i += kWave)
    {
        const int y = i >> log2, x = i & (n - 1);
        rec[j[3] + y * stride_rec + x] = (uint8_t)clip3(0, 255, (int)pred[j[2] + y * stride_pred + x] + (int)res[j[1] + i]);
    }
}

// res = src - pred (turing/Reconstruct.cpp:258-260, 1274-1286)
template <int S>
__global__ __launch_bounds__(256) void k_residual(int16_t *__restrict__ res, long stride_res, const int32_t *__restrict__ res_off,
                                                  const char *__restrict__ src, long stride_src, const char *__restrict__ pred, long stride_pred,
                                                  const int32_t *__restrict__ jobs, int njobs)
{
    typedef typename Sample<S>::T T;
    const int job = xcd_block(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (job >= njobs) return;
    const int32_t *j = jobs + job * 4;
    const int w = j[2], h = j[3];
    const T *s = reinterpret_cast<const T *>(src) + j[0];
    const T *p = reinterpret_cast<const T *>(pred) + j[1];
    int16_t *r = res + res_off[job];
    const FastDiv fd(w);
    for (int i = lane; i < w * h; i += kWave)
    {
        const int y = fd.div(i), x = i - y * w;
        r[y * stride_res + x] = (int16_t)((int)s[y * stride_src + x] - (int)p[y * stride_pred + x]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------

template <int LOG2, int TR>
static void go_fwd(hipStream_t st, int16_t *co, const int16_t *res, long sr, const int32_t *j, int n, int bd)
{
    constexpr int TPW = 64 >> LOG2;
    hipLaunchKernelGGL((k_transform<LOG2, TR>), dim3((n + TPW - 1) / TPW), dim3(64), 0, st, co, res, sr, j, n, bd);
}

hipError_t launch_transform(hipStream_t st, int bitDepth, int log2, int trType, int16_t *coeffs, const int16_t *res, long stride_res,
                            const void *jobs, int n)
{
    if (n <= 0) return hipSuccess;
    const int32_t *j = (const int32_t *)jobs;
    if (trType)
    {
        if (log2 != 2) return hipErrorInvalidValue;
        go_fwd<2, 1>(st, coeffs, res, stride_res, j, n, bitDepth);
    }
    else
        switch (log2)
        {
        case 2: go_fwd<2, 0>(st, coeffs, res, stride_res, j, n, bitDepth); break;
        case 3: go_fwd<3, 0>(st, coeffs, res, stride_res, j, n, bitDepth); break;
        case 4: go_fwd<4, 0>(st, coeffs, res, stride_res, j, n, bitDepth); break;
        case 5: go_fwd<5, 0>(st, coeffs, res, stride_res, j, n, bitDepth); break;
        default: return hipErrorInvalidValue;
        }
    return hipGetLastError();
}

template <int LOG2, int TR>
static void go_inv(hipStream_t st, int mode, char *dst, long sd, const char *pred, long sp, int16_t *resout, const int16_t *co, const int32_t *j,
                   int n, int bd)
{
    constexpr int TPW = 64 >> LOG2;
    const dim3 g((n + TPW - 1) / TPW), b(64);
    if (mode == 0) hipLaunchKernelGGL((k_inverse_transform<LOG2, TR, 0>), g, b, 0, st, dst, sd, pred, sp, resout, co, j, n, bd);
    else if (mode == 1) hipLaunchKernelGGL((k_inverse_transform<LOG2, TR, 1>), g, b, 0, st, dst, sd, pred, sp, resout, co, j, n, bd);
    else hipLaunchKernelGGL((k_inverse_transform<LOG2, TR, 2>), g, b, 0, st, dst, sd, pred, sp, resout, co, j, n, bd);
}

// mode 0: int16 residual to resout; mode 1 / 2: add to 8-bit / 16-bit prediction
hipError_t launch_inverse_transform(hipStream_t st, int mode, int bitDepth, int log2, int trType, void *dst, long sd, const void *pred, long sp,
                                    int16_t *resout, const int16_t *coeffs, const void *jobs, int n)
{
    if (n <= 0) return hipSuccess;
    const int32_t *j = (const int32_t *)jobs;
    char *d = (char *)dst;
    const char *p = (const char *)pred;
    if (trType)
    {
        if (log2 != 2) return hipErrorInvalidValue;
        go_inv<2, 1>(st, mode, d, sd, p, sp, resout, coeffs, j, n, bitDepth);
    }
    else
        switch (log2)
        {
        case 2: go_inv<2, 0>(st, mode, d, sd, p, sp, resout, coeffs, j, n, bitDepth); break;
        case 3: go_inv<3, 0>(st, mode, d, sd, p, sp, resout, coeffs, j, n, bitDepth); break;
        case 4: go_inv<4, 0>(st, mode, d, sd, p, sp, resout, coeffs, j, n, bitDepth); break;
        case 5: go_inv<5, 0>(st, mode, d, sd, p, sp, resout, coeffs, j, n, bitDepth); break;
        default: return hipErrorInvalidValue;
        }
    return hipGetLastError();
}

hipError_t launch_level_stats(hipStream_t st, const int16_t *levels, const void *jobs, int n, int32_t *out)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_level_stats, dim3((n + 3) / 4), dim3(256), 0, st, levels, (const int32_t *)jobs, n, out);
    return hipGetLastError();
}

hipError_t launch_quantize(hipStream_t st, int16_t *dst, const int16_t *src, const void *jobs, int n, int32_t *cbf)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_quantize, dim3((n + 3) / 4), dim3(256), 0, st, dst, src, (const int32_t *)jobs, n, cbf);
    return hipGetLastError();
}

hipError_t launch_quantize_inverse(hipStream_t st, int16_t *dst, const int16_t *src, const void *jobs, int n)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_quantize_inverse, dim3((n + 3) / 4), dim3(256), 0, st, dst, src, (const int32_t *)jobs, n);
    return hipGetLastError();
}

hipError_t launch_quantize_reconstruct(hipStream_t st, int log2, uint8_t *rec, long sr, const uint8_t *pred, long sp, const int16_t *res,
                                       const void *jobs, int n)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_quantize_reconstruct, dim3((n + 3) / 4), dim3(256), 0, st, rec, sr, pred, sp, res, (const int32_t *)jobs, n, log2);
    return hipGetLastError();
}

hipError_t launch_residual(hipStream_t st, int S, int16_t *res, long sres, const int32_t *res_off, const void *src, long ss, const void *pred, long sp,
                           const void *jobs, int n)
{
    if (n <= 0) return hipSuccess;
    if (S == 1)
        hipLaunchKernelGGL((k_residual<1>), dim3((n + 3) / 4), dim3(256), 0, st, res, sres, res_off, (const char *)src, ss, (const char *)pred, sp,
                           (const int32_t *)jobs, n);
    else
        hipLaunchKernelGGL((k_residual<2>), dim3((n + 3) / 4), dim3(256), 0, st, res, sres, res_off, (const char *)src, ss, (const char *)pred, sp,
                           (const int32_t *)jobs, n);
    return hipGetLastError();
}

} // namespace havoc_gpu
