// Distortion metrics: SAD, 4-way SAD, SSD, Hadamard SATD, linear SSD.
//
// Work mapping (all kernels): one wavefront (64 lanes) per job, four jobs per 256-thread workgroup, so a launch
// of N jobs is N/4 workgroups (>> 256 CUs for any real batch).  Each lane loads 4/8/16-byte row chunks straight
// from HBM/L2 (rows of a block are contiguous, candidate positions are arbitrary, hence the unaligned vector
// loads), accumulates with the packed byte/word SAD and dot instructions, and the 64 partial sums are folded
// with DPP row shifts/broadcasts -- integer arithmetic only, no LDS, no MFMA.
#include "common.h"

namespace havoc_gpu {

// ---------------------------------------------------------------------------------------------------------
// SAD  (reference: havoc/sad.cpp:432-449 single, :513-542 four-way; 16-bit results >> 2)
// ---------------------------------------------------------------------------------------------------------

template <int S>
__device__ __forceinline__ uint32_t sad_dword(uint32_t a, uint32_t b, uint32_t acc)
{
    if (S == 1) return __builtin_amdgcn_sad_u8(a, b, acc);   // 4 samples
    return __builtin_amdgcn_sad_u16(a, b, acc);              // 2 samples
}

// accumulate |src - ref_k| over a w x h block.  CB = bytes per lane chunk (4, 8, 16); rowBytes % CB == 0.
template <int S, int WAYS, int CB>
__device__ __forceinline__ void sad_block(const char *src, long ssb, const char *const (&ref)[WAYS], long rsb, int rowBytes, int h,
                                          int lane, uint32_t (&acc)[WAYS])
{
    const int cpr = rowBytes / CB;      // chunks per row: 1, 2, 3, 4, 6 or 8
    const int rpi = kWave / cpr;        // rows per iteration
    const int y0 = lane / cpr;
    const int xb = (lane - y0 * cpr) * CB;
    if (y0 >= rpi) return;              // lanes beyond rpi*cpr idle (cpr = 3, 6)
    for (int y = y0; y < h; y += rpi)
    {
        const char *s = src + y * ssb + xb;
        if (CB == 4)
        {
            const uint32_t a = ld4(s);
#pragma unroll
            for (int k = 0; k < WAYS; ++k) acc[k] = sad_dword<S>(a, ld4(ref[k] + y * rsb + xb), acc[k]);
        }
        else if (CB == 8)
        {
            const u32x2 a = ld8(s);
#pragma unroll
            for (int k = 0; k < WAYS; ++k)
            {
                const u32x2 b = ld8(ref[k] + y * rsb + xb);
                acc[k] = sad_dword<S>(a.x, b.x, acc[k]);
                acc[k] = sad_dword<S>(a.y, b.y, acc[k]);
            }
        }
        else
        {
            const u32x4 a = ld16(s);
#pragma unroll
            for (int k = 0; k < WAYS; ++k)
            {
                const u32x4 b = ld16(ref[k] + y * rsb + xb);
                acc[k] = sad_dword<S>(a.x, b.x, acc[k]);
                acc[k] = sad_dword<S>(a.y, b.y, acc[k]);
                acc[k] = sad_dword<S>(a.z, b.z, acc[k]);
                acc[k] = sad_dword<S>(a.w, b.w, acc[k]);
            }
        }
    }
}

template <int S, int WAYS>
__global__ __launch_bounds__(256) void k_sad(const char *__restrict__ src, long stride_src, const char *__restrict__ ref, long stride_ref,
                                             const int32_t *__restrict__ jobs, int njobs, int32_t *__restrict__ out)
{
    typedef typename Sample<S>::T T;
    const int job = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (job >= njobs) return;
    int so, w, h;
    int ro[WAYS];
    if (WAYS == 1)
    {
        const int32_t *j = jobs + job * 4;   // havoc_mi355x_pair_job
        so = j[0]; ro[0] = j[1]; w = j[2]; h = j[3];
    }
    else
    {
        const int32_t *j = jobs + job * 8;   // havoc_mi355x_sad4_job
        so = j[0];
#pragma unroll
        for (int k = 0; k < WAYS; ++k) ro[k] = j[1 + k];
        w = j[5]; h = j[6];
    }
    const long ssb = stride_src * S, rsb = stride_ref * S;
    const char *s = src + (long)so * S;
    const char *r[WAYS];
#pragma unroll
    for (int k = 0; k < WAYS; ++k) r[k] = ref + (long)ro[k] * S;
    uint32_t acc[WAYS];
#pragma unroll
    for (int k = 0; k < WAYS; ++k) acc[k] = 0;

    const int rowBytes = w * S;
    if ((rowBytes & 15) == 0) sad_block<S, WAYS, 16>(s, ssb, r, rsb, rowBytes, h, lane, acc);
    else if ((rowBytes & 7) == 0) sad_block<S, WAYS, 8>(s, ssb, r, rsb, rowBytes, h, lane, acc);
    else if ((rowBytes & 3) == 0) sad_block<S, WAYS, 4>(s, ssb, r, rsb, rowBytes, h, lane, acc);
    else
    {
        // generic widths (the reference's sadGeneric entry): one sample per lane per step
        const FastDiv fd(w);
        for (int i = lane; i < w * h; i += kWave)
        {
            const int y = fd.div(i), x = i - y * w;
            const int a = reinterpret_cast<const T *>(s + y * ssb)[x];
#pragma unroll
            for (int k = 0; k < WAYS; ++k) acc[k] += abs(a - (int)reinterpret_cast<const T *>(r[k] + y * rsb)[x]);
        }
    }
#pragma unroll
    for (int k = 0; k < WAYS; ++k)
    {
        int t = wave_sum((int)acc[k]);
        if (S == 2) t >>= 2;
        if (lane == 0) out[job * WAYS + k] = t;
    }
}

// ---------------------------------------------------------------------------------------------------------
// SSD  (reference: havoc/ssd.cpp:28-43; uint32 accumulation, 16-bit result >> 4)
// sum (a-b)^2 = sum a*a + sum b*b - 2 sum a*b (mod 2^32), three unsigned dot instructions per dword
// ---------------------------------------------------------------------------------------------------------

template <int S>
__device__ __forceinline__ uint32_t ssd_dword(uint32_t a, uint32_t b, uint32_t acc)
{
    if (S == 1)
    {
        acc = __builtin_amdgcn_udot4(a, a, acc, false);
        acc = __builtin_amdgcn_udot4(b, b, acc, false);
        const uint32_t ab = __builtin_amdgcn_udot4(a, b, 0u, false);
        return acc - 2u * ab;
    }
    const u16x2 va = __builtin_bit_cast(u16x2, a), vb = __builtin_bit_cast(u16x2, b);
    acc = __builtin_amdgcn_udot2(va, va, acc, false);
    acc = __builtin_amdgcn_udot2(vb, vb, acc, false);
    const uint32_t ab = __builtin_amdgcn_udot2(va, vb, 0u, false);
    return acc - 2u * ab;
}

template <int S>
__global__ __launch_bounds__(256) void k_ssd(const char *__restrict__ pa, long stride_a, const char *__restrict__ pb, long stride_b,
                                             const int32_t *__restrict__ jobs, int njobs, uint32_t *__restrict__ out)
{
    typedef typename Sample<S>::T T;
    const int job = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (job >= njobs) return;
    const int32_t *j = jobs + job * 4;
    const int w = j[2], h = j[3];
    const long sab = stride_a * S, sbb = stride_b * S;
    const char *a = pa + (long)j[0] * S;
    const char *b = pb + (long)j[1] * S;
    uint32_t acc = 0;
    const int rowBytes = w * S;
    if ((rowBytes & 3) == 0)
    {
        const int cpr = rowBytes >> 2;          // dwords per row (1..32)
        const FastDiv fd(cpr);
        for (int i = lane; i < cpr * h; i += kWave)
        {
            const int y = fd.div(i), x = (i - y * cpr) * 4;
            acc = ssd_dword<S>(ld4(a + y * sab + x), ld4(b + y * sbb + x), acc);
        }
    }
    else
    {
        const FastDiv fd(w);
        for (int i = lane; i < w * h; i += kWave)
        {
            const int y = fd.div(i), x = i - y * w;
            const int d = (int)reinterpret_cast<const T *>(a + y * sab)[x] - (int)reinterpret_cast<const T *>(b + y * sbb)[x];
            acc += (uint32_t)(d * d);
        }
    }
    uint32_t t = (uint32_t)wave_sum((int)acc);
    if (S == 2) t >>= 4;
    if (lane == 0) out[job] = t;
}

// ---------------------------------------------------------------------------------------------------------
// Hadamard SATD  (reference: havoc/hadamard.cpp:58-98; PU tiling: turing/Measure.h:97-135)
// One lane owns one n x n tile: it loads the tile's rows of both operands, takes differences, runs the 2-D
// butterfly network in registers and sums |coefficients|.  A wave covers up to 64 tiles of its job per step.
// ---------------------------------------------------------------------------------------------------------

template <int N>
__device__ __forceinline__ void wht(int (&v)[N])
{
#pragma unroll
    for (int len = 1; len < N; len <<= 1)
#pragma unroll
        for (int i = 0; i < N; i += len << 1)
#pragma unroll
            for (int k = i; k < i + len; ++k)
            {
                const int a = v[k], b = v[k + len];
                v[k] = a + b;
                v[k + len] = a - b;
            }
}

template <int S, int N>
__device__ __forceinline__ void load_diff_row(const char *a, const char *b, int (&d)[N])
{
    if (S == 1)
    {
        if (N == 8)
        {
            const u32x2 va = ld8(a), vb = ld8(b);
#pragma unroll
            for (int x = 0; x < 4; ++x)
            {
                d[x] = (int)((va.x >> (8 * x)) & 0xff) - (int)((vb.x >> (8 * x)) & 0xff);
                d[x + 4] = (int)((va.y >> (8 * x)) & 0xff) - (int)((vb.y >> (8 * x)) & 0xff);
            }
        }
        else if (N == 4)
        {
            const uint32_t va = ld4(a), vb = ld4(b);
#pragma unroll
            for (int x = 0; x < 4; ++x) d[x] = (int)((va >> (8 * x)) & 0xff) - (int)((vb >> (8 * x)) & 0xff);
        }
        else
        {
#pragma unroll
            for (int x = 0; x < N; ++x) d[x] = (int)(uint8_t)a[x] - (int)(uint8_t)b[x];
        }
    }
    else
    {
        if (N == 8)
        {
            const u32x4 va = ld16(a), vb = ld16(b);
            const uint32_t wa[4] = {va.x, va.y, va.z, va.w}, wb[4] = {vb.x, vb.y, vb.z, vb.w};
#pragma unroll
            for (int x = 0; x < 4; ++x)
            {
                d[2 * x] = (int)(wa[x] & 0xffff) - (int)(wb[x] & 0xffff);
                d[2 * x + 1] = (int)(wa[x] >> 16) - (int)(wb[x] >> 16);
            }
        }
        else if (N == 4)
        {
            const u32x2 va = ld8(a), vb = ld8(b);
            d[0] = (int)(va.x & 0xffff) - (int)(vb.x & 0xffff);
            d[1] = (int)(va.x >> 16) - (int)(vb.x >> 16);
            d[2] = (int)(va.y & 0xffff) - (int)(vb.y & 0xffff);
            d[3] = (int)(va.y >> 16) - (int)(vb.y >> 16);
        }
        else
        {
            const uint32_t va = ld4(a), vb = ld4(b);
            d[0] = (int)(va & 0xffff) - (int)(vb & 0xffff);
            d[1] = (int)(va >> 16) - (int)(vb >> 16);
        }
    }
}

// SATD of one N x N tile, normalised and scaled exactly like compute_satd_c_ref<N>
template <int S, int N>
__device__ __forceinline__ int satd_tile(const char *a, long sab, const char *b, long sbb)
{
    int m[N][N];
#pragma unroll
    for (int y = 0; y < N; ++y)
    {
        load_diff_row<S, N>(a + y * sab, b + y * sbb, m[y]);
        wht<N>(m[y]);
    }
    int sum = N / 4;
#pragma unroll
    for (int x = 0; x < N; ++x)
    {
        int col[N];
#pragma unroll
        for (int y = 0; y < N; ++y) col[y] = m[y][x];
        wht<N>(col);
#pragma unroll
        for (int y = 0; y < N; ++y) sum += abs(col[y]);
    }
    sum /= N / 2;       // sum >= 0: a shift
    return S == 2 ? sum >> 2 : sum;
}

template <int S, int N>
__device__ __forceinline__ int satd_tiles(const char *a, long sab, const char *b, long sbb, int w, int h, int lane)
{
    const int tw = w / N, th = h / N;
    const FastDiv fd(tw);
    int acc = 0;
    for (int t = lane; t < tw * th; t += kWave)
    {
        const int ty = fd.div(t), tx = t - ty * tw;
        acc += satd_tile<S, N>(a + (long)ty * N * sab + tx * N * S, sab, b + (long)ty * N * sbb + tx * N * S, sbb);
    }
    return acc;
}

template <int S>
__global__ __launch_bounds__(256) void k_satd(const char *__restrict__ pa, long stride_a, const char *__restrict__ pb, long stride_b,
                                              const int32_t *__restrict__ jobs, int njobs, int32_t *__restrict__ out)
{
    const int job = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (job >= njobs) return;
    const int32_t *j = jobs + job * 4;
    const int w = j[2], h = j[3];
    const long sab = stride_a * S, sbb = stride_b * S;
    const char *a = pa + (long)j[0] * S;
    const char *b = pb + (long)j[1] * S;
    int acc;
    if ((w | h) & 3) acc = satd_tiles<S, 2>(a, sab, b, sbb, w, h, lane);        // turing/Measure.h:100-111
    else if ((w | h) & 7) acc = satd_tiles<S, 4>(a, sab, b, sbb, w, h, lane);   // :112-122
    else acc = satd_tiles<S, 8>(a, sab, b, sbb, w, h, lane);                    // :123-133
    const int t = wave_sum(acc);
    if (lane == 0) out[job] = t;
}

// ---------------------------------------------------------------------------------------------------------
// linear SSD over an 8-bit run (reference: havoc/diff.cpp:29-39) -- PSNR tool only; grid-stride + one atomic
// per wave.  *out must be zeroed by the caller (the host entry point does it).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ssd_linear(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, int n, int32_t *__restrict__ out)
{
    uint32_t acc = 0;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int nth = gridDim.x * blockDim.x;
    const int n4 = n >> 2;
    for (int i = gid; i < n4; i += nth) acc = ssd_dword<1>(ld4(a + 4 * i), ld4(b + 4 * i), acc);
    for (int i = n4 * 4 + gid; i < n; i += nth)
    {
        const int d = (int)a[i] - (int)b[i];
        acc += (uint32_t)(d * d);
    }
    const int t = wave_sum((int)acc);
    if ((threadIdx.x & 63) == 0 && t) atomicAdd(out, t);
}

// ---------------------------------------------------------------------------------------------------------
// launchers (called from api.hip)
// ---------------------------------------------------------------------------------------------------------

#define LAUNCH4(kernel, njobs, stream, ...) \
    hipLaunchKernelGGL(kernel, dim3(((njobs) + 3) / 4), dim3(256), 0, stream, __VA_ARGS__)

hipError_t launch_sad(hipStream_t st, int S, int ways, const void *src, long ss, const void *ref, long rs, const void *jobs, int n, int32_t *out)
{
    if (n <= 0) return hipSuccess;
    const char *s = (const char *)src, *r = (const char *)ref;
    const int32_t *j = (const int32_t *)jobs;
    if (S == 1 && ways == 1) LAUNCH4((k_sad<1, 1>), n, st, s, ss, r, rs, j, n, out);
    else if (S == 2 && ways == 1) LAUNCH4((k_sad<2, 1>), n, st, s, ss, r, rs, j, n, out);
    else if (S == 1 && ways == 4) LAUNCH4((k_sad<1, 4>), n, st, s, ss, r, rs, j, n, out);
    else if (S == 2 && ways == 4) LAUNCH4((k_sad<2, 4>), n, st, s, ss, r, rs, j, n, out);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_ssd(hipStream_t st, int S, const void *a, long sa, const void *b, long sb, const void *jobs, int n, uint32_t *out)
{
    if (n <= 0) return hipSuccess;
    if (S == 1) LAUNCH4((k_ssd<1>), n, st, (const char *)a, sa, (const char *)b, sb, (const int32_t *)jobs, n, out);
    else LAUNCH4((k_ssd<2>), n, st, (const char *)a, sa, (const char *)b, sb, (const int32_t *)jobs, n, out);
    return hipGetLastError();
}

hipError_t launch_satd(hipStream_t st, int S, const void *a, long sa, const void *b, long sb, const void *jobs, int n, int32_t *out)
{
    if (n <= 0) return hipSuccess;
    if (S == 1) LAUNCH4((k_satd<1>), n, st, (const char *)a, sa, (const char *)b, sb, (const int32_t *)jobs, n, out);
    else LAUNCH4((k_satd<2>), n, st, (const char *)a, sa, (const char *)b, sb, (const int32_t *)jobs, n, out);
    return hipGetLastError();
}

hipError_t launch_ssd_linear(hipStream_t st, const uint8_t *a, const uint8_t *b, int n, int32_t *out)
{
    hipError_t e = hipMemsetAsync(out, 0, sizeof(int32_t), st);
    if (e != hipSuccess || n <= 0) return e;
    const int blocks = min(2048, (n / 4 + 255) / 256 + 1);
    hipLaunchKernelGGL(k_ssd_linear, dim3(blocks), dim3(256), 0, st, a, b, n, out);
    return hipGetLastError();
}

} // namespace havoc_gpu
