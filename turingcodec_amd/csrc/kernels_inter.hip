// Inter prediction: HEVC 8-tap (luma) / 4-tap (chroma) fractional-sample interpolation, uni and bi, and the
// bi-search helper SubtractBi.
//
// Work mapping: one 64-lane workgroup per prediction block.  The (w+taps-1) x (h+taps-1) reference window is staged
// from HBM into LDS with 4-sample unaligned vector loads (rows are contiguous in the padded plane), the horizontal
// pass writes 16-bit intermediates back to LDS, the vertical pass reads them column-wise and writes the block with
// packed stores.  All arithmetic is int32 on int16/uint16 operands, exactly as the reference's generic C function
// (havoc/pred_inter.cpp:76-110), whose intermediates provably fit 16 bits for bit depths 8..10.
#include "common.h"

namespace havoc_gpu {

__constant__ int8_t c_luma[4][8] = {
    {0, 0, 0, 64, 0, 0, 0, 0}, {-1, 4, -10, 58, 17, -5, 1, 0}, {-1, 4, -11, 40, 40, -11, 4, -1}, {0, 1, -5, 17, 58, -10, 4, -1}};
__constant__ int8_t c_chroma[8][4] = {{0, 64, 0, 0},   {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4},
                                      {-4, 36, 36, -4}, {-4, 28, 46, -6}, {-2, 16, 54, -4}, {-2, 10, 58, -2}};

template <int TAPS>
__device__ __forceinline__ void load_taps(int frac, int (&c)[TAPS])
{
#pragma unroll
    for (int k = 0; k < TAPS; ++k) c[k] = TAPS == 8 ? (int)c_luma[frac][k] : (int)c_chroma[frac][k];
}

template <int TAPS> struct Geo
{
    static constexpr int kAbove = TAPS / 2 - 1;          // rows above / columns left of the block
    static constexpr int kMaxWin = 64 + TAPS - 1;        // window rows / columns for a 64 x 64 block
    static constexpr int kWinStride = 64 + TAPS;         // LDS row stride of the window (elements), even
};

// Stage the window of block (w x h) at `ref` into LDS as uint16.  4 samples per lane per step.
template <int S, int TAPS>
__device__ __forceinline__ void stage_window(uint16_t *win, const char *ref, long rsb, int w, int h, int lane)
{
    typedef Geo<TAPS> G;
    const int ww = w + TAPS - 1, wh = h + TAPS - 1;
    const int cpr = (ww + 3) >> 2;  // 4-sample chunks per row (the last chunk may read <= 3 samples past the window)
    const FastDiv fd(cpr);
    const char *base = ref - G::kAbove * rsb - G::kAbove * S;
    for (int i = lane; i < cpr * wh; i += kWave)
    {
        const int y = fd.div(i), x = (i - y * cpr) * 4;
        uint16_t *d = win + y * G::kWinStride + x;
        const char *p = base + y * rsb + x * S;
        if (S == 1)
        {
            const uint32_t v = ld4(p);
            d[0] = v & 0xff; d[1] = (v >> 8) & 0xff; d[2] = (v >> 16) & 0xff; d[3] = v >> 24;
        }
        else
        {
            const u32x2 v = ld8(p);
            d[0] = v.x & 0xffff; d[1] = v.x >> 16; d[2] = v.y & 0xffff; d[3] = v.y >> 16;
        }
    }
}

// horizontal pass over all window rows: tmp[y][x] = (sum_k c[k] * win[y][x + k]) >> shift1   (no rounding)
template <int TAPS>
__device__ __forceinline__ void hpass(int16_t *tmp, const uint16_t *win, int w, int h, int xFrac, int shift1, int lane)
{
    typedef Geo<TAPS> G;
    int c[TAPS];
    load_taps<TAPS>(xFrac, c);
    const int wh = h + TAPS - 1;
    const FastDiv fd(w);
    for (int i = lane; i < w * wh; i += kWave)
    {
        const int y = fd.div(i), x = i - y * w;
        const uint16_t *p = win + y * G::kWinStride + x;
        int a = 0;
#pragma unroll
        for (int k = 0; k < TAPS; ++k) a += c[k] * (int)p[k];
        tmp[y * 64 + x] = (int16_t)(a >> shift1);
    }
}

template <int S>
__device__ __forceinline__ void put(char *dst, long dsb, int x, int y, int v)
{
    if (S == 1) reinterpret_cast<uint8_t *>(dst + y * dsb)[x] = (uint8_t)v;
    else reinterpret_cast<uint16_t *>(dst + y * dsb)[x] = (uint16_t)v;
}

// HavocPredUni (havoc/pred_inter.h:35; C reference havoc/pred_inter.cpp:113-202)
template <int S, int TAPS>
__global__ __launch_bounds__(64) void k_pred_uni(char *__restrict__ dst, long stride_dst, const char *__restrict__ ref, long stride_ref,
                                                 const int32_t *__restrict__ jobs, int bitDepth)
{
    typedef Geo<TAPS> G;
    __shared__ uint16_t win[G::kMaxWin * G::kWinStride];
    __shared__ int16_t tmp[G::kMaxWin * 64];
    const int32_t *j = jobs + blockIdx.x * 8;   // havoc_mi355x_pred_uni_job
    const int lane = threadIdx.x;
    const int w = j[2], h = j[3], xFrac = j[4], yFrac = j[5];
    const long dsb = stride_dst * S, rsb = stride_ref * S;
    char *d = dst + (long)j[0] * S;
    const char *r = ref + (long)j[1] * S;
    const int maxv = (1 << bitDepth) - 1;
    const FastDiv fd(w);

    if (!xFrac && !yFrac)
    {   // havoc_pred_uni_copy_block (pred_inter.cpp:113-124)
        for (int i = lane; i < w * h; i += kWave)
        {
            const int y = fd.div(i), x = i - y * w;
            if (S == 1) reinterpret_cast<uint8_t *>(d + y * dsb)[x] = reinterpret_cast<const uint8_t *>(r + y * rsb)[x];
            else reinterpret_cast<uint16_t *>(d + y * dsb)[x] = reinterpret_cast<const uint16_t *>(r + y * rsb)[x];
        }
        return;
    }
    stage_window<S, TAPS>(win, r, rsb, w, h, lane);
    __syncthreads();
    if (xFrac && yFrac)
    {   // *_hv (pred_inter.cpp:146-163, :185-202)
        const int shift1 = min(4, bitDepth - 8);
        const int shift = 6 + max(2, 14 - bitDepth);
        hpass<TAPS>(tmp, win, w, h, xFrac, shift1, lane);
        __syncthreads();
        int c[TAPS];
        load_taps<TAPS>(yFrac, c);
        for (int i = lane; i < w * h; i += kWave)
        {
            const int y = fd.div(i), x = i - y * w;
            int a = 1 << (shift - 1);
#pragma unroll
            for (int k = 0; k < TAPS; ++k) a += c[k] * (int)tmp[(y + k) * 64 + x];
            put<S>(d, dsb, x, y, clip3(0, maxv, a >> shift));
        }
        return;
    }
    // *_h / *_v (pred_inter.cpp:127-143, :166-182): one pass, rounding 32, shift 6
    int c[TAPS];
    load_taps<TAPS>(xFrac ? xFrac : yFrac, c);
    const int step = xFrac ? 1 : G::kWinStride;
    const int origin = xFrac ? G::kAbove * G::kWinStride : G::kAbove;   // skip the unused rows / columns
    for (int i = lane; i < w * h; i += kWave)
    {
        const int y = fd.div(i), x = i - y * w;
        const uint16_t *p = win + origin + y * G::kWinStride + x;
        int a = 32;
#pragma unroll
        for (int k = 0; k < TAPS; ++k) a += c[k] * (int)p[k * step];
        put<S>(d, dsb, x, y, clip3(0, maxv, a >> 6));
    }
}

// HavocPredBi (havoc/pred_inter.h:63; C reference havoc/pred_inter.cpp:1207-1252): both references always take the
// two-pass route (frac 0 is the {..,64,..} filter), 14-bit intermediates, rounded mean, clip.
template <int S, int TAPS>
__global__ __launch_bounds__(64) void k_pred_bi(char *__restrict__ dst, long stride_dst, const char *__restrict__ ref, long stride_ref,
                                                const int32_t *__restrict__ jobs, int bitDepth)
{
    typedef Geo<TAPS> G;
    __shared__ uint16_t win[G::kMaxWin * G::kWinStride];
    __shared__ int16_t tmp[G::kMaxWin * 64];
    __shared__ int16_t first[64 * 64];
    const int32_t *j = jobs + blockIdx.x * 12;   // havoc_mi355x_pred_bi_job
    const int lane = threadIdx.x;
    const int w = j[3], h = j[4];
    const long dsb = stride_dst * S, rsb = stride_ref * S;
    char *d = dst + (long)j[0] * S;
    const int maxv = (1 << bitDepth) - 1;
    const int shift1 = min(4, bitDepth - 8);
    const int shift3 = max(2, 14 - bitDepth);
    const FastDiv fd(w);
#pragma unroll 1
    for (int r = 0; r < 2; ++r)
    {
        const char *p = ref + (long)j[1 + r] * S;
        const int xFrac = j[5 + 2 * r], yFrac = j[6 + 2 * r];
        __syncthreads();
        stage_window<S, TAPS>(win, p, rsb, w, h, lane);
        __syncthreads();
        hpass<TAPS>(tmp, win, w, h, xFrac, shift1, lane);
        __syncthreads();
        int c[TAPS];
        load_taps<TAPS>(yFrac, c);
        for (int i = lane; i < w * h; i += kWave)
        {
            const int y = fd.div(i), x = i - y * w;
            int a = 0;
#pragma unroll
            for (int k = 0; k < TAPS; ++k) a += c[k] * (int)tmp[(y + k) * 64 + x];
            a >>= 6;
            if (r == 0) first[y * 64 + x] = (int16_t)a;
            else put<S>(d, dsb, x, y, clip3(0, maxv, ((int)first[y * 64 + x] + a + (1 << shift3)) >> (shift3 + 1)));
        }
    }
}

// havoc::SubtractBi (havoc/pred_inter.h:87; havoc/pred_inter.cpp:2063-2080): dst = clip(2*src - pred)
template <int S>
__global__ __launch_bounds__(256) void k_subtract_bi(char *__restrict__ dst, long stride_dst, const char *__restrict__ pred, long stride_pred,
                                                     const char *__restrict__ src, long stride_src, const int32_t *__restrict__ jobs, int njobs,
                                                     int bitDepth)
{
    typedef typename Sample<S>::T T;
    const int job = blockIdx.x * 4 + (threadIdx.x >> 6);
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

hipError_t launch_pred_uni(hipStream_t st, int S, int taps, int bitDepth, void *dst, long sd, const void *ref, long sr, const void *jobs, int n)
{
    if (n <= 0) return hipSuccess;
    const int32_t *j = (const int32_t *)jobs;
    char *d = (char *)dst;
    const char *r = (const char *)ref;
    if (S == 1 && taps == 8) hipLaunchKernelGGL((k_pred_uni<1, 8>), dim3(n), dim3(64), 0, st, d, sd, r, sr, j, bitDepth);
    else if (S == 1 && taps == 4) hipLaunchKernelGGL((k_pred_uni<1, 4>), dim3(n), dim3(64), 0, st, d, sd, r, sr, j, bitDepth);
    else if (S == 2 && taps == 8) hipLaunchKernelGGL((k_pred_uni<2, 8>), dim3(n), dim3(64), 0, st, d, sd, r, sr, j, bitDepth);
    else if (S == 2 && taps == 4) hipLaunchKernelGGL((k_pred_uni<2, 4>), dim3(n), dim3(64), 0, st, d, sd, r, sr, j, bitDepth);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_pred_bi(hipStream_t st, int S, int taps, int bitDepth, void *dst, long sd, const void *ref, long sr, const void *jobs, int n)
{
    if (n <= 0) return hipSuccess;
    const int32_t *j = (const int32_t *)jobs;
    char *d = (char *)dst;
    const char *r = (const char *)ref;
    if (S == 1 && taps == 8) hipLaunchKernelGGL((k_pred_bi<1, 8>), dim3(n), dim3(64), 0, st, d, sd, r, sr, j, bitDepth);
    else if (S == 1 && taps == 4) hipLaunchKernelGGL((k_pred_bi<1, 4>), dim3(n), dim3(64), 0, st, d, sd, r, sr, j, bitDepth);
    else if (S == 2 && taps == 8) hipLaunchKernelGGL((k_pred_bi<2, 8>), dim3(n), dim3(64), 0, st, d, sd, r, sr, j, bitDepth);
    else if (S == 2 && taps == 4) hipLaunchKernelGGL((k_pred_bi<2, 4>), dim3(n), dim3(64), 0, st, d, sd, r, sr, j, bitDepth);
    else return hipErrorInvalidValue;
    return hipGetLastError();
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
