// Full-pel SAD surfaces: for one source block, the SAD against EVERY integer candidate (dx, dy) in [-R, R]^2 around a
// centre of the reference plane -- the super-set that serves havoc_sad / havoc_sad_multiref lookups of the integer
// motion search (reference: havoc/sad.cpp:432-449 per candidate; callers turing/Search.hpp:1447-1482 considerPattern,
// :1585-1623 the bi-prediction grid, :2224-2290 star / raster / refinement patterns).
//
// Mapping.  A workgroup stages the source block(s) and the reference window (block + 2R border) in LDS once; a work
// item is one candidate ROW SEGMENT of 8 consecutive dx at one dy.  8-bit samples use v_qsad_pk_u16_u8: one instruction
// = 4 sliding 4-byte SADs (dx .. dx+3) of an 8-byte window against one source dword, accumulated in four packed 16-bit
// sums which are widened every <= 256 pixels.  Per source dword an item issues 2 quad-SADs and re-uses the window
// dwords across neighbouring chunks: no global traffic and no byte re-alignment per candidate, which is what makes a
// candidate-pixel ~15x cheaper than through the per-candidate SAD4 kernel (measured: 73 T vs 4.9 T candidate-pixels/s).
// 16-bit samples: v_sad_u16 on dword pairs, odd offsets through v_alignbyte.
// A job with a small surface (the 11x11 / 3x3 bi-prediction grids) shares its workgroup with other jobs.
#include "common.h"

namespace havoc_gpu {

constexpr int kSurfThreads = 256;

struct SurfGeom
{
    int R, side, ngp;   // side = 2R+1 candidates per row, ngp = items (groups of 4*NG dx) per row
    int bd;             // dy rows per band (one band per blockIdx.y)
    int jpw;            // jobs per workgroup (> 1 only when bd == side)
    int max_w, max_h;
    int src_dw;         // LDS dwords reserved for one source block
    int pitch;          // LDS window row pitch in dwords (odd)
    int win_dw;         // LDS dwords reserved for one window
};

// ---- 8-bit: CH = source dwords per row (w / 4) -----------------------------------------------------------------

__device__ __forceinline__ uint64_t qsad(uint32_t lo, uint32_t hi, uint32_t s, uint64_t acc)
{
    return __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)hi << 32) | lo, s, acc);
}

template <int CH, int NG>
__device__ __forceinline__ void surf_item_u8(const uint32_t *lsrc, const uint32_t *lwin, int pitch, int h, uint32_t (&tot)[4 * NG])
{
    constexpr int F = 64 / CH;   // rows per 16-bit accumulation run: F * 4*CH <= 256 pixels, 256 * 255 < 65536
    for (int y0 = 0; y0 < h; y0 += F)
    {
        uint64_t a[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) a[g] = 0;
        const int ye = min(h, y0 + F);
        for (int y = y0; y < ye; ++y)
        {
            const uint32_t *wr = lwin + y * pitch;
            const uint32_t *sr = lsrc + y * CH;
            uint32_t W[CH + NG];
#pragma unroll
            for (int k = 0; k < CH + NG; ++k) W[k] = wr[k];
#pragma unroll
            for (int c = 0; c < CH; ++c)
            {
                const uint32_t s = sr[c];
#pragma unroll
                for (int g = 0; g < NG; ++g) a[g] = qsad(W[c + g], W[c + g + 1], s, a[g]);
            }
        }
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int k = 0; k < 4; ++k) tot[4 * g + k] += (uint32_t)(a[g] >> (16 * k)) & 0xffffu;
    }
}

// any width that is a multiple of 4 (not a PU width): the same, chunk loop not unrolled
template <int NG>
__device__ __forceinline__ void surf_item_u8_generic(const uint32_t *lsrc, const uint32_t *lwin, int pitch, int ch, int h, uint32_t (&tot)[4 * NG])
{
    for (int y = 0; y < h; ++y)
    {
        const uint32_t *wr = lwin + y * pitch;
        const uint32_t *sr = lsrc + y * ch;
        for (int c0 = 0; c0 < ch; c0 += 16)
        {
            uint64_t a[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) a[g] = 0;
            const int ce = min(ch, c0 + 16);
            for (int c = c0; c < ce; ++c)
            {
                const uint32_t s = sr[c];
#pragma unroll
                for (int g = 0; g < NG; ++g) a[g] = qsad(wr[c + g], wr[c + g + 1], s, a[g]);
            }
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int k = 0; k < 4; ++k) tot[4 * g + k] += (uint32_t)(a[g] >> (16 * k)) & 0xffffu;
        }
    }
}

// ---- 16-bit: ch = source dwords per row (w / 2); the item's 8 candidates span window dwords 0..3 (+1 for odd dx) ----

__device__ __forceinline__ void surf_item_u16(const uint32_t *lsrc, const uint32_t *lwin, int pitch, int ch, int h, uint32_t (&tot)[8])
{
    for (int y = 0; y < h; ++y)
    {
        const uint32_t *wr = lwin + y * pitch;
        const uint32_t *sr = lsrc + y * ch;
        uint32_t w0 = wr[0], w1 = wr[1], w2 = wr[2], w3 = wr[3];
        for (int c = 0; c < ch; ++c)
        {
            const uint32_t w4 = wr[c + 4], s = sr[c];
            tot[0] = __builtin_amdgcn_sad_u16(w0, s, tot[0]);
            tot[1] = __builtin_amdgcn_sad_u16(__builtin_amdgcn_alignbyte(w1, w0, 2), s, tot[1]);
            tot[2] = __builtin_amdgcn_sad_u16(w1, s, tot[2]);
            tot[3] = __builtin_amdgcn_sad_u16(__builtin_amdgcn_alignbyte(w2, w1, 2), s, tot[3]);
            tot[4] = __builtin_amdgcn_sad_u16(w2, s, tot[4]);
            tot[5] = __builtin_amdgcn_sad_u16(__builtin_amdgcn_alignbyte(w3, w2, 2), s, tot[5]);
            tot[6] = __builtin_amdgcn_sad_u16(w3, s, tot[6]);
            tot[7] = __builtin_amdgcn_sad_u16(__builtin_amdgcn_alignbyte(w4, w3, 2), s, tot[7]);
            w0 = w1; w1 = w2; w2 = w3; w3 = w4;
        }
    }
}

template <int NG>
__device__ __forceinline__ void surf_dispatch_u8(const uint32_t *lsrc, const uint32_t *lwin, int pitch, int w, int h, uint32_t (&tot)[4 * NG])
{
    switch (w >> 2)
    {
    case 1: surf_item_u8<1, NG>(lsrc, lwin, pitch, h, tot); break;
    case 2: surf_item_u8<2, NG>(lsrc, lwin, pitch, h, tot); break;
    case 3: surf_item_u8<3, NG>(lsrc, lwin, pitch, h, tot); break;
    case 4: surf_item_u8<4, NG>(lsrc, lwin, pitch, h, tot); break;
    case 6: surf_item_u8<6, NG>(lsrc, lwin, pitch, h, tot); break;
    case 8: surf_item_u8<8, NG>(lsrc, lwin, pitch, h, tot); break;
    case 12: surf_item_u8<12, NG>(lsrc, lwin, pitch, h, tot); break;
    case 16: surf_item_u8<16, NG>(lsrc, lwin, pitch, h, tot); break;
    default: surf_item_u8_generic<NG>(lsrc, lwin, pitch, w >> 2, h, tot); break;
    }
}

// NG: quad groups per item for 8-bit samples (an item = 4*NG consecutive dx); 16-bit items are always 8 dx (NG = 2)
template <int S, int NG>
__global__ __launch_bounds__(kSurfThreads) void k_sad_surface(const char *__restrict__ src, long stride_src, const char *__restrict__ ref,
                                                              long stride_ref, SurfGeom g, const int32_t *__restrict__ jobs, int njobs,
                                                              int32_t *__restrict__ out)
{
    constexpr int DXI = 4 * NG;                      // candidates per item
    extern __shared__ uint32_t lds[];
    const int tid = threadIdx.x;
    const int dyb = blockIdx.y * g.bd;               // first dy index (0 = -R) of this band
    const int nd = min(g.bd, g.side - dyb);          // candidate rows of this band
    const int job0 = xcd_block(blockIdx.x, gridDim.x) * g.jpw;
    const int nj = min(g.jpw, njobs - job0);
    const long ssb = stride_src * S, rsb = stride_ref * S;

    // ---- stage source blocks and reference windows (coalesced, unaligned dword loads) ----
    for (int q = 0; q < nj; ++q)
    {
        const int32_t *j = jobs + (long)(job0 + q) * 8;   // havoc_mi355x_surface_job
        const int so = j[0], ro = j[1], w = j[2], h = j[3];
        uint32_t *lsrc = lds + q * (g.src_dw + g.win_dw), *lwin = lsrc + g.src_dw;
        const int ch = w * S / 4;
        const FastDiv fs(ch);
        for (int i = tid; i < ch * h; i += kSurfThreads)
        {
            const int y = fs.div(i), c = i - y * ch;
            lsrc[i] = ld4(src + (long)so * S + y * ssb + 4 * c);
        }
        const int wdw = ((2 * g.R + w) * S + 3) / 4;      // dwords of a window row that hold candidate samples
        const int rows = nd + h - 1;
        const char *r0 = ref + ((long)ro - g.R) * S + (long)(dyb - g.R) * rsb;
        const FastDiv fw(wdw);
        for (int i = tid; i < wdw * rows; i += kSurfThreads)
        {
            int y, c;
            if (wdw * rows < 5000) { y = fw.div(i); c = i - y * wdw; }   // FastDiv is exact below 2^12.3
            else { y = i / wdw; c = i - y * wdw; }
            lwin[y * g.pitch + c] = ld4(r0 + y * rsb + 4 * c);
        }
    }
    __syncthreads();

    // ---- one item per thread: (job q, candidate row dyi, group gp of DXI dx) ----
    const int per_job = nd * g.ngp;
    const int q = tid / per_job;
    if (q >= nj) return;
    const int it = tid - q * per_job;
    const int dyi = it / g.ngp, gp = it - dyi * g.ngp;
    const int32_t *j = jobs + (long)(job0 + q) * 8;
    const int w = j[2], h = j[3], oo = j[4];
    const uint32_t *lsrc = lds + q * (g.src_dw + g.win_dw);
    const uint32_t *lwin = lsrc + g.src_dw + dyi * g.pitch + gp * (DXI * S / 4);
    uint32_t tot[DXI];
#pragma unroll
    for (int k = 0; k < DXI; ++k) tot[k] = 0;
    if constexpr (S == 1)
        surf_dispatch_u8<NG>(lsrc, lwin, g.pitch, w, h, tot);
    else
    {
        surf_item_u16(lsrc, lwin, g.pitch, w >> 1, h, tot);
#pragma unroll
        for (int k = 0; k < DXI; ++k) tot[k] >>= 2;   // the reference's 16-bit SAD (havoc/sad.cpp:447)
    }
    int32_t *o = out + oo + (long)(dyb + dyi) * g.side + gp * DXI;
#pragma unroll
    for (int k = 0; k < DXI; ++k)
        if (gp * DXI + k < g.side) o[k] = (int32_t)tot[k];
}

template <int S, int NG>
static hipError_t surface_launch(hipStream_t st, int range, int max_w, int max_h, const void *src, long ss, const void *ref, long sr,
                                 const void *jobs, int n, int32_t *out)
{
    SurfGeom g;
    g.R = range;
    g.side = 2 * range + 1;
    g.ngp = (g.side + 4 * NG - 1) / (4 * NG);
    g.max_w = max_w;
    g.max_h = max_h;
    g.bd = min(g.side, kSurfThreads / g.ngp);
    const int chmax = max_w * S / 4;
    g.src_dw = max(1, chmax * max_h);
    g.pitch = (g.ngp * NG * S + chmax + 2) | 1;
    g.win_dw = g.pitch * (g.bd + max_h - 1) + 8;
    g.jpw = 1;
    if (g.bd == g.side)
    {   // the whole surface is one band: several jobs per workgroup while threads and 48 KB of LDS last
        const int by_threads = kSurfThreads / (g.side * g.ngp);
        const int by_lds = (48 * 1024 / 4) / (g.src_dw + g.win_dw);
        g.jpw = max(1, min(by_threads, by_lds));
    }
    const int bands = (g.side + g.bd - 1) / g.bd;
    const dim3 grid((n + g.jpw - 1) / g.jpw, bands);
    const size_t lds = (size_t)g.jpw * (g.src_dw + g.win_dw) * 4;
    hipLaunchKernelGGL((k_sad_surface<S, NG>), grid, dim3(kSurfThreads), lds, st, (const char *)src, ss, (const char *)ref, sr, g,
                       (const int32_t *)jobs, n, out);
    return hipGetLastError();
}

hipError_t launch_sad_surface(hipStream_t st, int S, int range, int max_w, int max_h, const void *src, long ss, const void *ref, long sr,
                              const void *jobs, int n, int32_t *out)
{
    if (n == 0) return hipSuccess;
    if (S == 2) return surface_launch<2, 2>(st, range, max_w, max_h, src, ss, ref, sr, jobs, n, out);
    // 8 candidates per item (two quad-SADs per source dword); measured on MI355X: 16 per item is ~15 % slower (the kernel is
    // bound by the quad-SAD issue rate, not by LDS reads, and wider items compute more candidates beyond 2R+1)
    if (range >= 2) return surface_launch<1, 2>(st, range, max_w, max_h, src, ss, ref, sr, jobs, n, out);
    return surface_launch<1, 1>(st, range, max_w, max_h, src, ss, ref, sr, jobs, n, out);
}

} // namespace havoc_gpu
