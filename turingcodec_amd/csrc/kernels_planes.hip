// k_interp_planes: the 15 fractional-sample luma planes of a reference picture.
//
// A sub-pel motion search evaluates the same reference picture at many (position, phase) pairs; the reference
// interpolates a fresh block per candidate (turing/Search.hpp:1965-1998 -> HavocPredUni).  Every output SAMPLE of
// HavocPredUni depends only on the reference picture, the sample position and the phase (xFrac, yFrac) -- not on the
// block it is part of -- so on the GPU the interpolation is done ONCE per reference picture as a streaming pass that
// writes plane[4*yFrac + xFrac][y][x] = HavocPredUni(ref, x, y, xFrac, yFrac) for all 15 non-zero phases, and a
// candidate's prediction is then just a block of the right plane (its SATD a plain two-operand job).  Reads one
// plane, writes fifteen: the kernel is bound by HBM write bandwidth, which is the point.
//
// 64 x 16 output tile per 256-thread workgroup; wavefront k owns horizontal phase xFrac = k:
//   1. horizontal pass of the 23 needed rows straight from HBM/L2 (hfilter4: dot4 / dot2), transposed into LDS;
//   2. lane = column: three ds_read_b128 fetch the column's 23 intermediates, then for each vertical phase the 16
//      outputs are four v_dot2_i32_i16 each, clipped and stored (a wavefront writes 64 contiguous samples per row).
// All phases use the two-pass formula with the {..,64,..} filter for a zero phase, which is bit-identical to the
// reference's one-pass H-only / V-only forms for bit depths 8..10 (havoc/pred_inter.cpp:930-937 does the same).
#include "common.h"
#include "interp.h"

namespace havoc_gpu {

constexpr int kPlaneTileW = 64;   // output tile width (64 or 128)

template <int S>
__global__ __launch_bounds__(256) void k_interp_planes(char *__restrict__ planes, long plane_elems, const char *__restrict__ ref, long stride,
                                                       int x0, int y0, int x1, int y1, int bitDepth)
{
    typedef typename Sample<S>::T T;
    constexpr int TW = kPlaneTileW, THT = 16, COL = 24;   // column of intermediates: THT + 7 = 23 -> 24 (16-byte multiple)
    constexpr int CPL = TW / 64;                          // columns per lane (measured: 1 -> 24.5 us, 2 -> 31 us per 1080p plane set)
    __shared__ __attribute__((aligned(16))) int16_t s_t[4][TW * COL];

    const int xf = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tx = x0 + blockIdx.x * TW, ty = y0 + blockIdx.y * THT;
    const long rsb = stride * S;
    const int maxv = (1 << bitDepth) - 1;
    const int shift1 = min(4, bitDepth - 8);
    const int shift = 6 + max(2, 14 - bitDepth);

    {
        int cx[8];
        taps_of<8>(xf, cx);
        for (int i = lane; i < 23 * (TW / 4); i += kWave)
        {
            const int r = i / (TW / 4), q = i % (TW / 4);
            if (tx + 4 * q >= x1 || ty - 3 + r >= y1 + 4) continue;   // nothing in the region needs this quad
            int a[4];
            hfilter4<S, 8>(ref + (long)(ty - 3 + r) * rsb + (long)(tx + 4 * q - 3) * S, cx, a);
#pragma unroll
            for (int o = 0; o < 4; ++o) s_t[xf][(4 * q + o) * COL + r] = (int16_t)(a[o] >> shift1);
        }
    }
    __syncthreads();

    const int x = tx + CPL * lane;
    if (x >= x1) return;
    uint32_t e[CPL][12], od[CPL][11];   // (t[2k], t[2k+1]) and (t[2k+1], t[2k+2]) of each of the lane's columns
#pragma unroll
    for (int c = 0; c < CPL; ++c)
    {
        const int16_t *col = &s_t[xf][(CPL * lane + c) * COL];
        const u32x4 q0 = *reinterpret_cast<const u32x4 *>(col), q1 = *reinterpret_cast<const u32x4 *>(col + 8),
                    q2 = *reinterpret_cast<const u32x4 *>(col + 16);
        e[c][0] = q0.x; e[c][1] = q0.y; e[c][2] = q0.z; e[c][3] = q0.w; e[c][4] = q1.x; e[c][5] = q1.y;
        e[c][6] = q1.z; e[c][7] = q1.w; e[c][8] = q2.x; e[c][9] = q2.y; e[c][10] = q2.z; e[c][11] = q2.w;
#pragma unroll
        for (int k = 0; k < 11; ++k) od[c][k] = __builtin_amdgcn_alignbit(e[c][k + 1], e[c][k], 16);
    }
    const bool pair = x + 1 < x1;
    const int rnd = 1 << (shift - 1);
#pragma unroll 1
    for (int yf = 0; yf < 4; ++yf)
    {
        if ((xf | yf) == 0) continue;   // plane 0 is the reference picture itself
        int cy[8];
        taps_of<8>(yf, cy);
        uint32_t cp[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) cp[k] = pack_i16(cy[2 * k], cy[2 * k + 1]);
        T *out = reinterpret_cast<T *>(planes) + (long)(4 * yf + xf) * plane_elems + (long)ty * stride + x;
#pragma unroll
        for (int j = 0; j < THT; ++j)
        {
            int v[CPL];
#pragma unroll
            for (int c = 0; c < CPL; ++c)
            {
                int a = rnd;
#pragma unroll
                for (int k = 0; k < 4; ++k) a = sdot2((j & 1) ? od[c][(j >> 1) + k] : e[c][(j >> 1) + k], cp[k], a);
                v[c] = clip3(0, maxv, a >> shift);
            }
            if (ty + j < y1)
            {
                T *o = out + (long)j * stride;
                if (CPL == 2 && pair)
                {   // one 2-sample store per lane: 64 lanes cover 128 contiguous samples of the row
                    const int v1 = v[CPL - 1];
                    if (S == 1) *reinterpret_cast<uint16_t __attribute__((aligned(1))) *>(o) = (uint16_t)(v[0] | (v1 << 8));
                    else st4(o, (uint32_t)v[0] | ((uint32_t)v1 << 16));
                }
                else
                    o[0] = (T)v[0];
            }
        }
    }
}

hipError_t launch_interp_planes(hipStream_t st, int S, int bitDepth, void *planes, long plane_elems, const void *ref, long stride, int x0, int y0,
                                int width, int height)
{
    if (width <= 0 || height <= 0) return hipSuccess;
    const dim3 g((width + kPlaneTileW - 1) / kPlaneTileW, (height + 15) / 16), b(256);
    if (S == 1)
        hipLaunchKernelGGL((k_interp_planes<1>), g, b, 0, st, (char *)planes, plane_elems, (const char *)ref, stride, x0, y0, x0 + width, y0 + height,
                           bitDepth);
    else
        hipLaunchKernelGGL((k_interp_planes<2>), g, b, 0, st, (char *)planes, plane_elems, (const char *)ref, stride, x0, y0, x0 + width, y0 + height,
                           bitDepth);
    return hipGetLastError();
}

} // namespace havoc_gpu
