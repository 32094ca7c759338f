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
//   1. horizontal pass of the 23 needed rows straight from HBM/L2 (hfilter4: dot4 / dot2), column-major into LDS;
//   2. vertical pass, every vertical phase from the same intermediates (below).
// All phases use the two-pass formula with the {..,64,..} filter for a zero phase, which is bit-identical to the
// reference's one-pass H-only / V-only forms for bit depths 8..10 (havoc/pred_inter.cpp:930-937 does the same).
#include "common.h"
#include "interp.h"

namespace havoc_gpu {

constexpr int kPlaneTileW = 64;   // output tile width in samples; tiles sit on a 64-sample grid of the plane (64 / 128-byte aligned rows)

// ---- the vertical pass: lane = 4 adjacent columns x 4 rows of the tile --------------------------------------------------------
// (Round 1 gave a lane one COLUMN of 16 results: a row of the tile was then spread over 64 lanes and had to be transposed -- DPP +
// v_perm, then LDS -- before it could leave in stores wider than a sample: ~11 VALU instructions per output sample, 4 of them the
// filter, 24 us per 1080p reference against 15 now; profiles/r02_experiments.md.)  A lane owns columns 4q .. 4q+3 (q = lane & 15)
// of rows 4g .. 4g+3 (g = lane >> 4): its four results of a row ARE one dword (8-bit) / one 8-byte piece (16-bit) of that row, 16 lanes write 64 contiguous samples, and nothing is
// transposed.  The rest of the per-sample work is folded away:
//   * rounding: every phase's taps sum to 64, so adding rnd / 64 = 1 << (shift - 7) to each horizontal intermediate adds
//     exactly rnd to the vertical sum (gfx950 only has the accumulating v_dot2c form, so each output still pays one v_mov 0);
//   * shift + clip + pack: the vertical taps are scaled by 1 << (16 - shift) (exact: |sum| * scale < 2^30), so the result is the
//     HIGH HALF of the accumulator; one v_perm_b32 gathers two high halves, v_sat_pk_u8_i16 clips both to 0..255 (16-bit
//     samples: v_pk_max_i16 / v_pk_min_i16), one more v_perm_b32 makes the dword;
//   * odd rows: instead of re-aligning the column (v_alignbit per pair), row j odd uses the even-aligned pairs with the taps
//     shifted by one -- (0,c0)(c1,c2)(c3,c4)(c5,c6)(c7,0) -- and pairs whose two taps are zero are dropped at compile time
//     (the loop over the four vertical phases is unrolled; only the half-sample phase needs the fifth pair).
// Intermediates sit column-major in LDS with a pitch of 26 int16 (52 bytes): the 16 q of a half-wavefront then fall in 16
// different groups of 4 banks, for the b16 writes of the horizontal pass and the dword reads of the vertical one alike.
constexpr int luma_tap(int f, int k)
{
    constexpr int t[4][8] = {{0, 0, 0, 64, 0, 0, 0, 0}, {-1, 4, -10, 58, 17, -5, 1, 0}, {-1, 4, -11, 40, 40, -11, 4, -1}, {0, 1, -5, 17, 58, -10, 4, -1}};
    return k < 0 || k > 7 ? 0 : t[f][k];
}

typedef uint32_t __attribute__((ext_vector_type(2), aligned(4))) u32x2u;

__device__ __forceinline__ uint32_t sat_pk_u8_i16(uint32_t x)
{
    uint32_t r;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(x));
    return r;   // only bits 15:0 are used below
}

template <int S, int YF>
__device__ __forceinline__ void planes_vphase(const uint32_t (&e)[4][6], int sl, int maxv, char *rowp, long rsb, int rowsLeft, int xq, int x0, int x1)
{
    typedef typename Sample<S>::T T;
    uint32_t cf[2][5];   // cf[j & 1][m]: taps of pair m for even / odd rows
#pragma unroll
    for (int odd = 0; odd < 2; ++odd)
#pragma unroll
        for (int m = 0; m < 5; ++m)
        {
            const int lo = luma_tap(YF, 2 * m - odd), hi = luma_tap(YF, 2 * m - odd + 1);
            cf[odd][m] = pack_i16(lo << sl, hi << sl);
        }
    uint32_t d[4][S];   // row j of the lane's 4 x 4 block, clipped and packed
#pragma unroll
    for (int j = 0; j < 4; ++j)
    {
        int a[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
        {
            int acc = 0;
#pragma unroll
            for (int m = 0; m < 5; ++m)
            {
                // pair m of row j covers intermediate rows (j & ~1) + 2m, +1; their taps are k = 2m - (j & 1), k + 1
                if (luma_tap(YF, 2 * m - (j & 1)) == 0 && luma_tap(YF, 2 * m - (j & 1) + 1) == 0) continue;
                acc = sdot2(e[c][(j >> 1) + m], cf[j & 1][m], acc);
            }
            a[c] = acc;
        }
        const uint32_t p01 = __builtin_amdgcn_perm((uint32_t)a[1], (uint32_t)a[0], 0x07060302u);   // high halves of a[0], a[1]
        const uint32_t p23 = __builtin_amdgcn_perm((uint32_t)a[3], (uint32_t)a[2], 0x07060302u);
        if (S == 1) d[j][0] = __builtin_amdgcn_perm(sat_pk_u8_i16(p23), sat_pk_u8_i16(p01), 0x05040100u);
        else
        {
            const s16x2 zero = {0, 0}, top = {(short)maxv, (short)maxv};
            d[j][0] = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_elementwise_max(__builtin_bit_cast(s16x2, p01), zero), top));
            d[j][S - 1] = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_elementwise_max(__builtin_bit_cast(s16x2, p23), zero), top));
        }
    }
    if (xq >= x0 && xq + 4 <= x1)
    {
#pragma unroll
        for (int j = 0; j < 4; ++j)
        {
            if (j >= rowsLeft) continue;
            if (S == 1) st4(rowp + (long)j * rsb, d[j][0]);
            else *reinterpret_cast<u32x2u *>(rowp + (long)j * rsb) = u32x2u{d[j][0], d[j][S - 1]};
        }
        return;
    }
    for (int j = 0; j < min(4, rowsLeft); ++j)   // a block the rectangle's left or right edge cuts: sample by sample
    {
        T *o = reinterpret_cast<T *>(rowp + (long)j * rsb);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (xq + k >= x0 && xq + k < x1) o[k] = (T)(d[j][(k * S) >> 2] >> (8 * ((k * S) & 3)));
    }
}

template <int S>
__global__ __launch_bounds__(256) void k_interp_planes_q(char *__restrict__ planes, long plane_elems, const char *__restrict__ ref, long stride,
                                                         int x0, int y0, int x1, int y1, int bitDepth)
{
    constexpr int TW = kPlaneTileW, THT = 16, COL = 26, NR = THT + 7;
    __shared__ __attribute__((aligned(16))) int16_t s_t[4][TW * COL];

    const int xf = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int q = lane & 15, g = lane >> 4;
    const int tx = (x0 & ~(TW - 1)) + blockIdx.x * TW, ty = y0 + blockIdx.y * THT;
    const long rsb = stride * S;
    const int shift1 = S == 1 ? 0 : min(4, bitDepth - 8);
    const int shift = S == 1 ? 12 : 6 + max(2, 14 - bitDepth);
    const int xq = tx + 4 * q;

    {
        int cx[8];
        taps_of<8>(xf, cx);
        const int bias = (1 << (shift - 7)) << shift1;   // (a + bias) >> shift1 == (a >> shift1) + rnd / 64
        const bool wanted = xq < x1 && xq + 3 >= x0;
        int16_t *t = &s_t[xf][4 * q * COL + g];
        if (wanted)
        {
            // all six row loads in flight before the first is used (rows past the last one any output needs are clamped onto
            // it: their intermediates are written but never read by a stored sample)
            const int last = y1 + 3 - (ty - 3);
            HRaw<S, 8> raw[(NR + 3) / 4];
#pragma unroll
            for (int it = 0; it < (NR + 3) / 4; ++it)
                hfilter4_load<S, 8>(ref + (long)(ty - 3 + min(g + 4 * it, last)) * rsb + (long)(xq - 3) * S, raw[it]);
#pragma unroll
            for (int it = 0; it < (NR + 3) / 4; ++it)
            {
                int a[4];
                hfilter4_eval<S, 8>(raw[it], cx, a, bias);
                // row 23 (it = 5, g = 3) is no row of the tile: it lands in the column's padding (COL = 26 > 24)
#pragma unroll
                for (int o = 0; o < 4; ++o) t[o * COL + 4 * it] = (int16_t)(a[o] >> shift1);
            }
        }
    }
    __syncthreads();

    if (xq >= x1 || xq + 3 < x0 || ty + 4 * g >= y1) return;
    uint32_t e[4][6];   // e[c][k] = intermediates (4g + 2k, 4g + 2k + 1) of column 4q + c
#pragma unroll
    for (int c = 0; c < 4; ++c)
    {
        const uint32_t *col = reinterpret_cast<const uint32_t *>(&s_t[xf][(4 * q + c) * COL + 4 * g]);
#pragma unroll
        for (int k = 0; k < 6; ++k) e[c][k] = col[k];
    }
    const int maxv = (1 << bitDepth) - 1, sl = 16 - shift, rowsLeft = y1 - (ty + 4 * g);
    char *rowp = planes + ((long)xf * plane_elems + (long)(ty + 4 * g) * stride + xq) * S;
    const long pb = 4 * plane_elems * S;   // plane 4 * yf + xf
    if (xf != 0) planes_vphase<S, 0>(e, sl, maxv, rowp, rsb, rowsLeft, xq, x0, x1);   // plane 0 is the reference picture itself
    planes_vphase<S, 1>(e, sl, maxv, rowp + pb, rsb, rowsLeft, xq, x0, x1);
    planes_vphase<S, 2>(e, sl, maxv, rowp + 2 * pb, rsb, rowsLeft, xq, x0, x1);
    planes_vphase<S, 3>(e, sl, maxv, rowp + 3 * pb, rsb, rowsLeft, xq, x0, x1);
}

hipError_t launch_interp_planes(hipStream_t st, int S, int bitDepth, void *planes, long plane_elems, const void *ref, long stride, int x0, int y0,
                                int width, int height)
{
    if (width <= 0 || height <= 0) return hipSuccess;
    const int xa = x0 & ~(kPlaneTileW - 1);
    const dim3 g((x0 + width - xa + kPlaneTileW - 1) / kPlaneTileW, (height + 15) / 16), b(256);
    if (S == 1) hipLaunchKernelGGL((k_interp_planes_q<1>), g, b, 0, st, (char *)planes, plane_elems, (const char *)ref, stride, x0, y0, x0 + width, y0 + height, bitDepth);
    else hipLaunchKernelGGL((k_interp_planes_q<2>), g, b, 0, st, (char *)planes, plane_elems, (const char *)ref, stride, x0, y0, x0 + width, y0 + height, bitDepth);
    return hipGetLastError();
}

} // namespace havoc_gpu
