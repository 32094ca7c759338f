// k_subpel_satd: one sub-pel motion candidate = fractional-sample interpolation of the PU (HavocPredUni,
// havoc/pred_inter.cpp:113-202) followed by the PU SATD against the source block (measureSatd,
// turing/Measure.h:97-135) -- costDistortionMv in the reference (turing/Search.hpp:1965-1998).  The prediction never
// leaves the CU.
//
// Two phases, one barrier:
//   1. horizontal pass straight from HBM/L2: a lane loads the 12 (22) bytes around four adjacent output samples with
//      unaligned dword loads, slides the 8-tap window with v_alignbyte_b32 and evaluates it with v_dot4_i32_i8
//      (8-bit samples, biased to signed: sum(c) = 64 so the bias is a constant) or v_dot2_i32_i16 (16-bit samples);
//      results go to LDS TRANSPOSED (tmp[x][y]) so that phase 2 reads columns as 16-byte rows.  The source block is
//      staged the same way.
//   2. one tile COLUMN per lane: two ds_read_b128 give the 15 intermediates of the column, the vertical filter is four
//      v_dot2_i32_i16 per output on (t[y], t[y+1]) pairs, the difference against the source column feeds the
//      Hadamard: in-lane transform + DPP mirror butterflies across the 8 lanes of the tile (common.h, satd_rows).
// Every fraction combination runs the two-pass route with the {..,64,..} filter for a zero phase: that is what the
// reference's C_REF table does (havoc/pred_inter.cpp:930-937) and it is bit-identical to the one-pass / copy forms
// for bit depths 8..10 (checked bit-exactly by the u8.subpel / u16.subpel groups of tests/test_gpu_parity.py).
// PU shapes whose SATD uses 4x4 or 2x2 tiles (AMP and small chroma shapes) take a slower in-kernel path.
//
// A launch is uniform in a size class (the reference's table is indexed by width class, havoc/pred_inter.h:47-50):
// G = 8 / 32 / 128 / 256 lanes per PU for classes 8x8 / 16x16 / 32x32 / 64x64, 256 / G PUs per workgroup.
#include "common.h"
#include "interp.h"

namespace havoc_gpu {

template <int S, int TAPS, int MAXW, int MAXH, int G>
__global__ __launch_bounds__(256) void k_subpel_satd(const char *__restrict__ src, long stride_src, const char *__restrict__ ref, long stride_ref,
                                                     const int32_t *__restrict__ jobs, int njobs, int bitDepth, int32_t *__restrict__ cost)
{
    constexpr int JPW = 256 / G;
    constexpr int AB = TAPS / 2 - 1;
    constexpr int TH = MAXH + 8;      // intermediate column length: h + TAPS - 1 <= MAXH + 7, rounded to 8
    // +8 elements per PU: skews the PUs of one workgroup across LDS banks (dense arrays put every PU on bank 0)
    __shared__ __attribute__((aligned(16))) int16_t s_tmp[JPW][MAXW * TH + 8];     // [x][y], horizontal-pass output
    __shared__ __attribute__((aligned(16))) uint16_t s_src[JPW][MAXW * MAXH + 8];  // [x][y], source block
    __shared__ __attribute__((aligned(16))) int16_t s_diff[JPW][MAXH * MAXW];  // [y][x], only for 4x4 / 2x2-tile shapes
    __shared__ int s_total[JPW];

    const int sub = threadIdx.x / G, l = threadIdx.x - sub * G;
    const int job = xcd_block(blockIdx.x, gridDim.x) * JPW + sub;
    const bool live = job < njobs;
    const int32_t *j = jobs + (long)(live ? job : 0) * 8;   // havoc_mi355x_pred_uni_job; dst_off = source block offset
    const int w = j[2], h = j[3], xFrac = j[4], yFrac = j[5];
    const long ssb = stride_src * S, rsb = stride_ref * S;
    const char *s = src + (long)j[0] * S;
    const char *r0 = ref + (long)j[1] * S;
    const int maxv = (1 << bitDepth) - 1;
    const int shift1 = min(4, bitDepth - 8);
    const int shift = 6 + max(2, 14 - bitDepth);
    int16_t *tmp = s_tmp[sub];
    uint16_t *sb = s_src[sub];
    if (l == 0) s_total[sub] = 0;

    // ---- phase 1: horizontal pass and source block, both transposed into LDS
    {
        int cx[TAPS];
        taps_of<TAPS>(xFrac, cx);
        const int qpr = (w + 3) >> 2, wh = h + TAPS - 1;
        const FastDiv fq(qpr);
        for (int i = l; i < wh * qpr; i += G)
        {
            const int y = fq.div(i), x0 = (i - y * qpr) * 4;
            int a[4];
            hfilter4<S, TAPS>(r0 + (y - AB) * rsb + (x0 - AB) * S, cx, a);
#pragma unroll
            for (int o = 0; o < 4; ++o)
                if (x0 + o < w) tmp[(x0 + o) * TH + y] = (int16_t)(a[o] >> shift1);
        }
        for (int i = l; i < h * qpr; i += G)
        {
            const int y = fq.div(i), x0 = (i - y * qpr) * 4;
            const char *p = s + y * ssb + x0 * S;
            uint32_t v[4];
            if (S == 1)
            {
                const uint32_t q = ld4(p);
                v[0] = q & 0xff; v[1] = (q >> 8) & 0xff; v[2] = (q >> 16) & 0xff; v[3] = q >> 24;
            }
            else
            {
                const u32x2 q = ld8(p);
                v[0] = q.x & 0xffff; v[1] = q.x >> 16; v[2] = q.y & 0xffff; v[3] = q.y >> 16;
            }
#pragma unroll
            for (int o = 0; o < 4; ++o)
                if (x0 + o < w) sb[(x0 + o) * MAXH + y] = (uint16_t)v[o];
        }
    }
    __syncthreads();

    // ---- phase 2: vertical pass + difference + Hadamard
    int cy[TAPS];
    taps_of<TAPS>(yFrac, cy);
    const int rnd = 1 << (shift - 1);
    int acc = 0;
    const bool tiles8 = ((w | h) & 7) == 0;
    if (live && tiles8)
    {
        uint32_t cp[TAPS / 2];
#pragma unroll
        for (int k = 0; k < TAPS / 2; ++k) cp[k] = pack_i16(cy[2 * k], cy[2 * k + 1]);
        const int tw = w >> 3;
        const FastDiv ft(tw);
        for (int it = l; it < tw * (h >> 3) * 8; it += G)
        {
            const int tile = it >> 3, c = it & 7;
            const int ty = ft.div(tile), tx = tile - ty * tw;
            const int x = tx * 8 + c, y0 = ty * 8;
            const u32x4 q0 = *reinterpret_cast<const u32x4 *>(&tmp[x * TH + y0]);
            const u32x4 q1 = *reinterpret_cast<const u32x4 *>(&tmp[x * TH + y0 + 8]);
            const uint32_t e[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};   // (t[2k], t[2k+1])
            uint32_t od[7];                                                            // (t[2k+1], t[2k+2])
#pragma unroll
            for (int k = 0; k < 7; ++k) od[k] = __builtin_amdgcn_alignbit(e[k + 1], e[k], 16);
            const u32x4 qs = *reinterpret_cast<const u32x4 *>(&sb[x * MAXH + y0]);
            const uint32_t sv[4] = {qs.x, qs.y, qs.z, qs.w};
            int d[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj)
            {
                int a = rnd;
#pragma unroll
                for (int k = 0; k < TAPS / 2; ++k) a = sdot2((jj & 1) ? od[(jj >> 1) + k] : e[(jj >> 1) + k], cp[k], a);
                const int v = clip3(0, maxv, a >> shift);
                const int sj = (jj & 1) ? (int)(sv[jj >> 1] >> 16) : (int)(sv[jj >> 1] & 0xffff);
                d[jj] = sj - v;
            }
            if (S == 1)
            {
                uint32_t pk[4] = {pack_i16(d[0], d[1]), pack_i16(d[2], d[3]), pack_i16(d[4], d[5]), pack_i16(d[6], d[7])};
                acc += satd_rows_pk<8>(pk, c);
            }
            else
                acc += satd_rows<S, 8>(d, c);
        }
    }
    const bool other = live && !tiles8;
    if (other)
    {
        const FastDiv fw(w);
        for (int i = l; i < w * h; i += G)
        {
            const int y = fw.div(i), x = i - y * w;
            int a = rnd;
#pragma unroll
            for (int k = 0; k < TAPS; ++k) a += cy[k] * (int)tmp[x * TH + y + k];
            s_diff[sub][y * MAXW + x] = (int16_t)((int)sb[x * MAXH + y] - clip3(0, maxv, a >> shift));
        }
    }
    __syncthreads();
    if (other)
    {
        const int16_t *df = s_diff[sub];
        if (((w | h) & 3) == 0)
        {
            const int tw = w >> 2;
            const FastDiv ft(tw);
            for (int it = l; it < tw * (h >> 2) * 4; it += G)
            {
                const int tile = it >> 2, r = it & 3;
                const int ty = ft.div(tile), tx = tile - ty * tw;
                const u32x2 q = *reinterpret_cast<const u32x2 *>(&df[(ty * 4 + r) * MAXW + tx * 4]);
                int d[4] = {(int16_t)(q.x & 0xffff), (int16_t)(q.x >> 16), (int16_t)(q.y & 0xffff), (int16_t)(q.y >> 16)};
                acc += satd_rows<S, 4>(d, r);
            }
        }
        else
        {
            const int tw = w >> 1;
            const FastDiv ft(tw);
            for (int t = l; t < tw * (h >> 1); t += G)
            {
                const int ty = ft.div(t), tx = t - ty * tw;
                const int a = df[(2 * ty) * MAXW + 2 * tx], b = df[(2 * ty) * MAXW + 2 * tx + 1];
                const int c = df[(2 * ty + 1) * MAXW + 2 * tx], e = df[(2 * ty + 1) * MAXW + 2 * tx + 1];
                int sum = abs(a + b + c + e) + abs(a - b + c - e) + abs(a + b - c - e) + abs(a - b - c + e);
                if (S == 2) sum >>= 2;
                acc += sum;
            }
        }
    }
    if (acc) atomicAdd(&s_total[sub], acc);
    __syncthreads();
    if (live && l == 0) cost[job] = s_total[sub];
}

template <int S, int TAPS>
static hipError_t launch_subpel_satd_st(hipStream_t st, int bd, int maxw, int maxh, const char *src, long ss, const char *ref, long rs,
                                        const int32_t *jobs, int n, int32_t *cost)
{
    const dim3 b(256);
    if (maxw <= 8 && maxh <= 8)
        hipLaunchKernelGGL((k_subpel_satd<S, TAPS, 8, 8, 8>), dim3((n + 31) / 32), b, 0, st, src, ss, ref, rs, jobs, n, bd, cost);
    else if (maxw <= 16 && maxh <= 16)
        hipLaunchKernelGGL((k_subpel_satd<S, TAPS, 16, 16, 32>), dim3((n + 7) / 8), b, 0, st, src, ss, ref, rs, jobs, n, bd, cost);
    else if (maxw <= 32 && maxh <= 32)
        hipLaunchKernelGGL((k_subpel_satd<S, TAPS, 32, 32, 128>), dim3((n + 1) / 2), b, 0, st, src, ss, ref, rs, jobs, n, bd, cost);
    else
        hipLaunchKernelGGL((k_subpel_satd<S, TAPS, 64, 64, 256>), dim3(n), b, 0, st, src, ss, ref, rs, jobs, n, bd, cost);
    return hipGetLastError();
}

hipError_t launch_subpel_satd(hipStream_t st, int S, int taps, int bd, int maxw, int maxh, const void *src, long ss, const void *ref, long rs,
                              const void *jobs, int n, int32_t *cost)
{
    if (n <= 0) return hipSuccess;
    const char *s = (const char *)src, *r = (const char *)ref;
    const int32_t *j = (const int32_t *)jobs;
    if (S == 1 && taps == 8) return launch_subpel_satd_st<1, 8>(st, bd, maxw, maxh, s, ss, r, rs, j, n, cost);
    if (S == 1 && taps == 4) return launch_subpel_satd_st<1, 4>(st, bd, maxw, maxh, s, ss, r, rs, j, n, cost);
    if (S == 2 && taps == 8) return launch_subpel_satd_st<2, 8>(st, bd, maxw, maxh, s, ss, r, rs, j, n, cost);
    if (S == 2 && taps == 4) return launch_subpel_satd_st<2, 4>(st, bd, maxw, maxh, s, ss, r, rs, j, n, cost);
    return hipErrorInvalidValue;
}

} // namespace havoc_gpu
