// Shared pieces of the fractional-sample interpolation kernels (kernels_subpel.hip, kernels_planes.hip): the HEVC
// 8-tap / 4-tap coefficient tables (havoc/pred_inter.cpp:39-69) and the horizontal filter evaluated straight from
// HBM/L2 with dot instructions.
#pragma once

#include "common.h"

namespace havoc_gpu {

static __constant__ int8_t c_sp_luma[4][8] = {
    {0, 0, 0, 64, 0, 0, 0, 0}, {-1, 4, -10, 58, 17, -5, 1, 0}, {-1, 4, -11, 40, 40, -11, 4, -1}, {0, 1, -5, 17, 58, -10, 4, -1}};
static __constant__ int8_t c_sp_chroma[8][4] = {{0, 64, 0, 0},   {-2, 58, 10, -2}, {-4, 54, 16, -2}, {-6, 46, 28, -4},
                                                {-4, 36, 36, -4}, {-4, 28, 46, -6}, {-2, 16, 54, -4}, {-2, 10, 58, -2}};

template <int TAPS>
__device__ __forceinline__ void taps_of(int frac, int (&c)[TAPS])
{
#pragma unroll
    for (int k = 0; k < TAPS; ++k) c[k] = TAPS == 8 ? (int)c_sp_luma[frac][k] : (int)c_sp_chroma[frac][k];
}

__device__ __forceinline__ uint32_t pack_i16(int lo, int hi) { return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16); }
__device__ __forceinline__ uint32_t pack_i8(int a, int b, int c, int d)
{
    return ((uint32_t)a & 0xffu) | (((uint32_t)b & 0xffu) << 8) | (((uint32_t)c & 0xffu) << 16) | ((uint32_t)d << 24);
}
__device__ __forceinline__ int sdot2(uint32_t a, uint32_t b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b), c, false);
}

// four adjacent horizontal-filter outputs at p = first sample of the first window (no shift applied), in two halves so
// that a caller can have the loads of all its rows in flight before it needs the first: the dwords covering samples
// 0 .. TAPS + 2 ...
template <int S, int TAPS>
struct HRaw
{
    static constexpr int ND = S == 1 ? (TAPS == 8 ? 3 : 2) : (TAPS == 8 ? 6 : 4);
    uint32_t d[ND];
};

template <int S, int TAPS>
__device__ __forceinline__ void hfilter4_load(const char *p, HRaw<S, TAPS> &r)
{
#pragma unroll
    for (int k = 0; k < HRaw<S, TAPS>::ND; ++k) r.d[k] = ld4(p + 4 * k);
}

// ... and the four sums (+ bias)
template <int S, int TAPS>
__device__ __forceinline__ void hfilter4_eval(const HRaw<S, TAPS> &r, const int (&c)[TAPS], int (&out)[4], int bias = 0)
{
    if (S == 1)
    {
        // signed-byte trick: sum c[k]*u[k] = sum c[k]*(u[k]-128) + 128*64
        const uint32_t f = 0x80808080u;
        if (TAPS == 8)
        {
            const uint32_t clo = pack_i8(c[0], c[1], c[2], c[3]), chi = pack_i8(c[4], c[5], c[6], c[7]);
            const uint32_t d0 = r.d[0] ^ f, d1 = r.d[1] ^ f, d2 = r.d[HRaw<S, TAPS>::ND - 1] ^ f;
            out[0] = __builtin_amdgcn_sdot4(d1, chi, __builtin_amdgcn_sdot4(d0, clo, 8192 + bias, false), false);
#pragma unroll
            for (int o = 1; o < 4; ++o)
            {
                const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, o), hi = __builtin_amdgcn_alignbyte(d2, d1, o);
                out[o] = __builtin_amdgcn_sdot4(hi, chi, __builtin_amdgcn_sdot4(lo, clo, 8192 + bias, false), false);
            }
        }
        else
        {
            const uint32_t cc = pack_i8(c[0], c[1], c[2], c[3]);
            const uint32_t d0 = r.d[0] ^ f, d1 = r.d[1] ^ f;
            out[0] = __builtin_amdgcn_sdot4(d0, cc, 8192 + bias, false);
#pragma unroll
            for (int o = 1; o < 4; ++o) out[o] = __builtin_amdgcn_sdot4(__builtin_amdgcn_alignbyte(d1, d0, o), cc, 8192 + bias, false);
        }
    }
    else
    {
        constexpr int ND = HRaw<S, TAPS>::ND;
        uint32_t od[ND - 1], cp[TAPS / 2];
#pragma unroll
        for (int k = 0; k < ND - 1; ++k) od[k] = __builtin_amdgcn_alignbit(r.d[k + 1], r.d[k], 16);
#pragma unroll
        for (int k = 0; k < TAPS / 2; ++k) cp[k] = pack_i16(c[2 * k], c[2 * k + 1]);
#pragma unroll
        for (int o = 0; o < 4; ++o)
        {
            int a = bias;
#pragma unroll
            for (int k = 0; k < TAPS / 2; ++k) a = sdot2((o & 1) ? od[(o >> 1) + k] : r.d[(o >> 1) + k], cp[k], a);
            out[o] = a;
        }
    }
}

template <int S, int TAPS>
__device__ __forceinline__ void hfilter4(const char *p, const int (&c)[TAPS], int (&out)[4], int bias = 0)
{
    HRaw<S, TAPS> r;
    hfilter4_load<S, TAPS>(p, r);
    hfilter4_eval<S, TAPS>(r, c, out, bias);
}

} // namespace havoc_gpu
