// HEVC core transform basis tables and the row-times-basis building block shared by kernels_tu.hip and
// kernels_tu_fused.hip.  (Each translation unit gets its own copy of the small __constant__ tables.)
#pragma once

#include "common.h"

namespace havoc_gpu {

// ---- HEVC core transform basis, generated at compile time.  kMag[j] ~ 64*sqrt(2)*cos(j*pi/64); row k of the N-point
// DCT is row k*(32/N) of the 32-point matrix (values as in havoc/transform.cpp:85-91,119-129,170-188,243-277).
constexpr int kMag[33] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                          61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9,  4,  0};
constexpr int kDst7[4][4] = {{29, 55, 74, 84}, {74, 74, 0, -74}, {84, -29, -74, 55}, {55, -84, 74, -29}};

constexpr int basis(int n, int tr, int k, int c)
{
    if (tr) return kDst7[k][c];
    const int kk = k * (32 / n);
    if (kk == 0) return 64;
    const int j = (kk * (2 * c + 1)) & 127;
    return j <= 32 ? kMag[j] : (j <= 64 ? -kMag[64 - j] : (j <= 96 ? -kMag[j - 64] : kMag[128 - j]));
}

// v[k][p] packs (M[k][2p], M[k][2p+1]) for the forward transform, (M[2p][k], M[2p+1][k]) for the inverse
template <int N> struct PackedBasis { uint32_t v[N][N / 2]; };

template <int N, int TR, bool INV>
constexpr PackedBasis<N> make_basis()
{
    PackedBasis<N> m{};
    for (int k = 0; k < N; ++k)
        for (int p = 0; p < N / 2; ++p)
        {
            const int lo = INV ? basis(N, TR, 2 * p, k) : basis(N, TR, k, 2 * p);
            const int hi = INV ? basis(N, TR, 2 * p + 1, k) : basis(N, TR, k, 2 * p + 1);
            m.v[k][p] = ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16);
        }
    return m;
}

static __constant__ PackedBasis<4> c_fwd_dst4 = make_basis<4, 1, false>();
static __constant__ PackedBasis<4> c_fwd_dct4 = make_basis<4, 0, false>();
static __constant__ PackedBasis<8> c_fwd_dct8 = make_basis<8, 0, false>();
static __constant__ PackedBasis<16> c_fwd_dct16 = make_basis<16, 0, false>();
static __constant__ PackedBasis<32> c_fwd_dct32 = make_basis<32, 0, false>();
static __constant__ PackedBasis<4> c_inv_dst4 = make_basis<4, 1, true>();
static __constant__ PackedBasis<4> c_inv_dct4 = make_basis<4, 0, true>();
static __constant__ PackedBasis<8> c_inv_dct8 = make_basis<8, 0, true>();
static __constant__ PackedBasis<16> c_inv_dct16 = make_basis<16, 0, true>();
static __constant__ PackedBasis<32> c_inv_dct32 = make_basis<32, 0, true>();

template <int N, int TR, bool INV> __device__ __forceinline__ const PackedBasis<N> &basis_table();
template <> __device__ __forceinline__ const PackedBasis<4> &basis_table<4, 1, false>() { return c_fwd_dst4; }
template <> __device__ __forceinline__ const PackedBasis<4> &basis_table<4, 0, false>() { return c_fwd_dct4; }
template <> __device__ __forceinline__ const PackedBasis<8> &basis_table<8, 0, false>() { return c_fwd_dct8; }
template <> __device__ __forceinline__ const PackedBasis<16> &basis_table<16, 0, false>() { return c_fwd_dct16; }
template <> __device__ __forceinline__ const PackedBasis<32> &basis_table<32, 0, false>() { return c_fwd_dct32; }
template <> __device__ __forceinline__ const PackedBasis<4> &basis_table<4, 1, true>() { return c_inv_dst4; }
template <> __device__ __forceinline__ const PackedBasis<4> &basis_table<4, 0, true>() { return c_inv_dct4; }
template <> __device__ __forceinline__ const PackedBasis<8> &basis_table<8, 0, true>() { return c_inv_dct8; }
template <> __device__ __forceinline__ const PackedBasis<16> &basis_table<16, 0, true>() { return c_inv_dct16; }
template <> __device__ __forceinline__ const PackedBasis<32> &basis_table<32, 0, true>() { return c_inv_dct32; }

__device__ __forceinline__ int dot2(uint32_t a, uint32_t b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b), c, false);
}

// ---- partial butterflies for the DCT sizes where they pay (N >= 16).  DCT-II rows are (anti)symmetric:
// M[k][N-1-i] = (-1)^k M[k][i] (the structure havoc/transform.cpp's partialButterfly functions use), which halves the
// multiply-accumulates.  The products summed are exactly those of the full matrix product, in 32 bits, so results are
// bit-identical.

// inverse: even / odd coefficient pairs down a column: ve[i][q] = (M[4q][i], M[4q+2][i]), vo[i][q] = (M[4q+1][i], M[4q+3][i])
template <int N> struct PackedBasisEO { uint32_t ve[N / 2][N / 4], vo[N / 2][N / 4]; };

template <int N>
constexpr PackedBasisEO<N> make_basis_eo()
{
    PackedBasisEO<N> m{};
    for (int i = 0; i < N / 2; ++i)
        for (int q = 0; q < N / 4; ++q)
        {
            m.ve[i][q] = ((uint32_t)basis(N, 0, 4 * q, i) & 0xffffu) | ((uint32_t)basis(N, 0, 4 * q + 2, i) << 16);
            m.vo[i][q] = ((uint32_t)basis(N, 0, 4 * q + 1, i) & 0xffffu) | ((uint32_t)basis(N, 0, 4 * q + 3, i) << 16);
        }
    return m;
}
static __constant__ PackedBasisEO<16> c_eo_dct16 = make_basis_eo<16>();
static __constant__ PackedBasisEO<32> c_eo_dct32 = make_basis_eo<32>();
template <int N> __device__ __forceinline__ const PackedBasisEO<N> &basis_eo();
template <> __device__ __forceinline__ const PackedBasisEO<16> &basis_eo<16>() { return c_eo_dct16; }
template <> __device__ __forceinline__ const PackedBasisEO<32> &basis_eo<32>() { return c_eo_dct32; }

// out[k] = sum_j M[k][j] * row[j] + add, for all k, M uniform across lanes (scalar operands).
// FOLD (forward only): the caller guarantees |row[j]| <= 2^14 (the residual of samples of <= 10 bits in the first
// pass), so row[j] +- row[N-1-j] can be formed in packed 16 bits.
template <int N, int TR, bool INV, bool FOLD = false>
__device__ __forceinline__ void basis_times_row(const uint32_t (&row)[N / 2], int add, int (&out)[N])
{
    if constexpr (INV && TR == 0 && N >= 16)
    {
        const PackedBasisEO<N> &m = basis_eo<N>();
        uint32_t ce[N / 4], co[N / 4];
#pragma unroll
        for (int q = 0; q < N / 4; ++q)
        {
            ce[q] = __builtin_amdgcn_perm(row[2 * q + 1], row[2 * q], 0x05040100u);   // (c[4q], c[4q+2])
            co[q] = __builtin_amdgcn_perm(row[2 * q + 1], row[2 * q], 0x07060302u);   // (c[4q+1], c[4q+3])
        }
#pragma unroll
        for (int i = 0; i < N / 2; ++i)
        {
            int e = add, o = 0;
#pragma unroll
            for (int q = 0; q < N / 4; ++q)
            {
                e = dot2(ce[q], m.ve[i][q], e);
                o = dot2(co[q], m.vo[i][q], o);
            }
            out[i] = e + o;
            out[N - 1 - i] = e - o;
        }
    }
    else if constexpr (!INV && TR == 0 && N >= 16 && FOLD)
    {
        const PackedBasis<N> &m = basis_table<N, TR, INV>();
        uint32_t ev[N / 4], od[N / 4];
#pragma unroll
        for (int p = 0; p < N / 4; ++p)
        {
            const uint32_t b = row[N / 2 - 1 - p];
            const uint32_t bs = __builtin_amdgcn_alignbit(b, b, 16);                  // (x[N-1-2p], x[N-2-2p])
            ev[p] = pk_add(row[p], bs);
            od[p] = pk_sub(row[p], bs);
        }
#pragma unroll
        for (int k = 0; k < N; ++k)
        {
            int a = add;
#pragma unroll
            for (int p = 0; p < N / 4; ++p) a = dot2((k & 1) ? od[p] : ev[p], m.v[k][p], a);
            out[k] = a;
        }
    }
    else
    {
        const PackedBasis<N> &m = basis_table<N, TR, INV>();
#pragma unroll
        for (int k = 0; k < N; ++k)
        {
            int a = add;
#pragma unroll
            for (int p = 0; p < N / 2; ++p) a = dot2(row[p], m.v[k][p], a);
            out[k] = a;
        }
    }
}

// load N contiguous int16 (N/2 dwords) from global memory with the widest loads the size allows
template <int N>
__device__ __forceinline__ void load_row16(const int16_t *p, uint32_t (&row)[N / 2])
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
            const u32x4 v = ld16(p + 8 * q);
            row[4 * q] = v.x; row[4 * q + 1] = v.y; row[4 * q + 2] = v.z; row[4 * q + 3] = v.w;
        }
    }
}

} // namespace havoc_gpu
