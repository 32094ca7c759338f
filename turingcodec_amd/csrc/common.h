// Shared device/host helpers for libhavoc_mi355x.so (gfx950 only; wave64 is hard-coded).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/havoc_mi355x.h"

namespace havoc_gpu {

constexpr int kWave = 64;

// ---- unaligned vector loads/stores: gfx950 global/flat accesses have no alignment requirement, and block
// positions inside a picture are arbitrary (candidate motion vectors), so every wide access is typed align(1).
typedef uint32_t __attribute__((aligned(1))) u32u;
typedef uint32_t __attribute__((ext_vector_type(2), aligned(1))) u32x2u;
typedef uint32_t __attribute__((ext_vector_type(4), aligned(1))) u32x4u;
typedef uint32_t __attribute__((ext_vector_type(2))) u32x2;
typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
typedef short __attribute__((ext_vector_type(2))) s16x2;
typedef unsigned short __attribute__((ext_vector_type(2))) u16x2;

__device__ __forceinline__ uint32_t ld4(const void *p) { return *reinterpret_cast<const u32u *>(p); }
__device__ __forceinline__ u32x2 ld8(const void *p)
{
    u32x2u v = *reinterpret_cast<const u32x2u *>(p);
    return u32x2{v.x, v.y};
}
__device__ __forceinline__ u32x4 ld16(const void *p)
{
    u32x4u v = *reinterpret_cast<const u32x4u *>(p);
    return u32x4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ void st4(void *p, uint32_t v) { *reinterpret_cast<u32u *>(p) = v; }
__device__ __forceinline__ void st8(void *p, u32x2 v) { *reinterpret_cast<u32x2u *>(p) = u32x2u{v.x, v.y}; }

// ---- wave64 reductions.  DPP row operations + row broadcasts: no LDS traffic, result valid in lane 63,
// then read back uniformly with readlane.
__device__ __forceinline__ int wave_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xe, true);  // row_shr:4  (bank_mask 0xe)
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xc, true);  // row_shr:8  (bank_mask 0xc)
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);  // row_bcast:15 (row_mask 0xa)
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);  // row_bcast:31 (row_mask 0xc)
    return __builtin_amdgcn_readlane(v, 63);
}

// sum over each aligned row of 16 lanes; valid in lane 15 of the row (first four steps of wave_sum)
__device__ __forceinline__ int row16_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xe, true);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xc, true);  // row_shr:8
    return v;
}

// sum within aligned groups of G consecutive lanes (G = power of two <= 64); every lane of the group gets the sum
template <int G>
__device__ __forceinline__ int group_sum(int v)
{
#pragma unroll
    for (int o = 1; o < G; o <<= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

// XCD-aware block index.  Workgroup b is observed to run on XCD b % 8 (MI355X_MICROARCH.md, "Workgroup dispatch": for
// speed only, nothing depends on it), and every XCD has its own 4 MB L2.  Job tables are in picture (CTU raster)
// order, so handing XCD x the x-th CONTIGUOUS eighth of the blocks keeps each L2 on one band of the picture instead of
// all eight L2s holding all of it.  A bijection of [0, nb) for any nb.
__device__ __forceinline__ int xcd_block(int b, int nb)
{
    const int q = nb >> 3, r = nb & 7;
    const int x = b & 7, i = b >> 3;
    return x * q + min(x, r) + i;
}

__device__ __forceinline__ int clip3(int lo, int hi, int v) { return min(max(v, lo), hi); }

// exact y = i / d for 0 <= i < 2^12.3 (i*d_err < 2^20) and 2 <= d <= 128: one multiply instead of an integer division
struct FastDiv
{
    uint32_t inv;
    int d;
    // inv = ceil(2^20 / d), 1 <= d <= 2^20, WITHOUT an integer division (25 - 30 vector instructions where this is built per call): the float quotient is within
    // one of the true one, so "floor - 1" is at most 2 below floor(2^20 / d) and the remainder says how many steps are missing
    __device__ __forceinline__ explicit FastDiv(int d_) : d(d_)
    {
        const uint32_t ud = (uint32_t)d_;
        const uint32_t i0 = (uint32_t)(1048576.0f * __builtin_amdgcn_rcpf((float)d_)) - 1u;
        const uint32_t r = (1u << 20) - i0 * ud;
        inv = i0 + (r > 0u) + (r > ud) + (r > 2u * ud);
    }
    __device__ __forceinline__ int div(int i) const { return d == 1 ? i : (int)(((uint32_t)i * inv) >> 20); }
};

// ---- Walsh-Hadamard helpers shared by the SATD kernels -----------------------------------------------------------

// in-register length-N butterfly network (unnormalised; output order is irrelevant to SATD)
template <int N>
__device__ __forceinline__ void wht_inplace(int (&v)[N])
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

template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}

constexpr int kDppXor1 = 0xB1;        // quad_perm [1,0,3,2]
constexpr int kDppXor3 = 0x1B;        // quad_perm [3,2,1,0]
constexpr int kDppHalfMirror = 0x141; // row_half_mirror: lane ^ 7 inside each aligned group of 8 lanes

// SATD of a TS x TS tile distributed over TS consecutive lanes: lane r (0..TS-1) holds one row (or one column: SATD
// is transposition invariant) of the difference tile in d[].  In-lane transform, then the cross-lane transform with
// DPP mirror butterflies (lane^7, lane^3, lane^1): a Walsh ordering of the Hadamard coefficients, so sum|coeff| is
// the reference's (havoc/hadamard.cpp:58-98).  Lane r == 0 returns the normalised tile cost, the others 0.
template <int S, int TS>
__device__ __forceinline__ int satd_rows(int (&d)[TS], int r)
{
    wht_inplace<TS>(d);
    if (TS == 8)
    {
        const int s4 = (r & 4) ? -1 : 1;
#pragma unroll
        for (int k = 0; k < TS; ++k) d[k] = dpp_mov<kDppHalfMirror>(d[k]) + s4 * d[k];
    }
    const int s2 = (r & 2) ? -1 : 1, s1 = (r & 1) ? -1 : 1;
#pragma unroll
    for (int k = 0; k < TS; ++k) d[k] = dpp_mov<kDppXor3>(d[k]) + s2 * d[k];
    int sum = 0;
#pragma unroll
    for (int k = 0; k < TS; ++k) sum += abs(dpp_mov<kDppXor1>(d[k]) + s1 * d[k]);
    sum += dpp_mov<kDppXor1>(sum);
    sum += dpp_mov<kDppXor3>(sum);
    if (TS == 8) sum += dpp_mov<kDppHalfMirror>(sum);
    sum = (sum + TS / 4) / (TS / 2);
    if (S == 2) sum >>= 2;
    return r == 0 ? sum : 0;
}

// ---- the same on packed 16-bit pairs, for 8-bit samples (|coefficient| <= 64*255 fits int16) ------------------------
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, a) + __builtin_bit_cast(s16x2, b));
}
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, a) - __builtin_bit_cast(s16x2, b));
}
__device__ __forceinline__ uint32_t pk_bfly(uint32_t v)   // (a, b) -> (a + b, a - b)
{
    const u16x2 t = __builtin_bit_cast(u16x2, __builtin_amdgcn_alignbit(v, v, 16));   // (b, a)
    const u16x2 sgn = {1, 0xffff};   // unsigned form: the low 16 bits are those of the signed product, and it selects v_pk_mad_u16
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, v) * sgn + t);
}
__device__ __forceinline__ uint32_t pk_abs_acc(uint32_t v, uint32_t acc)   // acc + |v.lo| + |v.hi|
{
    return __builtin_amdgcn_sad_u16(v ^ 0x80008000u, 0x80008000u, acc);
}

// satd_rows on packed pairs: p[] holds the lane's TS differences two per register IN ANY ORDER that is a permutation of
// the index bits (the Walsh-Hadamard coefficient set is invariant under such permutations)
template <int TS>
__device__ __forceinline__ int satd_rows_pk(uint32_t (&p)[TS / 2], int r)
{
#pragma unroll
    for (int k = 0; k < TS / 2; ++k) p[k] = pk_bfly(p[k]);
#pragma unroll
    for (int len = 1; len < TS / 2; len <<= 1)
#pragma unroll
        for (int i = 0; i < TS / 2; i += len << 1)
#pragma unroll
            for (int k = i; k < i + len; ++k)
            {
                const uint32_t a = p[k], b = p[k + len];
                p[k] = pk_add(a, b);
                p[k + len] = pk_sub(a, b);
            }
    const s16x2 one = {1, 1}, neg = {-1, -1};
    if (TS == 8)
    {
        const s16x2 s4 = (r & 4) ? neg : one;
#pragma unroll
        for (int k = 0; k < TS / 2; ++k)
            p[k] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, p[k]) * s4 + __builtin_bit_cast(s16x2, (uint32_t)dpp_mov<kDppHalfMirror>((int)p[k])));
    }
    const s16x2 s2 = (r & 2) ? neg : one, s1 = (r & 1) ? neg : one;
#pragma unroll
    for (int k = 0; k < TS / 2; ++k)
        p[k] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, p[k]) * s2 + __builtin_bit_cast(s16x2, (uint32_t)dpp_mov<kDppXor3>((int)p[k])));
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < TS / 2; ++k)
        sum = pk_abs_acc(__builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, p[k]) * s1 + __builtin_bit_cast(s16x2, (uint32_t)dpp_mov<kDppXor1>((int)p[k]))), sum);
    int t = (int)sum;
    t += dpp_mov<kDppXor1>(t);
    t += dpp_mov<kDppXor3>(t);
    if (TS == 8) t += dpp_mov<kDppHalfMirror>(t);
    t = (t + TS / 4) / (TS / 2);
    return r == 0 ? t : 0;
}

// SATD of an 8 x 8 tile of 8-bit samples held by ONE lane (round 6): d[r][k] = row r's eight differences two per register, in the pairing of the masks
// (b0, b2), (b1, b3), (b4, b6), (b5, b7) -- any pairing that permutes the index bits leaves the coefficient set alone.  The 64-point transform is six butterfly
// stages over the index bits: two between a row's registers, three between rows (plain packed adds / subtracts), the sixth inside a register (pk_bfly) right before
// the magnitudes are summed.  352 instructions per tile where the row-per-lane form (satd_rows_pk<8>: eight lanes, three DPP stages) issues 8 x 75: the step is bound
// by instruction issue.  Returns the normalised tile cost, (sum + 2) >> 2 (havoc/hadamard.cpp:58-98).
__device__ __forceinline__ int satd_tile8_pk(uint32_t (&d)[8][4])
{
#pragma unroll
    for (int r = 0; r < 8; ++r)
    {
        const uint32_t a0 = pk_add(d[r][0], d[r][1]), a1 = pk_sub(d[r][0], d[r][1]), a2 = pk_add(d[r][2], d[r][3]), a3 = pk_sub(d[r][2], d[r][3]);
        d[r][0] = pk_add(a0, a2); d[r][2] = pk_sub(a0, a2);
        d[r][1] = pk_add(a1, a3); d[r][3] = pk_sub(a1, a3);
    }
#pragma unroll
    for (int len = 1; len < 8; len <<= 1)
#pragma unroll
        for (int i = 0; i < 8; i += len << 1)
#pragma unroll
            for (int r = i; r < i + len; ++r)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                {
                    const uint32_t a = d[r][k], b = d[r + len][k];
                    d[r][k] = pk_add(a, b);
                    d[r + len][k] = pk_sub(a, b);
                }
    uint32_t sum = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) sum = pk_abs_acc(pk_bfly(d[r][k]), sum);
    return (int)((sum + 2) >> 2);
}

// The same for 9 / 10-bit samples (the contract of S = 2, include/havoc_mi355x.h): |difference| <= 1023, so the five stages between registers stay inside 16 bits
// (1023 * 32 = 32 736) and only the sixth, inside a register, would leave them.  It is not made: |lo + hi| + |lo - hi| = 2 max(|lo|, |hi|) = |lo| + |hi| + ||lo| - |hi||,
// so X = sum of (|lo| + |hi|) and Y = sum of 2 ||lo| - |hi|| (v_sad_u16 against zero / against the swapped halves) give the tile's sum of magnitudes as X + Y / 2.
// 352 instructions per tile where the row-per-lane form (32-bit butterflies, satd_rows<2, 8>) issues 8 x ~110.  Returns ((sum + 2) >> 2) >> 2 (hadamard.cpp: the
// 16-bit tables' extra >> 2).
__device__ __forceinline__ int satd_tile8_pk16(uint32_t (&d)[8][4])
{
#pragma unroll
    for (int r = 0; r < 8; ++r)
    {
        const uint32_t a0 = pk_add(d[r][0], d[r][1]), a1 = pk_sub(d[r][0], d[r][1]), a2 = pk_add(d[r][2], d[r][3]), a3 = pk_sub(d[r][2], d[r][3]);
        d[r][0] = pk_add(a0, a2); d[r][2] = pk_sub(a0, a2);
        d[r][1] = pk_add(a1, a3); d[r][3] = pk_sub(a1, a3);
    }
#pragma unroll
    for (int len = 1; len < 8; len <<= 1)
#pragma unroll
        for (int i = 0; i < 8; i += len << 1)
#pragma unroll
            for (int r = i; r < i + len; ++r)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                {
                    const uint32_t a = d[r][k], b = d[r + len][k];
                    d[r][k] = pk_add(a, b);
                    d[r + len][k] = pk_sub(a, b);
                }
    uint32_t X = 0, Y = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k)
        {
            const s16x2 v = __builtin_bit_cast(s16x2, d[r][k]);
            const uint32_t mag = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(v, (s16x2){0, 0} - v));      // (|lo|, |hi|)
            X = __builtin_amdgcn_sad_u16(mag, 0u, X);
            Y = __builtin_amdgcn_sad_u16(mag, __builtin_amdgcn_alignbit(mag, mag, 16), Y);
        }
    const uint32_t sum = X + (Y >> 1);
    return (int)(((sum + 2) >> 2) >> 2);
}

// ---- intra prediction angles ----------------------------------------------------------------------------------

// intraPredAngle (havoc/pred_intra.cpp angle table) without a memory look-up: |angle| depends on the distance from the
// pure horizontal (10) / vertical (26) mode, nine 6-bit entries packed in one constant
__host__ __device__ constexpr int angle_of(int mode)
{
    const int d = mode < 18 ? mode - 10 : mode - 26;
    const int mag = (int)((0x2069544d245080ull >> (6 * (d < 0 ? -d : d))) & 63);   // 0 2 5 9 13 17 21 26 32
    return (mode < 18) == (d < 0) ? mag : -mag;
}
__host__ __device__ constexpr int inv_angle_of(int mode)   // modes 11..25: -round(8192 / |angle|); eight 13-bit entries
{
    const int i = (mode < 18 ? mode - 10 : 26 - mode) - 1;                      // 0..7 -> 4096 1638 910 630 482 390 315 256
    const unsigned long long lo = 4096ull | (1638ull << 13) | (910ull << 26) | (630ull << 39);
    const unsigned long long hi = 482ull | (390ull << 13) | (315ull << 26) | (256ull << 39);
    return -(int)(((i < 4 ? lo : hi) >> (13 * (i & 3))) & 0x1fff);
}

// ---- packed helpers of the intra kernels ------------------------------------------------------------------------

// TS consecutive 16-bit LDS entries as TS/2 packed pairs; the address is only 2-byte aligned (gfx950 DS instructions
// take unaligned addresses: one ds_read_b128 / b64 instead of TS ds_read_u16)
typedef uint32_t __attribute__((ext_vector_type(4), aligned(2))) u32x4h;
typedef uint32_t __attribute__((ext_vector_type(2), aligned(2))) u32x2h;

template <int TS>
__device__ __forceinline__ void ld_pairs(const uint16_t *q, uint32_t (&o)[TS / 2])
{
    if constexpr (TS == 8)
    {
        const u32x4h v = *reinterpret_cast<const u32x4h *>(q);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
    else
    {
        const u32x2h v = *reinterpret_cast<const u32x2h *>(q);
        o[0] = v.x; o[1] = v.y;
    }
}

// two angular samples at once: ((32 - f) * a + f * b + 16) >> 5 per 16-bit half.  Exact in 16 bits for samples < 2^11:
// (32 - f) * a + f * b + 16 <= 32 * 2047 + 16 < 65536; f == 0 gives a (no special case).  w0 = (32-f, 32-f), w1 = (f, f)
__device__ __forceinline__ uint32_t pk_mad_u16(uint32_t a, uint32_t b, uint32_t c)   // per 16-bit half: a * b + c (mod 2^16)
{
    uint32_t d;
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));   // the compiler splits a*b+c+k into mul, mad, add
    return d;
}
__device__ __forceinline__ uint32_t pk_lerp(uint32_t a, uint32_t b, uint32_t w0, uint32_t w1)
{
    const u16x2 five = {5, 5};
    const uint32_t t = pk_mad_u16(b, w1, pk_mad_u16(a, w0, 0x00100010u));
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, t) >> five);
}

template <int S> struct Sample;
template <> struct Sample<1> { typedef uint8_t T; };
template <> struct Sample<2> { typedef uint16_t T; };

} // namespace havoc_gpu
