// Distortion metrics: SAD, 4-way SAD, SSD, Hadamard SATD, linear SSD.
//
// Work mapping: a job (one block pair, or one block against 4 / up to 16 candidates) is owned by a GROUP of lanes -- 16
// lanes (one DPP row) for SAD / SSD, 8..64 lanes (one per 8-sample tile row) for SATD -- so a 256-thread workgroup
// carries 4..32 jobs and a frame's batch is thousands of workgroups.  Each lane loads 4/8/16-byte row chunks straight
// from L2/HBM (rows of a block are contiguous, candidate positions are arbitrary, hence the unaligned vector loads),
// accumulates with the packed byte/word SAD and dot instructions or runs the packed Hadamard, and the group's partial
// sums are folded with DPP row shifts / mirrors -- integer arithmetic only, no LDS, no MFMA.  Workgroup b takes the
// b % 8-th contiguous eighth of the job table (xcd_block): one band of the picture per XCD L2.
#include "common.h"

#include <cstdlib>

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
// 16 lanes (one DPP row) per job, four jobs per wavefront: a 16x16 8-bit block is exactly one 16-byte chunk per lane,
// 8x8 uses half a row, 64x64 takes 16 steps; the per-job cost (job fetch, address set-up, reduction) is shared four ways
#ifndef SADL
#define SADL 16
#endif
constexpr int kSadLanes = SADL;
// sum over each aligned group of kSadLanes lanes, valid in the group's last lane
__device__ __forceinline__ int sad_group_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xe, true);  // row_shr:4
    if (kSadLanes == 16) v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xc, true);  // row_shr:8
    return v;
}

// x / c for 0 <= x <= 1024, 1 <= c <= 128 without the 25-instruction integer division: (x + 0.5) / c is at least 0.5 / c from an integer, far beyond what the
// reciprocal's and the product's rounding can move it
__device__ __forceinline__ int smallDiv(int x, int c) { return (int)(((float)x + 0.5f) * __builtin_amdgcn_rcpf((float)c)); }
// products of values below 2^23 on the full-rate 24-bit multiplier (v_mul_lo_u32 runs at a quarter of the rate; k_sad4w's per-job set-up has two dozen of them)
__device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }
__device__ __forceinline__ uint32_t mulu24(uint32_t a, uint32_t b) { return __umul24(a, b); }

template <int S, int WAYS, int CB, int U = 4>
__device__ __forceinline__ void sad_block(const char *src, long ssb, const char *const (&ref)[WAYS], long rsb, int rowBytes, int h,
                                          int lane, uint32_t (&acc)[WAYS])
{
    const int cpr = rowBytes / CB;      // chunks per row: 1, 2, 3, 4, 6 or 8
    const int rpi = smallDiv(kSadLanes, cpr);    // rows per iteration
    const int y0 = smallDiv(lane, cpr);
    const int xb = (lane - y0 * cpr) * CB;
    if (y0 >= rpi) return;              // lanes beyond rpi*cpr idle (cpr = 3, 6)
#pragma unroll U                         // several rows' loads in flight: the loop is latency-, not issue-bound
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

template <int S, int WAYS, int U = 4, int MINW = 1>
__global__ __launch_bounds__(256, MINW) void k_sad(const char *__restrict__ src, long stride_src, const char *__restrict__ ref, long stride_ref,
                                             const int32_t *__restrict__ jobs, int njobs, int32_t *__restrict__ out)
{
    typedef typename Sample<S>::T T;
    const int job = (xcd_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x) / kSadLanes;
    const int lane = threadIdx.x & (kSadLanes - 1);
    const bool live = job < njobs;
    int so, w, h;
    int ro[WAYS];
    if (WAYS == 1)
    {
        const int32_t *j = jobs + (long)(live ? job : 0) * 4;   // havoc_mi355x_pair_job
        so = j[0]; ro[0] = j[1]; w = j[2]; h = live ? j[3] : 0;
    }
    else
    {
        const int32_t *j = jobs + (long)(live ? job : 0) * 8;   // havoc_mi355x_sad4_job
        so = j[0];
#pragma unroll
        for (int k = 0; k < WAYS; ++k) ro[k] = j[1 + k];
        w = j[5]; h = live ? j[6] : 0;
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
    // a chunked path needs its chunks per row to fit the lane group (cpr <= kSadLanes), else rows-per-iteration is 0:
    // 16-bit widths 34, 38 .. 62 (rowBytes % 8 == 4, > 64) take the generic loop like the odd widths
    if ((rowBytes & 15) == 0 && rowBytes <= 16 * kSadLanes) sad_block<S, WAYS, 16, U>(s, ssb, r, rsb, rowBytes, h, lane, acc);
    else if ((rowBytes & 7) == 0 && rowBytes <= 8 * kSadLanes) sad_block<S, WAYS, 8, U>(s, ssb, r, rsb, rowBytes, h, lane, acc);
    else if ((rowBytes & 3) == 0 && rowBytes <= 4 * kSadLanes) sad_block<S, WAYS, 4, U>(s, ssb, r, rsb, rowBytes, h, lane, acc);
    else
    {
        // generic widths (the reference's sadGeneric entry): one sample per lane per step
        const FastDiv fd(w);
        for (int i = lane; i < w * h; i += kSadLanes)
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
        int t = sad_group_sum((int)acc[k]);   // the last lane of each group holds its job's total
        if (S == 2) t >>= 2;
        if (live && lane == kSadLanes - 1) out[job * WAYS + k] = t;
    }
}

// ---------------------------------------------------------------------------------------------------------
// 4-way SAD with the candidates' COMMON WINDOW staged in LDS (round 4).  The four reference blocks of a havoc_sad_multiref call are the
// positions of one pattern step (turing/Search.hpp:1447-1482: a diamond / star ring / raster line around one origin), so they overlap: for a
// one-sample diamond the union of four 16x16 blocks is 18x18 samples.  k_sad<S, 4> reads each block with unaligned 16-byte loads, a row per
// lane -- 4 x 2 cache-line accesses per row, and the measured call mix (1.27 M calls per 1080p picture) made it the step's longest kernel,
// bound by the vector-memory address path (SQ counters: 23 % VALU busy, 18 loads per wavefront, no LDS).  Here a job's lane group (16 lanes,
// as before) copies the bounding box of the four blocks into LDS ONCE, with 16-byte ALIGNED loads in strips of block rows, and the four
// SADs read their (unaligned) rows from LDS.  A box that does not fit (far rings of the star: positions 16+ samples apart) takes the direct path.
// MEASURED (profiles/r04/sad4_counters.txt): the direct kernel is bound by the L1's access rate -- 474 M cache accesses per launch, texture addresser
// busy 82 % of the launch; the window cuts the accesses to 112 M.  Its first form (byte-unaligned 16-byte LDS reads) was no faster, 0.572 against 0.561 ms:
// SQ_LDS_UNALIGNED_STALL took a third of the wavefronts' cycles and it issued 2.2 x the VALU instructions.  This form -- window rows at an odd dword pitch,
// dword-aligned reads + v_alignbyte_b32, no integer division, 32-bit offsets from scalar base pointers -- runs the same calls in 0.39 ms (the direct kernel:
// 0.57 ms in the same runs) and is the default; what is left is VALU issue (152 M instructions) and LDS cycles (124 M, a third of them bank conflicts
// between the two jobs that share a 32-lane access group).
// ---------------------------------------------------------------------------------------------------------
constexpr int kSadWinBytes = 1536;      // LDS per lane group and per byte of sample size: 24 KB (8-bit) / 48 KB (16-bit) per workgroup

template <int S, int CB>
__device__ __forceinline__ void sad4_window_strips(const char *src, uint32_t s0, uint32_t ssb, const char *ref, uint32_t a0, uint32_t rsb, const __attribute__((address_space(3))) uint32_t *buf_r,
                                                   __attribute__((address_space(3))) uint32_t *buf_w, int pitchD, int chunks, int lead, const int (&ox)[4], const int (&oy)[4],
                                                   int spready, int rowBytes, int h, int hs, int lane, uint32_t (&acc)[4])
{
    // LDS side in DWORDS.  A window row takes `pitchD` dwords, an ODD number: the 16 lanes of a job (a row or a quarter row each) then read 16 different
    // banks, and the next job's buffer starts 16 banks on.  Every read is a 4-byte-aligned dword; a candidate's byte shift inside its dwords (the same for
    // all rows of a lane) is undone with v_alignbyte_b32.  (Round 4's first form read byte-unaligned 16-byte vectors: a third of the wavefronts' cycles
    // went to SQ_LDS_UNALIGNED_STALL.)
    // the copy: a lane keeps its 16-byte column and walks down the rows
    const int wstep = smallDiv(kSadLanes, chunks);   // window rows per copy iteration
    const int wr0 = smallDiv(lane, chunks), wc = lane - mul24(wr0, chunks);
    const bool copies = wr0 < wstep;
    // the SADs: a lane keeps its chunk column of the block and walks down the rows; the four candidates' LDS addresses advance together
    const int cpr = rowBytes / CB;        // chunks per block row
    const int rpi = smallDiv(kSadLanes, cpr);      // block rows per iteration of the lane group
    const int y0 = smallDiv(lane, cpr);
    const int xb = (lane - mul24(y0, cpr)) * CB;
    const bool sums = y0 < rpi;
    int lo[4], sh[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
        const int bo = lead + ox[k] + xb;
        lo[k] = mul24(y0 + oy[k], pitchD) + (bo >> 2);
        sh[k] = bo & 3;
    }
    // global side in 32-bit byte offsets from the (uniform) base pointers: the loads take the base from scalar registers, no 64-bit vector arithmetic
    const int lstep = mul24(rpi, pitchD);
    const uint32_t gstep = mulu24(wstep, rsb), sstep = mulu24(rpi, ssb);
    const int wlstep = mul24(wstep, pitchD);
    for (int ys = 0; ys < h; ys += hs)
    {
        const int he = min(hs, h - ys), nrows = he + spready;
        if (copies)
        {
            uint32_t g = a0 + mulu24(ys + wr0, rsb) + wc * 16;
            int l = mul24(wr0, pitchD) + wc * 4;
            for (int r = wr0; r < nrows; r += wstep, g += gstep, l += wlstep)
            {
                const u32x4 v = ld16(ref + g);      // 16-byte aligned when the row stride is a multiple of 16 bytes (our planes: 64)
                buf_w[l] = v.x; buf_w[l + 1] = v.y; buf_w[l + 2] = v.z; buf_w[l + 3] = v.w;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (sums)
        {
            uint32_t sp = s0 + mulu24(ys + y0, ssb) + xb;
            int l = 0;
#pragma unroll 2
            for (int y = y0; y < he; y += rpi, sp += sstep, l += lstep)
            {
                if (CB == 16)
                {
                    const u32x4 a = ld16(src + sp);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                    {
                        const auto q = buf_r + lo[k] + l;
                        const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
                        acc[k] = sad_dword<S>(a.x, __builtin_amdgcn_alignbyte(d1, d0, sh[k]), acc[k]);
                        acc[k] = sad_dword<S>(a.y, __builtin_amdgcn_alignbyte(d2, d1, sh[k]), acc[k]);
                        acc[k] = sad_dword<S>(a.z, __builtin_amdgcn_alignbyte(d3, d2, sh[k]), acc[k]);
                        acc[k] = sad_dword<S>(a.w, __builtin_amdgcn_alignbyte(d4, d3, sh[k]), acc[k]);
                    }
                }
                else if (CB == 8)
                {
                    const u32x2 a = ld8(src + sp);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                    {
                        const auto q = buf_r + lo[k] + l;
                        const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
                        acc[k] = sad_dword<S>(a.x, __builtin_amdgcn_alignbyte(d1, d0, sh[k]), acc[k]);
                        acc[k] = sad_dword<S>(a.y, __builtin_amdgcn_alignbyte(d2, d1, sh[k]), acc[k]);
                    }
                }
                else
                {
                    const uint32_t a = ld4(src + sp);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                    {
                        const auto q = buf_r + lo[k] + l;
                        acc[k] = sad_dword<S>(a, __builtin_amdgcn_alignbyte(q[1], q[0], sh[k]), acc[k]);
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();      // the strip's reads are issued before the next strip's writes (one wavefront: LDS runs them in order)
    }
}

// one havoc_sad_multiref call by the 16 lanes of a group (all four groups of a wavefront go through it together: the strips' barriers are wavefront barriers);
// `gbuf` = the group's WB * S bytes (+ 16 dwords) of LDS; the job's four totals are left in lane kSadLanes - 1 of the group
template <int S, int WB>
__device__ __forceinline__ void sad4_job(const char *__restrict__ src, long stride_src, const char *__restrict__ ref, long stride_ref, float inv_stride_ref,
                                         const int32_t *__restrict__ j, bool live, uint32_t *gbuf, int lane, int (&total)[4])
{
    const int so = j[0], w = j[5], h = live ? j[6] : 0;
    int ro[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) ro[k] = j[1 + k];
    const long ssb = stride_src * S, rsb = stride_ref * S;
    const char *s = src + (long)so * S;
    uint32_t acc[4] = {0, 0, 0, 0};
    const int rowBytes = w * S;

    // the candidates as displacements (dx, dy) from the first one: |dx| < stride / 2, so the split of a linear offset is unique
    int dx[4] = {0, 0, 0, 0}, dy[4] = {0, 0, 0, 0};
    const int st = (int)stride_ref, half = st >> 1;
#pragma unroll
    for (int k = 1; k < 4; ++k)
    {
        const int delta = ro[k] - ro[0];
        int q = (int)floorf(((float)delta + (float)half) * inv_stride_ref);
        int r = delta - q * st;
        if (r < -half) { --q; r += st; }
        if (r >= st - half) { ++q; r -= st; }
        dx[k] = r; dy[k] = q;
    }
    const int mindx = min(min(dx[0], dx[1]), min(dx[2], dx[3])), maxdx = max(max(dx[0], dx[1]), max(dx[2], dx[3]));
    const int mindy = min(min(dy[0], dy[1]), min(dy[2], dy[3])), maxdy = max(max(dy[0], dy[1]), max(dy[2], dy[3]));
    const int spready = maxdy - mindy;
    // (a job whose candidates lie thousands of rows apart makes mul24 wrap: its spready is then far beyond what fits, and `window` below is false whatever minoff is)
    const long minoff = ((long)ro[0] + mul24(mindy, st) + mindx) * S;      // bytes from `ref` to the window's first sample
    const int lead = (int)(reinterpret_cast<uintptr_t>(ref + minoff) & 15);
    const int chunks = (lead + rowBytes + (maxdx - mindx) * S + 15) >> 4;      // 16-byte pieces of a window row
    const int pitchD = 4 * chunks + 1;                                          // dwords per window row in LDS: odd (see sad4_window_strips)
    const int fit = chunks > 0 && chunks <= kSadLanes ? smallDiv(WB * S / 4, pitchD) - spready : 0;      // block rows per strip
    // the chunk size the strips pick (16 / 8 / 4 bytes by the row's alignment) must cover a block row with the 16 lanes of a job -- the direct path's own guard;
    // other widths (16-bit 34, 38, ... 62: rowBytes % 8 == 4 beyond 64 bytes) take the direct / generic path
    const bool chunked = (rowBytes & 15) == 0 ? rowBytes <= 16 * kSadLanes : (rowBytes & 7) == 0 ? rowBytes <= 8 * kSadLanes : (rowBytes & 3) == 0 && rowBytes <= 4 * kSadLanes;
    // (the strips address both pictures with 32-bit byte offsets from their base pointers: a block that reaches beyond 4 GB takes the direct path)
    const bool rows24 = (unsigned)spready < 1024u && (unsigned)h <= 64u && ssb < (1 << 23);
    const bool near = rows24 && minoff + (long)mulu24(h + spready, (uint32_t)rsb) + 16 * chunks < (1ll << 32) && ((long)so * S + (long)mulu24(h, (uint32_t)ssb) + rowBytes) < (1ll << 32);
    const bool window = live && chunked && chunks <= kSadLanes && minoff >= 16 && fit >= min(h, 4) && fit >= 1 && near;
    if (window)
    {
        int ox[4], oy[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
        {
            ox[k] = (dx[k] - mindx) * S;
            oy[k] = dy[k] - mindy;
        }
        const uint32_t a0 = (uint32_t)(minoff - lead), s0 = (uint32_t)so * S;
        const auto buf_r = (const __attribute__((address_space(3))) uint32_t *)(gbuf);
        const auto buf_w = (__attribute__((address_space(3))) uint32_t *)(gbuf);
        const int hs = min(fit, h);
        if ((rowBytes & 15) == 0) sad4_window_strips<S, 16>(src, s0, (uint32_t)ssb, ref, a0, (uint32_t)rsb, buf_r, buf_w, pitchD, chunks, lead, ox, oy, spready, rowBytes, h, hs, lane, acc);
        else if ((rowBytes & 7) == 0) sad4_window_strips<S, 8>(src, s0, (uint32_t)ssb, ref, a0, (uint32_t)rsb, buf_r, buf_w, pitchD, chunks, lead, ox, oy, spready, rowBytes, h, hs, lane, acc);
        else sad4_window_strips<S, 4>(src, s0, (uint32_t)ssb, ref, a0, (uint32_t)rsb, buf_r, buf_w, pitchD, chunks, lead, ox, oy, spready, rowBytes, h, hs, lane, acc);
    }
    else
    {   // the direct path of k_sad<S, 4>: far-apart candidates, generic widths, a window at the very start of the buffer
        typedef typename Sample<S>::T T;
        const char *r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = ref + (long)ro[k] * S;
        if ((rowBytes & 15) == 0 && rowBytes <= 16 * kSadLanes) sad_block<S, 4, 16, 1>(s, ssb, r, rsb, rowBytes, h, lane, acc);
        else if ((rowBytes & 7) == 0 && rowBytes <= 8 * kSadLanes) sad_block<S, 4, 8, 1>(s, ssb, r, rsb, rowBytes, h, lane, acc);
        else if ((rowBytes & 3) == 0 && rowBytes <= 4 * kSadLanes) sad_block<S, 4, 4, 1>(s, ssb, r, rsb, rowBytes, h, lane, acc);
        else
        {
            const FastDiv fd(w);
            for (int i = lane; i < w * h; i += kSadLanes)
            {
                const int y = fd.div(i), x = i - y * w;
                const int a = reinterpret_cast<const T *>(s + y * ssb)[x];
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] += abs(a - (int)reinterpret_cast<const T *>(r[k] + y * rsb)[x]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
        int t = sad_group_sum((int)acc[k]);
        if (S == 2) t >>= 2;
        total[k] = t;
    }
}

template <int S, int WB = kSadWinBytes, int MINW = 1>
__global__ __launch_bounds__(256, MINW) void k_sad4w(const char *__restrict__ src, long stride_src, const char *__restrict__ ref, long stride_ref, float inv_stride_ref,
                                               const int32_t *__restrict__ jobs, int njobs, int32_t *__restrict__ out)
{
    // a job's buffer: kSadWinBytes * S bytes + 16 dwords, so that consecutive jobs' buffers start 16 banks apart, + the dword a shifted read takes beyond the last row
    constexpr int kBufD = WB * S / 4 + 16;
    __shared__ uint32_t lds[(256 / kSadLanes) * kBufD + 4];
    const int group = threadIdx.x / kSadLanes, lane = threadIdx.x & (kSadLanes - 1);
    const int job = xcd_block(blockIdx.x, gridDim.x) * (256 / kSadLanes) + group;
    const bool live = job < njobs;
    int total[4];
    sad4_job<S, WB>(src, stride_src, ref, stride_ref, inv_stride_ref, jobs + (long)(live ? job : 0) * 8, live, &lds[group * kBufD], lane, total);
    if (live && lane == kSadLanes - 1)
#pragma unroll
        for (int k = 0; k < 4; ++k) out[job * 4 + k] = total[k];
}

// ---------------------------------------------------------------------------------------------------------
// 4-way SAD by RUNS (round 5).  The ~112 havoc_sad_multiref calls of one motion search (turing/Search.hpp:2224-2297 -> considerPattern :1447-1482) share
// their source block and move around one centre: k_sad4w staged a window PER CALL and spent 470 wavefront instructions per four calls on set-up (displacement
// split, bounding box, copy-in, reduction) against the 32 its absolute differences need for a 16x16 block (profiles/r04_sq_counters.csv: 149.5 M VALU
// instructions per 1.27 M-call launch).  Here a WORKGROUP takes a run = the consecutive calls of one search (the caller's run table:
// havoc_mi355x_sad4_run): one pass splits every candidate's offset into (dx, dy) and finds the run's bounding box, the box and the source block are staged in
// LDS ONCE (16-byte aligned loads, rows at an odd dword pitch as in k_sad4w), then the lane groups (16 lanes each) take the calls in turn: four packed
// displacements from LDS, four LDS base addresses, the rows' dword reads + v_alignbyte_b32 + v_sad_u8/u16, one DPP-row reduction per candidate.
// Any split of a candidate's offset into dy * stride + dx addresses the same sample in the staged box, so the float reciprocal only has to be near.  A run whose
// box does not fit (far raster rings), whose calls differ in source / size, or with an unchunkable width goes call by call through sad4_job above --
// same results either way; runs are an accelerator, not a contract (jobs outside every run are simply not computed).
// ---------------------------------------------------------------------------------------------------------
constexpr int kRunMax = 128;        // calls of a run whose displacements are kept in LDS (longer runs: call by call)

template <int S, int CB, int U>
__device__ __forceinline__ void sad4_run_calls(const __attribute__((address_space(3))) uint32_t *win, const __attribute__((address_space(3))) uint32_t *srcw,
                                               const __attribute__((address_space(3))) uint32_t *cand, int pitchD, int lead, int mndx, int mndy, int rowBytes, int h,
                                               int count, int group, int ngroups, int lane, int32_t *__restrict__ out)
{
    // U calls per pass of a lane group: their 4 U candidates read the same source chunk and their LDS reads are in flight together -- the kernel is bound by the
    // latency of a pass (LDS round trip -> 16 packed SADs -> a 4-step DPP reduction -> store), not by issue (profiles/r05/sad4r_counters.csv: 29 % of a wavefront's
    // cycles issue, 40 % wait)
    const int cpr = rowBytes / CB;                  // chunks per block row
    const int rpi = smallDiv(kSadLanes, cpr);       // block rows per iteration of the lane group
    const int y0 = smallDiv(lane, cpr);
    const int xb = (lane - mul24(y0, cpr)) * CB;
    const bool sums = y0 < rpi;
    const int lstep = mul24(rpi, pitchD);
    const int s0 = (mul24(y0, rowBytes) + xb) >> 2, sstep = mul24(rpi, rowBytes) >> 2;      // source block: dense rows, dword index
    const int base = lead + xb - mndx * S;
    for (int c0 = group; c0 < count; c0 += ngroups * U)
    {
        int lo[4 * U], sh[4 * U];
        uint32_t acc[4 * U];
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            const int c = min(c0 + u * ngroups, count - 1);      // (a pass beyond the run's end repeats its last call; nothing is stored for it)
            uint32_t v4[4];
            if (true)
            {
                const u32x4 cv = *reinterpret_cast<const __attribute__((address_space(3))) u32x4 *>(cand + 4 * c);
                v4[0] = cv.x; v4[1] = cv.y; v4[2] = cv.z; v4[3] = cv.w;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
            {
                const int v = (int)v4[k];
                const int dx = (int)(short)(v & 0xffff), dy = v >> 16;
                const int bo = base + dx * S;
                lo[4 * u + k] = mul24(y0 + dy - mndy, pitchD) + (bo >> 2);
                sh[4 * u + k] = bo & 3;
                acc[4 * u + k] = 0;
            }
        }
        if (sums)
        {
            int l = 0, sp = s0;
#pragma unroll 2
            for (int y = y0; y < h; y += rpi, sp += sstep, l += lstep)
            {
                if (CB == 16)
                {
                    // (one ds_read_b128: the source rows are dense and 16-byte aligned; four dword reads at a 16-byte lane pitch would be 4-way bank conflicts)
                    const u32x4 av = *reinterpret_cast<const __attribute__((address_space(3))) u32x4 *>(srcw + sp);
#pragma unroll
                    for (int k = 0; k < 4 * U; ++k)
                    {
                        const auto q = win + lo[k] + l;
                        const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
                        acc[k] = sad_dword<S>(av.x, __builtin_amdgcn_alignbyte(d1, d0, sh[k]), acc[k]);
                        acc[k] = sad_dword<S>(av.y, __builtin_amdgcn_alignbyte(d2, d1, sh[k]), acc[k]);
                        acc[k] = sad_dword<S>(av.z, __builtin_amdgcn_alignbyte(d3, d2, sh[k]), acc[k]);
                        acc[k] = sad_dword<S>(av.w, __builtin_amdgcn_alignbyte(d4, d3, sh[k]), acc[k]);
                    }
                }
                else if (CB == 8)
                {
                    const u32x2 av = *reinterpret_cast<const __attribute__((address_space(3))) u32x2 *>(srcw + sp);
#pragma unroll
                    for (int k = 0; k < 4 * U; ++k)
                    {
                        const auto q = win + lo[k] + l;
                        const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
                        acc[k] = sad_dword<S>(av.x, __builtin_amdgcn_alignbyte(d1, d0, sh[k]), acc[k]);
                        acc[k] = sad_dword<S>(av.y, __builtin_amdgcn_alignbyte(d2, d1, sh[k]), acc[k]);
                    }
                }
                else
                {
                    const uint32_t a0 = srcw[sp];
#pragma unroll
                    for (int k = 0; k < 4 * U; ++k)
                    {
                        const auto q = win + lo[k] + l;
                        acc[k] = sad_dword<S>(a0, __builtin_amdgcn_alignbyte(q[1], q[0], sh[k]), acc[k]);
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            const int c = c0 + u * ngroups;
#pragma unroll
            for (int k = 0; k < 4; ++k)
            {
                int t = sad_group_sum((int)acc[4 * u + k]);
                if (S == 2) t >>= 2;
                if (lane == kSadLanes - 1 && c < count) out[4 * c + k] = t;
            }
        }
    }
}

// The same run with a LANE per CANDIDATE (round 5, second form; U = 0 of k_sad4r).  In the form above 16 lanes share a call and every pass pays the call's set-up and
// four DPP reductions -- for a 16x16 block 32 of ~100 wavefront instructions are absolute differences, and the launch issues 77 M vector instructions against the
// ~31 M its sample pairs need (one v_alignbyte_b32 + one v_sad_u8 per four).  Here a lane owns one candidate: it walks the candidate's rows in the staged box (dword reads
// at its own displacement, the carry dword of v_alignbyte kept from one read to the next), the source rows are the same for every lane (one broadcast ds_read_b128 per 16
// bytes), there is NO reduction, and 64 results leave in one coalesced store.  Runs of few candidates (big blocks: the cutter keeps them to 16 calls) are cut into row
// slices so that all the workgroup's wavefronts have work; the slices' partial sums meet in LDS (ds_add_u32).
typedef u32x4 __attribute__((aligned(4))) u32x4_dw;      // four dwords at a dword-aligned address (what a scalar load needs)
typedef u32x2 __attribute__((aligned(4))) u32x2_dw;

// SG: the source rows come from global memory at a wave-uniform address -- scalar loads (s_load_dwordx4: the operand of v_sad_u8 is then an SGPR and the LDS carries
// only the box) -- instead of the broadcast reads of the block staged in LDS
template <int S, int ND, bool SG>
__device__ __forceinline__ uint32_t sad_lane_rows(const __attribute__((address_space(3))) uint32_t *q, const __attribute__((address_space(3))) uint32_t *s, int pitchD,
                                                  int nd, int rows, int sh, const uint32_t *__restrict__ gs, int gpitch)
{
    uint32_t acc = 0;
    if (SG)
    {
        if (ND >= 4)
        {
            // (four accumulators rather than one chain of 16 dependent v_sad_u8 per row: measured, no difference at six wavefronts per SIMD)
            uint32_t a1 = 0, a2 = 0, a3 = 0;
#pragma unroll 2
            for (int y = 0; y < rows; ++y, q += pitchD, gs += gpitch)
            {
                uint32_t prev = q[0];
#pragma unroll
                for (int i = 0; i < ND; i += 4)
                {
                    const u32x4 a = *reinterpret_cast<const u32x4_dw *>(gs + i);
                    const uint32_t d1 = q[i + 1], d2 = q[i + 2], d3 = q[i + 3], d4 = q[i + 4];
                    acc = sad_dword<S>(a.x, __builtin_amdgcn_alignbyte(d1, prev, sh), acc);
                    a1 = sad_dword<S>(a.y, __builtin_amdgcn_alignbyte(d2, d1, sh), a1);
                    a2 = sad_dword<S>(a.z, __builtin_amdgcn_alignbyte(d3, d2, sh), a2);
                    a3 = sad_dword<S>(a.w, __builtin_amdgcn_alignbyte(d4, d3, sh), a3);
                    prev = d4;
                }
            }
            acc += a1 + a2 + a3;
        }
        else if (ND == 2)
        {
#pragma unroll 4
            for (int y = 0; y < rows; ++y, q += pitchD, gs += gpitch)
            {
                const u32x2 a = *reinterpret_cast<const u32x2_dw *>(gs);
                const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
                acc = sad_dword<S>(a.x, __builtin_amdgcn_alignbyte(d1, d0, sh), acc);
                acc = sad_dword<S>(a.y, __builtin_amdgcn_alignbyte(d2, d1, sh), acc);
            }
        }
        else
        {
            for (int y = 0; y < rows; ++y, q += pitchD, gs += gpitch)
            {
                uint32_t prev = q[0];
                for (int i = 0; i < nd; ++i)
                {
                    const uint32_t d = q[i + 1];
                    acc = sad_dword<S>(gs[i], __builtin_amdgcn_alignbyte(d, prev, sh), acc);
                    prev = d;
                }
            }
        }
        return acc;
    }
    if (ND >= 4)
    {
#pragma unroll 2
        for (int y = 0; y < rows; ++y, q += pitchD, s += ND)
        {
            uint32_t prev = q[0];
#pragma unroll
            for (int i = 0; i < ND; i += 4)
            {
                const u32x4 a = *reinterpret_cast<const __attribute__((address_space(3))) u32x4 *>(s + i);
                const uint32_t d1 = q[i + 1], d2 = q[i + 2], d3 = q[i + 3], d4 = q[i + 4];
                acc = sad_dword<S>(a.x, __builtin_amdgcn_alignbyte(d1, prev, sh), acc);
                acc = sad_dword<S>(a.y, __builtin_amdgcn_alignbyte(d2, d1, sh), acc);
                acc = sad_dword<S>(a.z, __builtin_amdgcn_alignbyte(d3, d2, sh), acc);
                acc = sad_dword<S>(a.w, __builtin_amdgcn_alignbyte(d4, d3, sh), acc);
                prev = d4;
            }
        }
    }
    else if (ND == 2)
    {
#pragma unroll 4
        for (int y = 0; y < rows; ++y, q += pitchD, s += 2)
        {
            const u32x2 a = *reinterpret_cast<const __attribute__((address_space(3))) u32x2 *>(s);
            const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
            acc = sad_dword<S>(a.x, __builtin_amdgcn_alignbyte(d1, d0, sh), acc);
            acc = sad_dword<S>(a.y, __builtin_amdgcn_alignbyte(d2, d1, sh), acc);
        }
    }
    else
    {   // any number of dwords per row (widths 4, 12, 24, 48 ...; 64 samples of 16 bits): the loop's control is scalar, the row is walked a dword at a time
        for (int y = 0; y < rows; ++y, q += pitchD, s += nd)
        {
            uint32_t prev = q[0];
            for (int i = 0; i < nd; ++i)
            {
                const uint32_t d = q[i + 1];
                acc = sad_dword<S>(s[i], __builtin_amdgcn_alignbyte(d, prev, sh), acc);
                prev = d;
            }
        }
    }
    return acc;
}

// (Measured and dropped, profiles/r05/sad4_lane_forms.txt: the box rows through ds_read_b64 -- the lanes sorted by the parity of their first box dword, a wavefront of one
// parity reading aligned pairs -- took 0.49 ms against 0.14: an 8-byte read whose 64 addresses are scattered costs ~25 LDS cycles, not the 2 of a regular stride.)
// Also measured and dropped: the lanes taking the candidates in the order of their box row (counting sort in LDS): bank conflicts 30.9 M -> 21.9 M cycles, but the sort's
// three barriers cost more than that (0.152 against 0.144 ms); and a lane per block ROW for 64x64 blocks (source row and box row in registers, the box read once per distinct
// row by ds_read_b128, a wavefront reduction per candidate): LDS cycles 66.8 M -> 43.3 M, time 0.156 against 0.143 ms.  What the launch waits for is not the LDS: a
// workgroup's chain run record -> first job -> box -> barrier -> a scalar load per source row is memory latency, with six workgroups per CU to cover it.  PERSISTENT
// workgroups that fetch their next run's box and jobs into registers while they compute (the run record carrying the source offset and size): 0.20 ms -- the registers
// that costs leave four workgroups per CU, and ONE workgroup alone on a CU still takes ~15 k cycles per run (the per-row scalar loads miss the scalar cache).
// profiles/r05/sad4_lane_forms.txt has every number.
template <int S, int NW, bool SG>
__device__ __forceinline__ void sad4_run_lanes(const __attribute__((address_space(3))) uint32_t *win, const __attribute__((address_space(3))) uint32_t *srcw,
                                               const __attribute__((address_space(3))) uint32_t *cand, uint32_t *s_acc, int pitchD, int lead, int mndx,
                                               int mndy, int rowBytes, int h, int count, int tid, int32_t *__restrict__ out, const uint32_t *__restrict__ gsrc, int gpitch)
{
    const int P = 4 * count, nd = rowBytes >> 2, chunks = (P + 63) >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    // row slices only where wavefronts would idle (a run of few candidates: the big blocks), and of at least eight rows
    int ns = 1;
    while (ns < 8 && chunks * ns < NW && h >= 16 * ns) ns *= 2;
    const int rps = (h + ns - 1) / ns;
    if (ns > 1)
    {   // (the partial sums meet in s_acc: cleared here)
        for (int i = tid; i < P; i += 64 * NW) s_acc[i] = 0;
        __syncthreads();
    }
    for (int t = wave; t < chunks * ns; t += NW)
    {
        const int chunk = ns == 1 ? t : t / ns, slice = t - chunk * ns;
        const int pos = chunk * 64 + lane;
        const int p = pos;
        const int v = (int)cand[min(p, P - 1)];
        const int col = (int)(short)(v & 0xffff) - mndx, row = (v >> 16) - mndy;
        const int bo = lead + col * S, sh = bo & 3;
        const int yBeg = slice * rps, rows = min(h, yBeg + rps) - yBeg;
        const auto q = win + mul24(row + yBeg, pitchD) + (bo >> 2);
        const auto sp = srcw + yBeg * nd;
        const uint32_t *gs = gsrc + (long)yBeg * gpitch;
        uint32_t acc;
        if (S == 2 && nd == 32) acc = sad_lane_rows<S, 32, SG>(q, sp, pitchD, nd, rows, sh, gs, gpitch);      // 64 samples of 16 bits
        else if (nd == 16) acc = sad_lane_rows<S, 16, SG>(q, sp, pitchD, nd, rows, sh, gs, gpitch);
        else if (nd == 8) acc = sad_lane_rows<S, 8, SG>(q, sp, pitchD, nd, rows, sh, gs, gpitch);
        else if (nd == 4) acc = sad_lane_rows<S, 4, SG>(q, sp, pitchD, nd, rows, sh, gs, gpitch);
        else if (nd == 2) acc = sad_lane_rows<S, 2, SG>(q, sp, pitchD, nd, rows, sh, gs, gpitch);
        else acc = sad_lane_rows<S, 0, SG>(q, sp, pitchD, nd, rows, sh, gs, gpitch);
        if (p < P)
        {
            if (ns == 1) out[p] = (int32_t)(S == 2 ? acc >> 2 : acc);
            else atomicAdd(&s_acc[p], acc);
        }
    }
    if (ns > 1)
    {
        __syncthreads();
        for (int p = tid; p < P; p += 64 * NW) out[p] = (int32_t)(S == 2 ? s_acc[p] >> 2 : s_acc[p]);
    }
}

// the LDS of a run's workgroup
template <int S>
struct RunLds
{
    static constexpr int kWinD = (S == 1 ? 16 : 32) * 256;      // the window: 16 KB (8-bit) / 32 KB (16-bit), in dwords
    static constexpr int kSrcD = 64 * 64 * S / 4;                // the source block
    __attribute__((aligned(16))) uint32_t lds[kWinD + kSrcD + 8];
    __attribute__((aligned(16))) uint32_t cand[kRunMax * 4];     // the candidates' (column, row) in the box
    uint32_t acc[kRunMax * 4];                                    // lane-per-candidate form: where the row slices' partial sums meet
    int box[6];
};

// one run, start to finish (every thread of the workgroup; returns at uniform points, LDS may be in any state afterwards)
template <int S, int NW, int U, bool SRCG>
__device__ __forceinline__ void sad4_run_general(RunLds<S> &sh, const char *__restrict__ src, long stride_src, const char *__restrict__ ref, long stride_ref,
                                                 float inv_stride_ref, const int32_t *__restrict__ jobs, int njobs, const int32_t *__restrict__ runs, int run,
                                                 int32_t *__restrict__ out)
{
    constexpr int T = 64 * NW, NG = T / kSadLanes;
    constexpr int kWinD = RunLds<S>::kWinD, kSrcD = RunLds<S>::kSrcD;
    constexpr int kFallWB = 1024, kFallD = kFallWB * S / 4 + 16;      // the call-by-call path's buffer per lane group
    static_assert(NG * kFallD + 4 <= kWinD + kSrcD, "the call-by-call path's buffers must fit the run's LDS");
    uint32_t (&lds)[kWinD + kSrcD + 8] = sh.lds;
    uint32_t (&s_cand)[kRunMax * 4] = sh.cand;
    uint32_t (&s_acc)[kRunMax * 4] = sh.acc;
    int (&s_box)[6] = sh.box;
    const int tid = threadIdx.x, lane = tid & (kSadLanes - 1), group = tid / kSadLanes;
    const int32_t *rr = runs + (long)run * 8;      // havoc_mi355x_sad4_run
    const int first = rr[0], count = rr[1], boxOff = rr[2], boxW = rr[3], boxH = rr[4];
    if (first < 0 || count <= 0 || (long)first + count > njobs) return;      // (uniform: the whole workgroup leaves)
    const int32_t *j0 = jobs + (long)first * 8;
    const int so = j0[0], ro0 = j0[1], w = j0[5], h = j0[6];
    const int st = (int)stride_ref, half = st >> 1;
    const int rowBytes = w * S;
    const long ssb = stride_src * S, rsb = stride_ref * S;
    const bool chunked = (rowBytes & 15) == 0 ? rowBytes <= 16 * kSadLanes : (rowBytes & 7) == 0 ? rowBytes <= 8 * kSadLanes : (rowBytes & 3) == 0 && rowBytes <= 4 * kSadLanes;
    // the source block's rows at dword-aligned ADDRESSES (base pointer included: scalar loads ignore the low address bits -- ADVICE r5): scalar loads
    const bool srcScalar = U == 0 && SRCG && (((reinterpret_cast<uintptr_t>(src) + (long)so * S) | ssb) & 3) == 0;
    const bool given = boxW > 0 && boxH > 0 && boxW < st;      // the cutter's box (havoc_mi355x_sad4_make_runs): staged at once, every candidate then checked against it
    int mndx = 0, mndy = 0, spanx, spready, ok = count <= kRunMax;
    long minoff;
    if (given)
    {
        spanx = boxW - w;
        spready = boxH - h;
        minoff = (long)boxOff * S;
        ok &= spanx >= 0 && spready >= 0 && boxOff >= 0;
    }
    else
    {
        // ---- the box by reduction: every candidate's displacement from the run's first one (any split of an offset into dy * stride + dx names the same sample)
        if (tid < 6) s_box[tid] = tid == 4 ? 1 : 0;      // min dx, max dx, min dy, max dy (candidate 0 of call 0 is (0, 0)), ok, -
        __syncthreads();
        int mn_x = 0, mx_x = 0, mn_y = 0, mx_y = 0;
        for (int p = tid; p < 4 * min(count, kRunMax); p += T)
        {
            const int32_t *j = j0 + (p >> 2) * 8;
            const int delta = j[1 + (p & 3)] - ro0;
            int q = (int)floorf(((float)delta + (float)half) * inv_stride_ref);
            long rl = (long)delta - (long)q * st;      // (64-bit: a far candidate's quotient times the stride does not fit 32 bits; such a run goes call by call)
            if (rl < -half) { --q; rl += st; }
            if (rl >= st - half) { ++q; rl -= st; }
            ok &= (j[0] == so) & (j[5] == w) & (j[6] == h) & (q >= -32768) & (q < 32768) & (rl >= -32768) & (rl < 32768);
            const int r = (int)rl;
            s_cand[p] = (uint32_t)(r & 0xffff) | ((uint32_t)q << 16);
            mn_x = min(mn_x, r); mx_x = max(mx_x, r); mn_y = min(mn_y, q); mx_y = max(mx_y, q);
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1)
        {
            mn_x = min(mn_x, __shfl_xor(mn_x, o, 64)); mx_x = max(mx_x, __shfl_xor(mx_x, o, 64));
            mn_y = min(mn_y, __shfl_xor(mn_y, o, 64)); mx_y = max(mx_y, __shfl_xor(mx_y, o, 64));
            ok &= __shfl_xor(ok, o, 64);
        }
        if ((tid & 63) == 0)
        {
            atomicMin(&s_box[0], mn_x); atomicMax(&s_box[1], mx_x); atomicMin(&s_box[2], mn_y); atomicMax(&s_box[3], mx_y); atomicAnd(&s_box[4], ok);
        }
        __syncthreads();
        mndx = s_box[0];
        mndy = s_box[2];
        spanx = s_box[1] - mndx;
        spready = s_box[3] - mndy;
        ok = s_box[4];
        minoff = ((long)ro0 + (long)mndy * st + mndx) * S;      // bytes from `ref` to the box's first sample
    }
    const int lead = (int)(reinterpret_cast<uintptr_t>(ref + minoff) & 15);
    const int chunks = (lead + rowBytes + spanx * S + 15) >> 4;      // 16-byte pieces of a box row
    const int pitchD = 4 * chunks + 1, rows = h + spready;
    const bool near = (unsigned)spready < 1024u && (unsigned)h <= 64u && ssb < (1 << 23) && rsb < (1 << 23) && chunks <= 1024 && spanx >= 0 &&
                      minoff + (long)rows * rsb + 16l * chunks < (1ll << 32) && ((long)so * S + (long)h * ssb + rowBytes) < (1ll << 32);
    bool fits = ok && chunked && near && w <= 64 && minoff >= 16 && (long)pitchD * rows + 4 <= kWinD;
    if (fits)
    {
        const auto win_w = (__attribute__((address_space(3))) uint32_t *)(&lds[0]);
        const auto src_w = (__attribute__((address_space(3))) uint32_t *)(&lds[kWinD + 4]);
        // the box: 16-byte pieces, aligned in memory (rows of our planes: multiples of 64 bytes), a piece per thread
        {
            const FastDiv fc(chunks);
            const uint32_t a0 = (uint32_t)(minoff - lead);
            for (int i = tid; i < rows * chunks; i += T)
            {
                const int r = fc.div(i), c = i - r * chunks;
                const u32x4 v = ld16(ref + a0 + (uint32_t)r * (uint32_t)rsb + c * 16);
                const int l = r * pitchD + c * 4;
                win_w[l] = v.x; win_w[l + 1] = v.y; win_w[l + 2] = v.z; win_w[l + 3] = v.w;
            }
            // the source block: dense rows of dwords (not needed where the lanes read it through scalar loads)
            if (!srcScalar)
            {
                const int dpr = rowBytes >> 2;
                const FastDiv fdw(dpr);
                const uint32_t sb0 = (uint32_t)so * S;
                for (int i = tid; i < h * dpr; i += T)
                {
                    const int y = fdw.div(i), x = i - y * dpr;
                    src_w[i] = ld4(src + sb0 + (uint32_t)y * (uint32_t)ssb + x * 4);
                }
            }
        }
        if (given)
        {   // every candidate against the cutter's box, as (column, row) inside it; the run must be one search (same source block and size)
            int in = 1;
            for (int p = tid; p < 4 * count; p += T)
            {
                const int32_t *j = j0 + (p >> 2) * 8;
                const int delta = j[1 + (p & 3)] - boxOff;
                int q = (int)(((float)delta + 0.5f) * inv_stride_ref);
                int r = delta - q * st;      // (|delta| beyond the box makes this wrap or leave the box either way: the checks below catch both)
                if (r < 0) { --q; r += st; }
                if (r >= st) { ++q; r -= st; }
                in &= (j[0] == so) & (j[5] == w) & (j[6] == h) & (delta >= 0) & (q >= 0) & (q <= spready) & (r >= 0) & (r <= spanx);
                s_cand[p] = (uint32_t)(r & 0xffff) | ((uint32_t)q << 16);
            }
            fits = __syncthreads_and(in) != 0;      // (also the barrier between the staging above and the reads below)
        }
        else
            __syncthreads();
    }
    if (fits)
    {
        const auto win = (const __attribute__((address_space(3))) uint32_t *)(&lds[0]);
        const auto srcw = (const __attribute__((address_space(3))) uint32_t *)(&lds[kWinD + 4]);
        const auto cand = (const __attribute__((address_space(3))) uint32_t *)(&s_cand[0]);
        int32_t *o = out + (long)first * 4;
        if (U == 0)
        {
            const long sbyte = (long)so * S;
            if (srcScalar)
                sad4_run_lanes<S, NW, true>(win, srcw, cand, s_acc, pitchD, lead, mndx, mndy, rowBytes, h, count, tid, o, reinterpret_cast<const uint32_t *>(src + sbyte), (int)(ssb >> 2));
            else
                sad4_run_lanes<S, NW, false>(win, srcw, cand, s_acc, pitchD, lead, mndx, mndy, rowBytes, h, count, tid, o, nullptr, 0);
        }
        else if ((rowBytes & 15) == 0) sad4_run_calls<S, 16, (U ? U : 1)>(win, srcw, cand, pitchD, lead, mndx, mndy, rowBytes, h, count, group, NG, lane, o);
        else if ((rowBytes & 7) == 0) sad4_run_calls<S, 8, (U ? U : 1)>(win, srcw, cand, pitchD, lead, mndx, mndy, rowBytes, h, count, group, NG, lane, o);
        else sad4_run_calls<S, 4, (U ? U : 1)>(win, srcw, cand, pitchD, lead, mndx, mndy, rowBytes, h, count, group, NG, lane, o);
        return;
    }
    __syncthreads();      // (a run whose box did not hold: the staged window is dropped, its LDS becomes the call-by-call path's buffers)
    // ---- call by call
    for (int c = group; c < ((count + NG - 1) / NG) * NG; c += NG)
    {
        const bool live = c < count;
        int total[4];
        sad4_job<S, kFallWB>(src, stride_src, ref, stride_ref, inv_stride_ref, j0 + (long)(live ? c : 0) * 8, live, &lds[group * kFallD], lane, total);
        if (live && lane == kSadLanes - 1)
#pragma unroll
            for (int k = 0; k < 4; ++k) out[((long)first + c) * 4 + k] = total[k];
    }
}

template <int S, int NW, int U, bool SRCG = true>
__global__ __launch_bounds__(64 * NW) void k_sad4r(const char *__restrict__ src, long stride_src, const char *__restrict__ ref, long stride_ref, float inv_stride_ref,
                                                   const int32_t *__restrict__ jobs, int njobs, const int32_t *__restrict__ runs, int nruns, int32_t *__restrict__ out)
{
    __shared__ RunLds<S> sh;
    sad4_run_general<S, NW, U, SRCG>(sh, src, stride_src, ref, stride_ref, inv_stride_ref, jobs, njobs, runs, xcd_block(blockIdx.x, gridDim.x), out);
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

// 16 lanes (one DPP row) per job, four jobs per wavefront, CB-byte chunks per lane -- the SAD kernel's mapping: most
// SSD calls are 4x4 / 8x8 transform blocks (16 / 64 bytes), far too little for a wavefront each
template <int S, int CB>
__device__ __forceinline__ uint32_t ssd_block(const char *a, long sab, const char *b, long sbb, int rowBytes, int h, int lane)
{
    const int cpr = rowBytes / CB;      // chunks per row
    const int rpi = kSadLanes / cpr;    // rows per iteration (cpr <= 16: rowBytes <= 64 * 2 with CB = 16 -> 8)
    const int y0 = lane / cpr;
    const int xb = (lane - y0 * cpr) * CB;
    uint32_t acc = 0;
    if (y0 >= rpi) return 0;
#pragma unroll 2
    for (int y = y0; y < h; y += rpi)
    {
        const char *p = a + y * sab + xb, *q = b + y * sbb + xb;
        if (CB == 4) acc = ssd_dword<S>(ld4(p), ld4(q), acc);
        else if (CB == 8)
        {
            const u32x2 u = ld8(p), v = ld8(q);
            acc = ssd_dword<S>(u.x, v.x, acc);
            acc = ssd_dword<S>(u.y, v.y, acc);
        }
        else
        {
            const u32x4 u = ld16(p), v = ld16(q);
            acc = ssd_dword<S>(u.x, v.x, acc);
            acc = ssd_dword<S>(u.y, v.y, acc);
            acc = ssd_dword<S>(u.z, v.z, acc);
            acc = ssd_dword<S>(u.w, v.w, acc);
        }
    }
    return acc;
}

template <int S>
__global__ __launch_bounds__(256) void k_ssd(const char *__restrict__ pa, long stride_a, const char *__restrict__ pb, long stride_b,
                                             const int32_t *__restrict__ jobs, int njobs, uint32_t *__restrict__ out)
{
    typedef typename Sample<S>::T T;
    const int job = (xcd_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x) / kSadLanes;
    const int lane = threadIdx.x & (kSadLanes - 1);
    const bool live = job < njobs;
    const int32_t *j = jobs + (long)(live ? job : 0) * 4;
    const int w = j[2], h = live ? j[3] : 0;
    const long sab = stride_a * S, sbb = stride_b * S;
    const char *a = pa + (long)j[0] * S;
    const char *b = pb + (long)j[1] * S;
    uint32_t acc = 0;
    const int rowBytes = w * S;
    if ((rowBytes & 15) == 0 && rowBytes <= 16 * kSadLanes) acc = ssd_block<S, 16>(a, sab, b, sbb, rowBytes, h, lane);
    else if ((rowBytes & 7) == 0 && rowBytes <= 8 * kSadLanes) acc = ssd_block<S, 8>(a, sab, b, sbb, rowBytes, h, lane);
    else if ((rowBytes & 3) == 0 && rowBytes <= 4 * kSadLanes) acc = ssd_block<S, 4>(a, sab, b, sbb, rowBytes, h, lane);
    else
    {
        const FastDiv fd(w);
        for (int i = lane; i < w * h; i += kSadLanes)
        {
            const int y = fd.div(i), x = i - y * w;
            const int d = (int)reinterpret_cast<const T *>(a + y * sab)[x] - (int)reinterpret_cast<const T *>(b + y * sbb)[x];
            acc += (uint32_t)(d * d);
        }
    }
    uint32_t t = (uint32_t)sad_group_sum((int)acc);   // the last lane of the group holds the job's total
    if (S == 2) t >>= 4;
    if (live && lane == kSadLanes - 1) out[job] = t;
}

// ---------------------------------------------------------------------------------------------------------
// Hadamard SATD  (reference: havoc/hadamard.cpp:58-98; PU tiling: turing/Measure.h:97-135)
// One tile ROW per lane: the lane loads its 8 (4) samples of both operands with one unaligned vector load each,
// takes the differences, transforms them in registers, and the 8 (4) lanes of a tile finish the transform with DPP
// mirror butterflies (common.h, satd_rows).  G = 8 / 16 / 32 / 64 lanes work on one job (chosen from the batch's
// max block size), so a wavefront carries 8 jobs of 8x8 or one job of 64x64 with every lane busy.
// ---------------------------------------------------------------------------------------------------------

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
        else
        {
            const uint32_t va = ld4(a), vb = ld4(b);
#pragma unroll
            for (int x = 0; x < 4; ++x) d[x] = (int)((va >> (8 * x)) & 0xff) - (int)((vb >> (8 * x)) & 0xff);
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
        else
        {
            const u32x2 va = ld8(a), vb = ld8(b);
            d[0] = (int)(va.x & 0xffff) - (int)(vb.x & 0xffff);
            d[1] = (int)(va.x >> 16) - (int)(vb.x >> 16);
            d[2] = (int)(va.y & 0xffff) - (int)(vb.y & 0xffff);
            d[3] = (int)(va.y >> 16) - (int)(vb.y >> 16);
        }
    }
}

template <int S, int G>
__global__ __launch_bounds__(256) void k_satd(const char *__restrict__ pa, long stride_a, const char *__restrict__ pb, long stride_b,
                                              const int32_t *__restrict__ jobs, int njobs, int32_t *__restrict__ out)
{
    typedef typename Sample<S>::T T;
    const int job = (xcd_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x) / G;
    const int l = threadIdx.x & (G - 1);
    const bool live = job < njobs;
    const int32_t *j = jobs + (long)(live ? job : 0) * 4;
    const int w = j[2], h = live ? j[3] : 0;
    const long sab = stride_a * S, sbb = stride_b * S;
    const char *a = pa + (long)j[0] * S;
    const char *b = pb + (long)j[1] * S;
    int acc = 0;
    if (((w | h) & 7) == 0)
    {   // 8x8 tiles (turing/Measure.h:123-133)
        const int tw = w >> 3;
        const FastDiv fd(tw);
        for (int it = l; it < tw * (h >> 3) * 8; it += G)
        {
            const int tile = it >> 3, r = it & 7;
            const int ty = fd.div(tile), tx = tile - ty * tw;
            const char *pa8 = a + (long)(ty * 8 + r) * sab + tx * 8 * S, *pb8 = b + (long)(ty * 8 + r) * sbb + tx * 8 * S;
            if (S == 1)
            {   // bytes (0,2) / (1,3) of each dword become 16-bit pairs: an index-bit permutation, fine for the Hadamard
                const u32x2 va = ld8(pa8), vb = ld8(pb8);
                const uint32_t m = 0x00ff00ffu;
                uint32_t p[4] = {pk_sub(va.x & m, vb.x & m), pk_sub((va.x >> 8) & m, (vb.x >> 8) & m),
                                 pk_sub(va.y & m, vb.y & m), pk_sub((va.y >> 8) & m, (vb.y >> 8) & m)};
                acc += satd_rows_pk<8>(p, r);
            }
            else
            {
                int d[8];
                load_diff_row<S, 8>(pa8, pb8, d);
                acc += satd_rows<S, 8>(d, r);
            }
        }
    }
    else if (((w | h) & 3) == 0)
    {   // 4x4 tiles (:112-122)
        const int tw = w >> 2;
        const FastDiv fd(tw);
        for (int it = l; it < tw * (h >> 2) * 4; it += G)
        {
            const int tile = it >> 2, r = it & 3;
            const int ty = fd.div(tile), tx = tile - ty * tw;
            const char *pa4 = a + (long)(ty * 4 + r) * sab + tx * 4 * S, *pb4 = b + (long)(ty * 4 + r) * sbb + tx * 4 * S;
            if (S == 1)
            {
                const uint32_t va = ld4(pa4), vb = ld4(pb4), m = 0x00ff00ffu;
                uint32_t p[2] = {pk_sub(va & m, vb & m), pk_sub((va >> 8) & m, (vb >> 8) & m)};
                acc += satd_rows_pk<4>(p, r);
            }
            else
            {
                int d[4];
                load_diff_row<S, 4>(pa4, pb4, d);
                acc += satd_rows<S, 4>(d, r);
            }
        }
    }
    else
    {   // 2x2 tiles (:100-111): one tile per lane, no normalisation
        const int tw = w >> 1;
        const FastDiv fd(tw);
        for (int t = l; t < tw * (h >> 1); t += G)
        {
            const int ty = fd.div(t), tx = t - ty * tw;
            const T *p = reinterpret_cast<const T *>(a + (long)(2 * ty) * sab) + 2 * tx;
            const T *q = reinterpret_cast<const T *>(b + (long)(2 * ty) * sbb) + 2 * tx;
            const int d0 = (int)p[0] - (int)q[0], d1 = (int)p[1] - (int)q[1];
            const int d2 = (int)p[stride_a] - (int)q[stride_b], d3 = (int)p[stride_a + 1] - (int)q[stride_b + 1];
            int sum = abs(d0 + d1 + d2 + d3) + abs(d0 - d1 + d2 - d3) + abs(d0 + d1 - d2 - d3) + abs(d0 - d1 - d2 + d3);
            if (S == 2) sum >>= 2;
            acc += sum;
        }
    }
    const int t = G == 64 ? wave_sum(acc) : group_sum<G>(acc);
    if (live && l == 0) out[job] = t;
}

// One block `a` against up to 16 candidate blocks `b` (the sub-pel stage evaluates 8 half- and 8 quarter-sample
// candidates per PU and list, turing/Search.hpp:1965-1998): the rows of `a` are fetched and unpacked once and stay in
// registers while the candidates stream through -- half the row fetches of 16 separate jobs.
// job: havoc_mi355x_satd_multi_job { a_off, w, h, count, b_off[16] }
constexpr int kSatdMulti = 16;

// CPS = candidates per lane group: a job's 16 candidates are split over 16 / CPS lane groups (more wavefronts in
// flight; the source rows are then fetched 16 / CPS times instead of once)
// TILE (round 6; blocks of >= four 8 x 8 tiles): a LANE takes a whole 8 x 8 tile instead of one of its rows (common.h: satd_tile8_pk / satd_tile8_pk16), G = the tiles of
// the class's largest block.  No cross-lane step but the final sum, 0.59 of the instructions.
template <int S, int G, int CPS, bool TILE = false>
__global__ __launch_bounds__(256) void k_satd_multi(const char *__restrict__ pa, long stride_a, const char *__restrict__ pb, long stride_b,
                                                    const int32_t *__restrict__ jobs, int njobs, int32_t *__restrict__ out)
{
    typedef typename Sample<S>::T T;
    constexpr int SL = kSatdMulti / CPS;             // slices per job
    const int grp = (xcd_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x) / G;
    const int job = grp / SL, k0 = (grp - job * SL) * CPS;
    const int l = threadIdx.x & (G - 1);
    const bool live = job < njobs;
    const int32_t *j = jobs + (long)(live ? job : 0) * (4 + kSatdMulti);
    const int w = j[1], cnt = min(max(j[3], 0), kSatdMulti), h = (live && k0 < cnt) ? j[2] : 0;   // count outside 1..16 is clamped
    const long sab = stride_a * S, sbb = stride_b * S;
    const char *a = pa + (long)j[0] * S;
    const char *b[CPS];
#pragma unroll
    for (int k = 0; k < CPS; ++k) b[k] = pb + (long)j[4 + max(0, min(k0 + k, cnt - 1))] * S;   // slots beyond count repeat the last one
    int acc[CPS];
#pragma unroll
    for (int k = 0; k < CPS; ++k) acc[k] = 0;
    if (TILE && S == 2 && ((w | h) & 7) == 0)
    {   // 9 / 10-bit samples: a row of eight is four dwords of pairs already
        const int tw = w >> 3;
        const FastDiv fd(tw);
        for (int t = l; t < tw * (h >> 3); t += G)
        {
            const int ty = fd.div(t), tx = t - ty * tw;
            const char *pa8 = a + (long)(ty * 8) * sab + tx * 16;
            const long ob = (long)(ty * 8) * sbb + tx * 16;
            uint32_t ap[8][4];
#pragma unroll
            for (int r = 0; r < 8; ++r)
            {
                const u32x4 va = ld16(pa8 + r * sab);
                ap[r][0] = va.x; ap[r][1] = va.y; ap[r][2] = va.z; ap[r][3] = va.w;
            }
#pragma unroll
            for (int k = 0; k < CPS; ++k)
            {
                uint32_t d[8][4];
#pragma unroll
                for (int r = 0; r < 8; ++r)
                {
                    const u32x4 vb = ld16(b[k] + ob + r * sbb);
                    d[r][0] = pk_sub(ap[r][0], vb.x); d[r][1] = pk_sub(ap[r][1], vb.y); d[r][2] = pk_sub(ap[r][2], vb.z); d[r][3] = pk_sub(ap[r][3], vb.w);
                }
                acc[k] += satd_tile8_pk16(d);
            }
        }
    }
    else if (TILE && S == 1 && ((w | h) & 7) == 0)
    {
        const int tw = w >> 3;
        const FastDiv fd(tw);
        const uint32_t m = 0x00ff00ffu;
        for (int t = l; t < tw * (h >> 3); t += G)
        {
            const int ty = fd.div(t), tx = t - ty * tw;
            const char *pa8 = a + (long)(ty * 8) * sab + tx * 8;
            const long ob = (long)(ty * 8) * sbb + tx * 8;
            uint32_t ap[8][4];
#pragma unroll
            for (int r = 0; r < 8; ++r)
            {
                const u32x2 va = ld8(pa8 + r * sab);
                ap[r][0] = va.x & m; ap[r][1] = (va.x >> 8) & m; ap[r][2] = va.y & m; ap[r][3] = (va.y >> 8) & m;
            }
#pragma unroll
            for (int k = 0; k < CPS; ++k)
            {
                uint32_t d[8][4];
#pragma unroll
                for (int r = 0; r < 8; ++r)
                {
                    const u32x2 vb = ld8(b[k] + ob + r * sbb);
                    d[r][0] = pk_sub(ap[r][0], vb.x & m); d[r][1] = pk_sub(ap[r][1], (vb.x >> 8) & m);
                    d[r][2] = pk_sub(ap[r][2], vb.y & m); d[r][3] = pk_sub(ap[r][3], (vb.y >> 8) & m);
                }
                acc[k] += satd_tile8_pk(d);
            }
        }
    }
    else if (((w | h) & 7) == 0)
    {
        const int tw = w >> 3;
        const FastDiv fd(tw);
        for (int it = l; it < tw * (h >> 3) * 8; it += G)
        {
            const int tile = it >> 3, r = it & 7;
            const int ty = fd.div(tile), tx = tile - ty * tw;
            const char *pa8 = a + (long)(ty * 8 + r) * sab + tx * 8 * S;
            const long ob = (long)(ty * 8 + r) * sbb + tx * 8 * S;
            if (S == 1)
            {
                const u32x2 va = ld8(pa8);
                u32x2 vb[CPS];
#pragma unroll
                for (int k = 0; k < CPS; ++k) vb[k] = ld8(b[k] + ob);
                const uint32_t m = 0x00ff00ffu;
                const uint32_t ap[4] = {va.x & m, (va.x >> 8) & m, va.y & m, (va.y >> 8) & m};
#pragma unroll
                for (int k = 0; k < CPS; ++k)
                {
                    uint32_t p[4] = {pk_sub(ap[0], vb[k].x & m), pk_sub(ap[1], (vb[k].x >> 8) & m), pk_sub(ap[2], vb[k].y & m),
                                     pk_sub(ap[3], (vb[k].y >> 8) & m)};
                    acc[k] += satd_rows_pk<8>(p, r);
                }
            }
            else
            {
                const u32x4 va = ld16(pa8);
                const uint32_t wa[4] = {va.x, va.y, va.z, va.w};
#pragma unroll
                for (int k = 0; k < CPS; ++k)
                {
                    const u32x4 vb = ld16(b[k] + ob);
                    const uint32_t wb[4] = {vb.x, vb.y, vb.z, vb.w};
                    int d[8];
#pragma unroll
                    for (int x = 0; x < 4; ++x)
                    {
                        d[2 * x] = (int)(wa[x] & 0xffff) - (int)(wb[x] & 0xffff);
                        d[2 * x + 1] = (int)(wa[x] >> 16) - (int)(wb[x] >> 16);
                    }
                    acc[k] += satd_rows<S, 8>(d, r);
                }
            }
        }
    }
    else if (((w | h) & 3) == 0)
    {
        const int tw = w >> 2;
        const FastDiv fd(tw);
        for (int it = l; it < tw * (h >> 2) * 4; it += G)
        {
            const int tile = it >> 2, r = it & 3;
            const int ty = fd.div(tile), tx = tile - ty * tw;
            const char *pa4 = a + (long)(ty * 4 + r) * sab + tx * 4 * S;
            const long ob = (long)(ty * 4 + r) * sbb + tx * 4 * S;
#pragma unroll
            for (int k = 0; k < CPS; ++k)
            {
                if (S == 1)
                {
                    const uint32_t va = ld4(pa4), vb = ld4(b[k] + ob), m = 0x00ff00ffu;
                    uint32_t p[2] = {pk_sub(va & m, vb & m), pk_sub((va >> 8) & m, (vb >> 8) & m)};
                    acc[k] += satd_rows_pk<4>(p, r);
                }
                else
                {
                    int d[4];
                    load_diff_row<S, 4>(pa4, b[k] + ob, d);
                    acc[k] += satd_rows<S, 4>(d, r);
                }
            }
        }
    }
    else
    {   // 2x2 tiles: one tile per lane, no normalisation
        const int tw = w >> 1;
        const FastDiv fd(tw);
        for (int t = l; t < tw * (h >> 1); t += G)
        {
            const int ty = fd.div(t), tx = t - ty * tw;
            const T *p = reinterpret_cast<const T *>(a + (long)(2 * ty) * sab) + 2 * tx;
#pragma unroll
            for (int k = 0; k < CPS; ++k)
            {
                const T *q = reinterpret_cast<const T *>(b[k] + (long)(2 * ty) * sbb) + 2 * tx;
                const int d0 = (int)p[0] - (int)q[0], d1 = (int)p[1] - (int)q[1];
                const int d2 = (int)p[stride_a] - (int)q[stride_b], d3 = (int)p[stride_a + 1] - (int)q[stride_b + 1];
                int sum = abs(d0 + d1 + d2 + d3) + abs(d0 - d1 + d2 - d3) + abs(d0 + d1 - d2 - d3) + abs(d0 - d1 - d2 + d3);
                if (S == 2) sum >>= 2;
                acc[k] += sum;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < CPS; ++k)
    {
        const int t = G == 64 ? wave_sum(acc[k]) : group_sum<G>(acc[k]);
        if (h != 0 && l == 0 && k0 + k < cnt) out[(long)job * kSatdMulti + k0 + k] = t;
    }
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

// which 4-way kernel (read once per process).  Default = k_sad4w, the candidates' common window through LDS: 0.39 ms for the 1.27 M calls of a 1080p picture
// (profiles/r04/sad4_counters.txt; 1 536 B of LDS per job and 6 wavefronts per SIMD -- 1 024 B / 8 wavefronts measured the same, 768 B slower).
// HAVOC_SAD4_WINDOW=0 = k_sad<S, 4> reading the four blocks directly, two rows in flight per lane (0.57 ms); HAVOC_SAD4_DIRECT=1 = round 1's form of it.
static int sad4_form()
{
    static const int v = [] {
        const char *w = getenv("HAVOC_SAD4_WINDOW"), *d = getenv("HAVOC_SAD4_DIRECT");
        return (d && *d == '1') ? 1 : ((w && *w == '0') ? 0 : 2);
    }();
    return v;
}

// workgroup size of the run kernel (read once per process): HAVOC_SAD4_RUN_WAVES = 1, 2 or 4 wavefronts share a run's window
static int sad4_run_waves()
{
    static const int v = [] {
        const char *e = getenv("HAVOC_SAD4_RUN_WAVES");
        const int n = e ? atoi(e) : 4;      // four wavefronts per run: 0.155 ms against 0.172 with two, 1.27 M calls (profiles/r05/sad4_run_policy.txt)
        return n == 1 || n == 2 ? n : 4;
    }();
    return v;
}

hipError_t launch_sad4_runs(hipStream_t st, int S, const void *src, long ss, const void *ref, long rs, const void *jobs, int n, const void *runs, int nruns, int32_t *out)
{
    if (n <= 0 || nruns <= 0) return hipSuccess;
    if (rs < 64 || rs >= (1 << 22) || (S != 1 && S != 2)) return hipErrorInvalidValue;
    const char *s = (const char *)src, *r = (const char *)ref;
    const int32_t *j = (const int32_t *)jobs, *rn = (const int32_t *)runs;
    const float inv = 1.0f / (float)rs;
    const int nw = sad4_run_waves();
    const dim3 g(nruns), b(64 * nw);
    // 0 (default): a lane per candidate; 1 / 2: a lane group of 16 per call, that many calls per pass (the round's first form, kept for comparison: HAVOC_SAD4_RUN_UNROLL)
    static const int unroll = [] { const char *e = getenv("HAVOC_SAD4_RUN_UNROLL"); return e && *e == '2' ? 2 : (e && *e == '1' ? 1 : 0); }();
    static const bool srcl = [] { const char *e = getenv("HAVOC_SAD4_RUN_SRC"); return e && *e == 'l'; }();      // l: the source block by broadcast reads of its LDS copy (default: scalar loads)
#define HAVOC_RUN(SS, NW) do { if (unroll == 0 && !srcl) hipLaunchKernelGGL((k_sad4r<SS, NW, 0, true>), g, b, 0, st, s, ss, r, rs, inv, j, n, rn, nruns, out); \
                               else if (unroll == 0) hipLaunchKernelGGL((k_sad4r<SS, NW, 0, false>), g, b, 0, st, s, ss, r, rs, inv, j, n, rn, nruns, out); \
                               else if (unroll == 1) hipLaunchKernelGGL((k_sad4r<SS, NW, 1, false>), g, b, 0, st, s, ss, r, rs, inv, j, n, rn, nruns, out); \
                               else hipLaunchKernelGGL((k_sad4r<SS, NW, 2, false>), g, b, 0, st, s, ss, r, rs, inv, j, n, rn, nruns, out); } while (0)
    if (S == 1) { if (nw == 1) HAVOC_RUN(1, 1); else if (nw == 2) HAVOC_RUN(1, 2); else HAVOC_RUN(1, 4); }
    else { if (nw == 1) HAVOC_RUN(2, 1); else if (nw == 2) HAVOC_RUN(2, 2); else HAVOC_RUN(2, 4); }
#undef HAVOC_RUN
    return hipGetLastError();
}

hipError_t launch_sad(hipStream_t st, int S, int ways, const void *src, long ss, const void *ref, long rs, const void *jobs, int n, int32_t *out)
{
    if (n <= 0) return hipSuccess;
    const char *s = (const char *)src, *r = (const char *)ref;
    const int32_t *j = (const int32_t *)jobs;
    const dim3 g((n + 256 / kSadLanes - 1) / (256 / kSadLanes)), b(256);
    if (S == 1 && ways == 1) hipLaunchKernelGGL((k_sad<1, 1>), g, b, 0, st, s, ss, r, rs, j, n, out);
    else if (S == 2 && ways == 1) hipLaunchKernelGGL((k_sad<2, 1>), g, b, 0, st, s, ss, r, rs, j, n, out);
    else if (ways == 4 && rs >= 64 && rs < (1 << 22) && sad4_form() == 2)
    {
        const float inv = 1.0f / (float)rs;
        if (S == 1) hipLaunchKernelGGL((k_sad4w<1>), g, b, 0, st, s, ss, r, rs, inv, j, n, out);
        else if (S == 2) hipLaunchKernelGGL((k_sad4w<2>), g, b, 0, st, s, ss, r, rs, inv, j, n, out);
        else return hipErrorInvalidValue;
    }
    else if (S == 1 && ways == 4 && sad4_form() == 0) hipLaunchKernelGGL((k_sad<1, 4, 2, 6>), g, b, 0, st, s, ss, r, rs, j, n, out);
    else if (S == 1 && ways == 4) hipLaunchKernelGGL((k_sad<1, 4>), g, b, 0, st, s, ss, r, rs, j, n, out);
    else if (S == 2 && ways == 4) hipLaunchKernelGGL((k_sad<2, 4>), g, b, 0, st, s, ss, r, rs, j, n, out);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_ssd(hipStream_t st, int S, const void *a, long sa, const void *b, long sb, const void *jobs, int n, uint32_t *out)
{
    if (n <= 0) return hipSuccess;
    const dim3 g((n + 256 / kSadLanes - 1) / (256 / kSadLanes)), b256(256);   // 16 lanes per job
    if (S == 1) hipLaunchKernelGGL((k_ssd<1>), g, b256, 0, st, (const char *)a, sa, (const char *)b, sb, (const int32_t *)jobs, n, out);
    else hipLaunchKernelGGL((k_ssd<2>), g, b256, 0, st, (const char *)a, sa, (const char *)b, sb, (const int32_t *)jobs, n, out);
    return hipGetLastError();
}

hipError_t launch_satd(hipStream_t st, int S, int maxw, int maxh, const void *a, long sa, const void *b, long sb, const void *jobs, int n,
                       int32_t *out)
{
    if (n <= 0) return hipSuccess;
    const char *x = (const char *)a, *y = (const char *)b;
    const int32_t *j = (const int32_t *)jobs;
    // lanes per job: one per 8-sample row of the largest block in the batch (8x8 -> 8 ... 64x64 -> 64, looping 8 times)
    const int rows = ((maxw + 7) / 8) * maxh;
    const int G = rows <= 8 ? 8 : rows <= 16 ? 16 : rows <= 32 ? 32 : 64;
    const dim3 g((n + 256 / G - 1) / (256 / G)), b256(256);
#define SATD_GO(SS, GG) hipLaunchKernelGGL((k_satd<SS, GG>), g, b256, 0, st, x, sa, y, sb, j, n, out)
    if (S == 1) { if (G == 8) SATD_GO(1, 8); else if (G == 16) SATD_GO(1, 16); else if (G == 32) SATD_GO(1, 32); else SATD_GO(1, 64); }
    else { if (G == 8) SATD_GO(2, 8); else if (G == 16) SATD_GO(2, 16); else if (G == 32) SATD_GO(2, 32); else SATD_GO(2, 64); }
#undef SATD_GO
    return hipGetLastError();
}

hipError_t launch_satd_multi(hipStream_t st, int S, int maxw, int maxh, const void *a, long sa, const void *b, long sb, const void *jobs, int n,
                             int32_t *out)
{
    if (n <= 0) return hipSuccess;
    const char *x = (const char *)a, *y = (const char *)b;
    const int32_t *j = (const int32_t *)jobs;
    const int rows = ((maxw + 7) / 8) * maxh;
    const int G = rows <= 8 ? 8 : rows <= 16 ? 16 : rows <= 32 ? 32 : 64;
    // measured on MI355X (184 k candidates of a 1080p frame): 2 candidates per lane group 63 us, 1: 69, 4: 70, 8: 83,
    // all 16: 181 -- wavefronts in flight matter more than fetching the source rows only once
    constexpr int CPS = 2;
    // round 6: classes whose largest block has at least four 8 x 8 tiles take a tile per lane (G = those tiles, a power of two); 4 candidates per lane group
    // instead of 2 measured slower there too (group 0.076 -> 0.097 ms, step 0.642 -> 0.652: gpu call r06v)
    const int tiles = ((maxw + 7) / 8) * ((maxh + 7) / 8);
    static const bool tileOff = getenv("HAVOC_SATD_TILE") && atoi(getenv("HAVOC_SATD_TILE")) == 0;      // diagnostic A/B switch (profiles/)
    if (tiles >= 4 && !tileOff)
    {
        const int GT = tiles <= 4 ? 4 : tiles <= 8 ? 8 : tiles <= 16 ? 16 : tiles <= 32 ? 32 : 64;
        const long groups = (long)n * (kSatdMulti / CPS);
        const dim3 g((unsigned)((groups + 256 / GT - 1) / (256 / GT))), b256(256);
#define SATD_GO_T(SS, GG) hipLaunchKernelGGL((k_satd_multi<SS, GG, CPS, true>), g, b256, 0, st, x, sa, y, sb, j, n, out)
        if (S == 1) { if (GT == 4) SATD_GO_T(1, 4); else if (GT == 8) SATD_GO_T(1, 8); else if (GT == 16) SATD_GO_T(1, 16); else if (GT == 32) SATD_GO_T(1, 32); else SATD_GO_T(1, 64); }
        else { if (GT == 4) SATD_GO_T(2, 4); else if (GT == 8) SATD_GO_T(2, 8); else if (GT == 16) SATD_GO_T(2, 16); else if (GT == 32) SATD_GO_T(2, 32); else SATD_GO_T(2, 64); }
#undef SATD_GO_T
        return hipGetLastError();
    }
    const long groups = (long)n * (kSatdMulti / CPS);
    const dim3 g((unsigned)((groups + 256 / G - 1) / (256 / G))), b256(256);
#define SATD_GO(SS, GG) hipLaunchKernelGGL((k_satd_multi<SS, GG, CPS>), g, b256, 0, st, x, sa, y, sb, j, n, out)
    if (S == 1) { if (G == 8) SATD_GO(1, 8); else if (G == 16) SATD_GO(1, 16); else if (G == 32) SATD_GO(1, 32); else SATD_GO(1, 64); }
    else { if (G == 8) SATD_GO(2, 8); else if (G == 16) SATD_GO(2, 16); else if (G == 32) SATD_GO(2, 32); else SATD_GO(2, 64); }
#undef SATD_GO
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
