// What the passes of the RDOQ entry points hand each other through the caller-provided workspace (havoc_mi355x_rdoq_workspace bytes): what a scan
// of a transform block found (which 4x4 groups hold a rounded level > 0 / > 1 / > 2, the block's energy), the histogram of blocks by groups to
// walk, the order the walk takes them in.  Shared by kernels_rdoq.hip (k_rdoq_scan, k_rdoq_hist, k_rdoq_order, the walks) and
// kernels_tu_fused.hip (k_tu_forward<..., SCAN>: the scan done where the coefficients are still in registers).
#pragma once

#include <cstddef>
#include <cstdint>

namespace havoc_gpu {

struct RdoqJob   // == havoc_mi355x_rdoq_job
{
    int32_t dst_off, src_off, quant_scale, quant_shift, inv_scale, lambda_q16, sdh_factor, ctx_index;
    uint8_t c_idx, scan_idx, is_intra, sdh;
    int32_t reserved[3];
};
static_assert(sizeof(RdoqJob) == 48, "rdoq job layout");

struct RdoqInfo { uint64_t mask, mask2, mask3; int64_t sumSq; };      // groups holding a rounded level > 0 / > 1 / > 2 (bit = raster group position), sum of squared coefficients
constexpr int kBins = 66;                               // 0..64 groups to walk (+1 spare)
struct RdoqWork
{
    uint32_t hist[kBins], cursor[kBins];
    uint32_t otherScans, pad[3];      // blocks of this launch that do not use the diagonal scan (walked by the sequential kernel)
    // followed by RdoqInfo info[njobs], then uint32_t order[njobs]
};
__host__ __device__ inline size_t rdoqInfoOffset() { return (sizeof(RdoqWork) + 15) & ~size_t(15); }
__device__ __forceinline__ int groupsToWalk(uint64_t mask) { return mask ? __popcll(mask | 1) : 0; }      // the DC group is always walked

// smallest |coefficient| whose rounded level is >= 1, 2, 3 (Rdoq.cpp:108): |c| * scale + half >= (2k) * half  <=>  |c| >= ceil((2k - 1) * half / scale)
__device__ __forceinline__ void rdoqThresholds(int quantScale, int quantShift, uint32_t (&thr)[3])
{
    const uint32_t half = 1u << (quantShift - 1), scale = (uint32_t)(quantScale > 1 ? quantScale : 1);
    thr[0] = (half + scale - 1) / scale;
    thr[1] = (3 * half + scale - 1) / scale;
    thr[2] = (5 * half + scale - 1) / scale;
}

} // namespace havoc_gpu
