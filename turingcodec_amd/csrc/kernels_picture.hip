// Picture-level helpers next to the primitive path.
//
//   k_pad_block : Padding::padBlock (turing/Padding.h:60-97; padImage :33-57 is the four-sides case; called per CTU band
//                 by the deblocking task, turing/TaskDeblock.cpp:151-159, before a reconstructed picture becomes a
//                 reference): replicate the edge samples of a w x h block into a border of `pad` samples.
// Every border sample is the block sample at the clamped coordinates -- which is what the reference's two passes
// (rows first, then copies of the padded first / last row) produce -- so no pass ordering is needed: one thread per
// border sample, reading only the interior.
#include "common.h"

namespace havoc_gpu {

template <int S>
__global__ __launch_bounds__(256) void k_pad_block(char *__restrict__ plane, long origin, int w, int h, long stride, int pad, int top, int bottom,
                                                   int left, int right)
{
    typedef typename Sample<S>::T T;
    T *p = reinterpret_cast<T *>(plane) + origin;
    const int x0 = left ? -pad : 0, wide = w + (left ? pad : 0) + (right ? pad : 0);
    const int nside = (left ? pad : 0) + (right ? pad : 0);          // border samples of an interior row
    const long n_rows = (long)h * nside;                              // left / right borders
    const long n_band = (long)pad * wide;                             // one horizontal band
    const long total = n_rows + (top ? n_band : 0) + (bottom ? n_band : 0);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256)
    {
        int x, y;
        if (i < n_rows)
        {
            y = (int)(i / nside);
            const int k = (int)(i - (long)y * nside);
            x = (left && k < pad) ? k - pad : w + k - (left ? pad : 0);
        }
        else
        {
            long r = i - n_rows;
            const bool isTop = top && r < n_band;
            if (!isTop && top) r -= n_band;
            const int row = (int)(r / wide);
            x = x0 + (int)(r - (long)row * wide);
            y = isTop ? -1 - row : h + row;
        }
        const int xc = min(max(x, 0), w - 1), yc = min(max(y, 0), h - 1);
        p[(long)y * stride + x] = p[(long)yc * stride + xc];
    }
}

hipError_t launch_pad_block(hipStream_t st, int S, void *plane, long origin, int w, int h, long stride, int pad, int top, int bottom, int left,
                            int right)
{
    const long total = (long)h * pad * ((left != 0) + (right != 0)) + (long)pad * (w + 2L * pad) * ((top != 0) + (bottom != 0));
    if (total <= 0) return hipSuccess;
    const int blocks = (int)min(4096L, (total + 255) / 256);
    if (S == 1) hipLaunchKernelGGL(k_pad_block<1>, dim3(blocks), dim3(256), 0, st, (char *)plane, origin, w, h, stride, pad, top, bottom, left, right);
    else hipLaunchKernelGGL(k_pad_block<2>, dim3(blocks), dim3(256), 0, st, (char *)plane, origin, w, h, stride, pad, top, bottom, left, right);
    return hipGetLastError();
}

} // namespace havoc_gpu
