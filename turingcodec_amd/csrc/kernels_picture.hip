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


// ---------------------------------------------------------------------------------------------------------------------
// k_deblock : the in-loop deblocking filter of a whole picture, one pass per edge direction.
//
// Reference: turing/LoopFilter.h -- Block (:52-91: an 8x8 luma region's QpY / filter-disabled bit and its four 2-bit
// boundary strengths), betaTable / tCTable (:217-227), LumaBlockEdge (:229-357), ChromaBlockEdge (:359-400),
// Picture::deblock<edgeType> (:739-777), driven CTU by CTU from turing/TaskDeblock.cpp:105-127.  A filtered edge segment
// changes at most 3 samples each side and reads 4; edges of one direction are 8 samples apart, so within a pass every
// 4-sample segment is independent and the reference's CTU order equals two picture passes: launch EDGE = 0 (vertical
// edges), then EDGE = 1 (horizontal edges, reading the first pass's output).
//
// One thread per 4-sample edge segment, segments of a picture row side by side in a wavefront: for vertical edges a lane
// reads 4 rows of 8 samples around its edge (lanes are 8 samples apart: the wavefront reads whole rows), for horizontal
// edges 8 rows of 4 samples (lanes 4 samples apart).  Decisions (dE, dEp, dEq) on lines 0 and 3, then the strong / normal
// filters per line, in registers; in place.  Chroma (4:2:0) segments follow the luma ones in the same launch: edges on the
// 8-sample chroma grid, strength 2 only, one sample each side.
// ---------------------------------------------------------------------------------------------------------------------
__constant__ uint8_t c_dbk_beta[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 22, 24, 26, 28,
                                       30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64};
__constant__ uint8_t c_dbk_tc[54] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5,
                                     6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24};

struct DeblockArgs
{
    char *luma, *cb, *cr;            // sample (0, 0) of each plane
    long strideY, strideC;           // samples
    int width, height, bitDepth;
    const int8_t *data;              // LoopFilter::Block::data   (QpY << 1 | disabled)
    const uint8_t *bs;               // LoopFilter::Block::packedBs
    int bstride;                     // blocks per grid row
    int tcOffsetDiv2, betaOffsetDiv2, cbQpOffset, crQpOffset;
};

__device__ __forceinline__ int dbk_qpc(int qPi)   // turing/Global.h:1417-1423
{
    if (qPi < 30) return qPi;
    if (qPi > 42) return qPi - 6;
    const unsigned long long lut = 0x8776655443210ull;   // lookup[qPi - 30] - 29, 4 bits each from the low end: 0 1 2 3 4 4 5 5 6 6 7 7 8
    return 29 + (int)((lut >> (4 * (qPi - 30))) & 15);
}

template <int S, int EDGE>
__global__ __launch_bounds__(256) void k_deblock(DeblockArgs a)
{
    typedef typename Sample<S>::T T;
    const int bw = a.width >> 3, bh = a.height >> 3;
    // luma segments: EDGE 0: (x8, row4) with row4 in [0, height/4); EDGE 1: (col4, y8) with col4 in [0, width/4)
    const int lumaAcross = EDGE == 0 ? bw : a.width >> 2;
    const long nLuma = (long)lumaAcross * (EDGE == 0 ? a.height >> 2 : bh);
    // chroma segments per plane: EDGE 0: chroma edge columns x8 even -> bw/2 (rounded up) x bh rows of 4 chroma lines
    const int chromaAcross = EDGE == 0 ? (bw + 1) >> 1 : bw;
    const long nChroma = (long)chromaAcross * (EDGE == 0 ? bh : (bh + 1) >> 1);
    const long i = blockIdx.x * 256L + threadIdx.x;
    const int maxv = (1 << a.bitDepth) - 1;
    const int scale = 1 << (a.bitDepth - 8);
    if (i < nLuma)
    {
        const int r = (int)(i / lumaAcross), c = (int)(i - (long)r * lumaAcross);
        const int x8 = EDGE == 0 ? c : c >> 1, y8 = EDGE == 0 ? r >> 1 : r, pos = EDGE == 0 ? r & 1 : c & 1;
        const long q = (long)a.bstride * y8 + x8;
        const int bS = 3 & (a.bs[q] >> (4 * EDGE + 2 * pos));
        if (!bS) return;
        const long p = EDGE ? q - a.bstride : q - 1;
        const int dq_ = a.data[q], dp_ = a.data[p];
        const bool enQ = !(dq_ & 1), enP = !(dp_ & 1);
        const int qPL = ((dq_ >> 1) + (dp_ >> 1) + 1) >> 1;
        const int beta = c_dbk_beta[clip3(0, 51, qPL + (a.betaOffsetDiv2 << 1))] * scale;
        const int tC = c_dbk_tc[clip3(0, 53, qPL + 2 * (bS - 1) + (a.tcOffsetDiv2 << 1))] * scale;
        // v[k][j]: line k of the segment, j = 0..7 = p3 p2 p1 p0 q0 q1 q2 q3
        int v[4][8];
        T *base = reinterpret_cast<T *>(a.luma) + (EDGE == 0 ? (long)(8 * y8 + 4 * pos) * a.strideY + 8 * x8 - 4 : (long)(8 * y8 - 4) * a.strideY + 8 * x8 + 4 * pos);
        if (EDGE == 0)
        {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 8; ++j) v[k][j] = base[(long)k * a.strideY + j];
        }
        else
        {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k][j] = base[(long)j * a.strideY + k];
        }
        const int dp0 = abs(v[0][1] - 2 * v[0][2] + v[0][3]), dp3 = abs(v[3][1] - 2 * v[3][2] + v[3][3]);
        const int dq0 = abs(v[0][6] - 2 * v[0][5] + v[0][4]), dq3 = abs(v[3][6] - 2 * v[3][5] + v[3][4]);
        const int dpq0 = dp0 + dq0, dpq3 = dp3 + dq3, dp = dp0 + dp3, dq = dq0 + dq3;
        if (dpq0 + dpq3 >= beta) return;
        auto dsam = [&](const int (&l)[8], int dpq) {
            return dpq < (beta >> 2) && abs(l[0] - l[3]) + abs(l[4] - l[7]) < (beta >> 3) && abs(l[3] - l[4]) < ((5 * tC + 1) >> 1);
        };
        const bool strong = dsam(v[0], 2 * dpq0) && dsam(v[3], 2 * dpq3);
        const bool dEp = dp < ((beta + (beta >> 1)) >> 3), dEq = dq < ((beta + (beta >> 1)) >> 3);
#pragma unroll
        for (int k = 0; k < 4; ++k)
        {
            const int p3 = v[k][0], p2 = v[k][1], p1 = v[k][2], p0 = v[k][3], q0 = v[k][4], q1 = v[k][5], q2 = v[k][6], q3 = v[k][7];
            int o[8] = {p3, p2, p1, p0, q0, q1, q2, q3};
            if (strong)
            {
                if (enP)
                {
                    o[3] = clip3(p0 - 2 * tC, p0 + 2 * tC, (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
                    o[2] = clip3(p1 - 2 * tC, p1 + 2 * tC, (p2 + p1 + p0 + q0 + 2) >> 2);
                    o[1] = clip3(p2 - 2 * tC, p2 + 2 * tC, (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
                }
                if (enQ)
                {
                    o[4] = clip3(q0 - 2 * tC, q0 + 2 * tC, (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
                    o[5] = clip3(q1 - 2 * tC, q1 + 2 * tC, (p0 + q0 + q1 + q2 + 2) >> 2);
                    o[6] = clip3(q2 - 2 * tC, q2 + 2 * tC, (p0 + q0 + q1 + 3 * q2 + 2 * q3 + 4) >> 3);
                }
            }
            else
            {
                int delta = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
                if (abs(delta) < tC * 10)
                {
                    delta = clip3(-tC, tC, delta);
                    if (enP) o[3] = clip3(0, maxv, p0 + delta);
                    if (enQ) o[4] = clip3(0, maxv, q0 - delta);
                    if (dEp && enP) o[2] = clip3(0, maxv, p1 + clip3(-(tC >> 1), tC >> 1, (((p2 + p0 + 1) >> 1) - p1 + delta) >> 1));
                    if (dEq && enQ) o[5] = clip3(0, maxv, q1 + clip3(-(tC >> 1), tC >> 1, (((q2 + q0 + 1) >> 1) - q1 - delta) >> 1));
                }
            }
#pragma unroll
            for (int j = 1; j < 7; ++j)   // p3 / q3 never change
                if (o[j] != v[k][j]) base[EDGE == 0 ? (long)k * a.strideY + j : (long)j * a.strideY + k] = (T)o[j];
        }
        return;
    }
    // ---- chroma: one thread per (plane, 4-sample segment)
    long ci = i - nLuma;
    if (ci >= 2 * nChroma) return;
    const int plane = ci >= nChroma;
    if (plane) ci -= nChroma;
    const int r = (int)(ci / chromaAcross), c = (int)(ci - (long)r * chromaAcross);
    const int x8 = EDGE == 0 ? 2 * c : c, y8 = EDGE == 0 ? r : 2 * r;
    if (x8 >= bw || y8 >= bh) return;
    const long q = (long)a.bstride * y8 + x8;
    if ((3 & (a.bs[q] >> (4 * EDGE))) != 2) return;   // position 0's strength decides for the whole chroma segment (LoopFilter.h:366, 385)
    const long p = EDGE ? q - a.bstride : q - 1;
    const int dq_ = a.data[q], dp_ = a.data[p];
    const bool enQ = !(dq_ & 1), enP = !(dp_ & 1);
    const int qPi = (((dq_ >> 1) + (dp_ >> 1) + 1) >> 1) + (plane ? a.crQpOffset : a.cbQpOffset);
    const int tC = c_dbk_tc[clip3(0, 53, dbk_qpc(qPi) + 2 + (a.tcOffsetDiv2 << 1))] * scale;
    T *s = reinterpret_cast<T *>(plane ? a.cr : a.cb) + (long)(4 * y8) * a.strideC + 4 * x8;   // q0 of line 0
    const long across = EDGE ? a.strideC : 1, along = EDGE ? 1 : a.strideC;
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
        T *l = s + k * along;
        const int p1 = l[-2 * across], p0 = l[-across], q0 = l[0], q1 = l[across];
        const int delta = clip3(-tC, tC, ((((q0 - p0) << 2) + p1 - q1 + 4) >> 3));
        if (enP) l[-across] = (T)clip3(0, maxv, p0 + delta);
        if (enQ) l[0] = (T)clip3(0, maxv, q0 - delta);
    }
}

// ---- boundary strengths from the block structure (LoopFilter::Picture::processCu / Pu / Tu / Rc, turing/LoopFilter.h:541-737) ----
struct Cell { int16_t mv[2][2]; int8_t dpb[2]; uint8_t flags; int8_t qp; uint8_t tuLog2; uint8_t pad[3]; };
static_assert(sizeof(Cell) == 16, "havoc_mi355x_cell");

// LoopFilter.h:402-409
__device__ __forceinline__ bool sameMotion1(const Cell &a, int la, const Cell &b, int lb)
{
    if (a.dpb[la] != b.dpb[lb]) return false;
    if (a.dpb[la] < 0) return true;
    return abs(a.mv[la][0] - b.mv[lb][0]) < 4 && abs(a.mv[la][1] - b.mv[lb][1]) < 4;
}
// LoopFilter.h:411-422
__device__ __forceinline__ bool sameMotion(const Cell &a, const Cell &b)
{
    if (sameMotion1(a, 0, b, 0) && sameMotion1(a, 1, b, 1)) return true;
    return sameMotion1(a, 0, b, 1) && sameMotion1(a, 1, b, 0);
}

// strength of the 4-sample edge segment between cell a (left / above; null outside the picture) and cell b (right / below; null outside),
// `pos` = the edge's coordinate across it (a multiple of 8), puEdge = b lies on that edge of its prediction unit
__device__ __forceinline__ int edgeStrength(const Cell *a, const Cell *b, int pos, bool puEdge)
{
    int bs = 0;
    if (a && (pos & ((1 << a->tuLog2) - 1)) == 0)      // the right / bottom edge of a's transform block
        bs = max(bs, (a->flags & HAVOC_CELL_INTRA) ? 2 : ((a->flags & HAVOC_CELL_CODED) ? 1 : 0));
    if (b && (pos & ((1 << b->tuLog2) - 1)) == 0)      // the left / top edge of b's
        bs = max(bs, (b->flags & HAVOC_CELL_INTRA) ? 2 : ((b->flags & HAVOC_CELL_CODED) ? 1 : 0));
    if (b && puEdge && !(b->flags & HAVOC_CELL_INTRA))
    {
        Cell none;
        none.dpb[0] = none.dpb[1] = -1;
        none.mv[0][0] = none.mv[0][1] = none.mv[1][0] = none.mv[1][1] = 0;
        if (!sameMotion(a ? *a : none, *b)) bs = max(bs, 1);
    }
    return bs;
}

// one thread per 8x8 region of the grid (incl. the extra column / row and the part of the last CTUs beyond the picture)
__global__ __launch_bounds__(256) void k_derive_bs(const Cell *__restrict__ cells, long cstride, int width, int height, int gridW, int gridH,
                                                   int8_t *__restrict__ data, uint8_t *__restrict__ bsOut)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= gridW * gridH) return;
    const int rx = i % gridW, ry = i / gridW, x = rx * 8, y = ry * 8;
    const int cw = width >> 2, ch = height >> 2;
    auto at = [&](int cx, int cy) -> const Cell * { return (cx >= 0 && cy >= 0 && cx < cw && cy < ch) ? cells + (long)cy * cstride + cx : nullptr; };
    int packed = 0;
    int8_t d = 0;
    if (x <= width && y <= height)
    {
        for (int k = 0; k < 2; ++k)
        {
            // vertical edge at x, rows y + 4k .. +3
            const Cell *a = at(rx * 2 - 1, ry * 2 + k), *b = at(rx * 2, ry * 2 + k);
            if (a || b) packed |= edgeStrength(a, b, x, b && (b->flags & HAVOC_CELL_PU_LEFT)) << (2 * k);
            // horizontal edge at y, columns x + 4k .. +3
            a = at(rx * 2 + k, ry * 2 - 1);
            b = at(rx * 2 + k, ry * 2);
            if (a || b) packed |= edgeStrength(a, b, y, b && (b->flags & HAVOC_CELL_PU_TOP)) << (4 + 2 * k);
        }
        // processCtu runs AFTER a CTU's units (turing/Decode.h:281-286) and clears the strengths of edges whose neighbouring CTU is not
        // available (LoopFilter.h:484-510): with one slice, the picture's left column and top row
        if (rx == 0) packed &= 0xF0;
        if (ry == 0) packed &= 0x0F;
        if (const Cell *c = at(rx * 2, ry * 2)) d = (int8_t)((c->qp << 1) | ((c->flags & HAVOC_CELL_NO_FILTER) ? 1 : 0));
    }
    data[i] = d;
    bsOut[i] = (uint8_t)packed;
}

hipError_t launch_derive_bs(hipStream_t st, const void *cells, long cstride, int width, int height, int8_t *data, uint8_t *bs)
{
    const int gridW = (width + 63) / 64 * 8 + 1, gridH = (height + 63) / 64 * 8 + 1;
    hipLaunchKernelGGL(k_derive_bs, dim3((gridW * gridH + 255) / 256), dim3(256), 0, st, (const Cell *)cells, cstride, width, height, gridW, gridH, data, bs);
    return hipGetLastError();
}

hipError_t launch_deblock(hipStream_t st, int S, int bitDepth, void *luma, long strideY, void *cb, void *cr, long strideC, int width, int height,
                          const int8_t *data, const uint8_t *bs, int tcOffsetDiv2, int betaOffsetDiv2, int cbQpOffset, int crQpOffset)
{
    DeblockArgs a{(char *)luma, (char *)cb, (char *)cr, strideY, strideC, width, height, bitDepth, data, bs, ((width + 63) / 64) * 8 + 1,
                  tcOffsetDiv2, betaOffsetDiv2, cbQpOffset, crQpOffset};
    const int bw = width >> 3, bh = height >> 3;
    for (int edge = 0; edge < 2; ++edge)
    {
        const long nLuma = edge == 0 ? (long)bw * (height >> 2) : (long)(width >> 2) * bh;
        const long nChroma = edge == 0 ? (long)((bw + 1) >> 1) * bh : (long)bw * ((bh + 1) >> 1);
        const long total = nLuma + 2 * nChroma;
        if (total <= 0) continue;
        const dim3 g((unsigned)((total + 255) / 256)), b(256);
        if (S == 1) { if (edge == 0) hipLaunchKernelGGL((k_deblock<1, 0>), g, b, 0, st, a); else hipLaunchKernelGGL((k_deblock<1, 1>), g, b, 0, st, a); }
        else { if (edge == 0) hipLaunchKernelGGL((k_deblock<2, 0>), g, b, 0, st, a); else hipLaunchKernelGGL((k_deblock<2, 1>), g, b, 0, st, a); }
    }
    return hipGetLastError();
}

} // namespace havoc_gpu
