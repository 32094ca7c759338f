// TEST INFRASTRUCTURE ONLY (see ref_shim.cpp).  The reference's OWN deblocking templates -- LoopFilter::Picture::deblock<EDGE_VER / EDGE_HOR>
// with LumaBlockEdge / ChromaBlockEdge (/root/reference/turing/LoopFilter.h:229-400, 739-777) -- compiled from the reference header
// where it lies (oracle/Makefile, target `ref`) and driven in the CTU order of /root/reference/turing/TaskDeblock.cpp:105-127.
// The templates take the encoder's state through a generic handler `h[Tag()]`; the handler below answers exactly the tags
// deblock() reads (bit depths, picture size in CTUs, chroma QP offsets) from plain arguments.  Boundary strengths and QPs
// -- the encoder's decisions -- come in as the two arrays of LoopFilter::Block.  Nothing of the header is copied.
#include "turing/LoopFilter.h"
#include <cstdint>
struct MiniH
{
    int bitDepthY, bitDepthC, widthCtbs, heightCtbs, log2Ctb, cbOff, crOff, picW, picH;
    int operator[](BitDepthY) const { return bitDepthY; }
    int operator[](BitDepthC) const { return bitDepthC; }
    int operator[](PicOrderCntVal) const { return 0; }
    int operator[](PicWidthInCtbsY) const { return widthCtbs; }
    int operator[](PicHeightInCtbsY) const { return heightCtbs; }
    int operator[](PicSizeInCtbsY) const { return widthCtbs * heightCtbs; }
    int operator[](CtbLog2SizeY) const { return log2Ctb; }
    int operator[](SubWidthC) const { return 2; }
    int operator[](SubHeightC) const { return 2; }
    int operator[](pps_cb_qp_offset) const { return cbOff; }
    int operator[](pps_cr_qp_offset) const { return crOff; }
};
template <typename Sample>
void run(Sample *y, intptr_t sy, Sample *cb, Sample *cr, intptr_t sc, int W, int H, int bd, const int8_t *data, const uint8_t *bs, int tc2, int beta2, int cbOff, int crOff)
{
    MiniH h{bd, bd, (W + 63) / 64, (H + 63) / 64, 6, cbOff, crOff, W, H};
    LoopFilter::Picture pic(h);
    for (size_t i = 0; i < pic.blocks.size(); ++i) { pic.blocks[i].data = data[i]; pic.blocks[i].packedBs = bs[i]; }
    for (auto &c : pic.ctus) { c.tc_offset_div2 = tc2; c.beta_offset_div2 = beta2; }
    Raster<Sample> Y(y, sy), Cb(cb, sc), Cr(cr, sc);
    for (int ry = 0; ry < h.heightCtbs; ++ry)
        for (int rx = 0; rx < h.widthCtbs; ++rx)
        {
            {
                int xBegin = rx << 6, yBegin = ry << 6;
                if (rx) xBegin += 8;
                if (ry) yBegin += 8;
                int xEnd = std::min(((rx + 1) << 6) + 8, W), yEnd = std::min(((ry + 1) << 6) + 8, H);
                pic.deblock<EDGE_VER>(h, Y, Cb, Cr, xBegin, yBegin, xEnd, yEnd);
            }
            {
                int xBegin = rx << 6, yBegin = ry << 6;
                if (ry) yBegin += 8;
                int xEnd = std::min((rx + 1) << 6, W), yEnd = std::min(((ry + 1) << 6) + 8, H);
                pic.deblock<EDGE_HOR>(h, Y, Cb, Cr, xBegin, yBegin, xEnd, yEnd);
            }
        }
}
extern "C" void ref_deblock_u8(uint8_t *y, intptr_t sy, uint8_t *cb, uint8_t *cr, intptr_t sc, int W, int H, int bd, const int8_t *d, const uint8_t *b, int t, int be, int c1, int c2)
{ run<uint8_t>(y, sy, cb, cr, sc, W, H, bd, d, b, t, be, c1, c2); }
extern "C" void ref_deblock_u16(uint16_t *y, intptr_t sy, uint16_t *cb, uint16_t *cr, intptr_t sc, int W, int H, int bd, const int8_t *d, const uint8_t *b, int t, int be, int c1, int c2)
{ run<uint16_t>(y, sy, cb, cr, sc, W, H, bd, d, b, t, be, c1, c2); }
