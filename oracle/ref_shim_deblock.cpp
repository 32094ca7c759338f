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

// ---- boundary strengths: the reference's OWN derivation -- LoopFilter::Picture::processCu / processTu / processRc (LoopFilter.h:541-737)
// driven with a picture's units in decoding order, and the prediction-unit rule of processPu (:608-643) through the reference's own
// LoopFilter::sameMotion (:402-422).  processPu itself reads its neighbours through the encoder's Snake cursor, which is encoder state and
// not reproducible from plain arguments: the loop around sameMotion is restated here (same edges, same 4-sample segments, increaseBs), the
// neighbour's PuData coming from a map of the units given.  units: int32 rows
//   cus [n][6] = x0, y0, log2CbSize, intra, QpY, cu_transquant_bypass_flag
//   pus [n][10] = x0, y0, nPbW, nPbH, mvL0 x, y, mvL1 x, y, dpb index L0, L1 (-1: list unused)      (inter units only)
//   tus [n][5] = x0, y0, log2TrafoSize, cbf_luma, intra
struct BsH
{
    typedef void Tag;
    int widthCtbs, heightCtbs;
    const int32_t *cu;          // the unit being processed
    int intraAt;                // CuPredMode answered for the unit being processed
    int operator[](PicWidthInCtbsY) const { return widthCtbs; }
    int operator[](PicHeightInCtbsY) const { return heightCtbs; }
    int operator[](PicSizeInCtbsY) const { return widthCtbs * heightCtbs; }
    int operator[](CtbLog2SizeY) const { return 6; }
    int operator[](slice_deblocking_filter_disabled_flag) const { return 0; }
    int operator[](pcm_loop_filter_disabled_flag) const { return 0; }
    int operator[](cu_transquant_bypass_flag) const { return cu[5]; }
    int operator[](cu_qp_delta_enabled_flag) const { return 0; }
    int operator[](QpY) const { return cu[4]; }
    int operator[](Neighbouring<pcm_flag, Current>) const { return 0; }
    int operator[](Neighbouring<CuPredMode, Current>) const { return intraAt ? MODE_INTRA : MODE_INTER; }
    operator QpState *() const { return nullptr; }
};

extern "C" void ref_derive_bs(int W, int H, const int32_t *cus, int ncu, const int32_t *pus, int npu, const int32_t *tus, int ntu, int8_t *data, uint8_t *bs)
{
    BsH h{(W + 63) / 64, (H + 63) / 64, nullptr, 0};
    LoopFilter::Picture pic(h);
    for (auto &b : pic.blocks) { b.data = 0; b.packedBs = 0; }
    for (int i = 0; i < ncu; ++i)
    {
        h.cu = cus + 6 * i;
        h.intraAt = h.cu[3];
        pic.processCu(h, coding_unit(h.cu[0], h.cu[1], h.cu[2]));
    }
    // prediction units: PuData per 4x4 cell (the state the cursor would hand out for already-decoded neighbours)
    const int cw = W / 4, ch = H / 4;
    std::vector<PuData> map(size_t(cw) * ch);
    for (auto &p : map) { p = PuData(); p.inter.dpbIndexPlus1[0] = p.inter.dpbIndexPlus1[1] = 0; }
    PuData none = PuData();
    none.inter.dpbIndexPlus1[0] = none.inter.dpbIndexPlus1[1] = 0;
    auto fill = [&](const int32_t *p) {
        PuData d = PuData();
        for (int l = 0; l < 2; ++l)
        {
            d.inter.motionVector[l][0] = int16_t(p[4 + 2 * l]);
            d.inter.motionVector[l][1] = int16_t(p[5 + 2 * l]);
            d.inter.dpbIndexPlus1[l] = int8_t(p[8 + l] + 1);
        }
        return d;
    };
    for (int i = 0; i < npu; ++i)
    {
        const int32_t *p = pus + 10 * i;
        const PuData d = fill(p);
        for (int y = p[1] / 4; y < (p[1] + p[3]) / 4; ++y)
            for (int x = p[0] / 4; x < (p[0] + p[2]) / 4; ++x) map[size_t(y) * cw + x] = d;
    }
    for (int i = 0; i < npu; ++i)      // LoopFilter.h:618-641
    {
        const int32_t *p = pus + 10 * i;
        const PuData cur = fill(p);
        if (p[0] % 8 == 0)
            for (int y = p[1]; y < p[1] + p[3]; y += 4)
            {
                const PuData &left = p[0] > 0 ? map[size_t(y / 4) * cw + p[0] / 4 - 1] : none;
                if (!LoopFilter::sameMotion(left, cur)) pic.blockAt(p[0] / 8, y / 8).increaseBs(EDGE_VER, (y / 4) % 2, 1);
            }
        if (p[1] % 8 == 0)
            for (int x = p[0]; x < p[0] + p[2]; x += 4)
            {
                const PuData &above = p[1] > 0 ? map[size_t(p[1] / 4 - 1) * cw + x / 4] : none;
                if (!LoopFilter::sameMotion(above, cur)) pic.blockAt(x / 8, p[1] / 8).increaseBs(EDGE_HOR, (x / 4) % 2, 1);
            }
    }
    for (int i = 0; i < ntu; ++i)
    {
        const int32_t *t = tus + 5 * i;
        h.intraAt = t[4];
        pic.processTu(h, transform_unit(t[0], t[1], t[0], t[1], t[2], 0, 0));
        if (t[3]) pic.processRc(h, residual_coding(t[0], t[1], t[2], 0));
    }
    // processCtu (LoopFilter.h:473-537) is the LAST event of a CTU (turing/Decode.h:281-286: after Read<coding_tree_unit>) and needs the slice /
    // tile address tags of the real handler; what it does to the strengths with one slice per picture -- setBs(.., 0) along the edges whose
    // neighbouring CTU does not exist -- through the reference's own Block::setBs:
    for (int y = 0; y < h.heightCtbs * 8; ++y) { pic.blockAt(0, y).setBs(EDGE_VER, 0, 0); pic.blockAt(0, y).setBs(EDGE_VER, 1, 0); }
    for (int x = 0; x < h.widthCtbs * 8; ++x) { pic.blockAt(x, 0).setBs(EDGE_HOR, 0, 0); pic.blockAt(x, 0).setBs(EDGE_HOR, 1, 0); }
    for (size_t i = 0; i < pic.blocks.size(); ++i) { data[i] = pic.blocks[i].data; bs[i] = pic.blocks[i].packedBs; }
}
