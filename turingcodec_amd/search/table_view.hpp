// table_view.hpp -- the per-call View of decision.hpp on top of the reference's own havoc TABLE API (include/havoc/*.h,
// the same headers and table layouts as /root/reference/havoc): every call goes through a function pointer fetched
// with havoc_get_* / havocGetPredUni exactly as the reference's callers fetch it (turing/Search.hpp:1422-1423, 1460-1461,
// 1976-1982; turing/Measure.h:97-135).  Linked against libhavoc_classic.so the calls are served by the MI355X library
// (from SAD surfaces and phase planes when the pictures are registered: INTEGRATION.md); linked against any other build
// of the reference's havoc library they run there -- which is how the tests get their expected values.
#pragma once

#include "decision.hpp"

#include "havoc/hadamard.h"
#include "havoc/pred_inter.h"
#include "havoc/sad.h"

#include <cstddef>

namespace havoc_search {

// the tables StateFunctionTables owns that the motion search uses (turing/StateFunctionTables.h:37-61)
template <typename Sample>
struct MotionTables
{
    havoc_table_sad<Sample> sad;
    havoc_table_sad_multiref<Sample> sad4;
    havoc_table_hadamard_satd<Sample> satd;
    HavocTablePredUni<Sample> predUni;
    havoc::TableSubtractBi<Sample> subtractBi;
    void populate(havoc_code code)
    {
        havoc_populate_sad<Sample>(&sad, code);
        havoc_populate_sad_multiref<Sample>(&sad4, code);
        havoc_populate_hadamard_satd<Sample>(&satd, code);
        havocPopulatePredUni<Sample>(&predUni, code);
        havoc::populateSubtractBi<Sample>(&subtractBi, code);
    }
};

// a plane of a picture in the reference's padded layout: p(0, 0) = first sample of the picture proper
template <typename Sample>
struct Plane
{
    const Sample *origin;
    intptr_t stride;
    const Sample *at(int x, int y) const { return origin + intptr_t(y) * stride + x; }
};

// measureSatd, turing/Measure.h:97-135
template <typename Sample>
int32_t measureSatd(havoc_table_hadamard_satd<Sample> *table, const Sample *a, intptr_t sa, const Sample *b, intptr_t sb, int width, int height)
{
    int32_t satd = 0;
    const int n = ((width | height) & 0x3) ? 2 : (((width | height) & 0x7) ? 4 : 8);
    auto f = *havoc_get_hadamard_satd<Sample>(table, n == 2 ? 1 : (n == 4 ? 2 : 3));
    for (int y = 0; y < height; y += n)
        for (int x = 0; x < width; x += n) satd += f(a + y * sa + x, sa, b + y * sb + x, sb);
    return satd;
}

template <typename Sample>
struct TableView
{
    MotionTables<Sample> &t;
    const Sample *src;          // the PU's source block (the input picture, or the ideal second predictor of a bi search)
    intptr_t srcStride;
    Plane<Sample> ref;          // reference picture, luma
    int x0, y0, w, h, bitDepth;
    havoc_sad<Sample> *fSad;
    havoc_sad_multiref<Sample> *fSad4;

    TableView(MotionTables<Sample> &tables, const Sample *src_, intptr_t srcStride_, Plane<Sample> ref_, int x0_, int y0_, int w_, int h_, int bitDepth_)
        : t(tables), src(src_), srcStride(srcStride_), ref(ref_), x0(x0_), y0(y0_), w(w_), h(h_), bitDepth(bitDepth_)
    {
        fSad = *havoc_get_sad(&t.sad, w, h);
        fSad4 = *havoc_get_sad_multiref(&t.sad4, 4, w, h);
    }

    int sad(int dx, int dy) { return fSad(src, srcStride, ref.at(x0 + dx, y0 + dy), ref.stride, HAVOC_RECT(w, h)); }

    void sad4(const Mv d[4], int32_t out[4])
    {
        const Sample *refs[4];
        for (int i = 0; i < 4; ++i) refs[i] = ref.at(x0 + d[i].x, y0 + d[i].y);
        int sads[4];
        fSad4(src, srcStride, refs, ref.stride, sads, HAVOC_RECT(w, h));
        for (int i = 0; i < 4; ++i) out[i] = sads[i];
    }

    // costDistortionMv's distortion, turing/Search.hpp:1963-1982
    int satdQpel(Mv mv)
    {
        HAVOC_ALIGN(32, Sample, buffer[64 * 64]);
        const int xf = mv.x & 3, yf = mv.y & 3;
        const Sample *r = ref.at(x0 + (mv.x >> 2), y0 + (mv.y >> 2));
        auto *f = *havocGetPredUni(&t.predUni, 8, w, h, xf, yf, bitDepth);
        f(buffer, 64, r, ref.stride, w, h, xf, yf, bitDepth);
        return measureSatd(&t.satd, src, srcStride, buffer, 64, w, h);
    }
};

// the ideal second predictor of a bi search into `ideal` (stride 64): prediction from the OTHER list's vector, then
// SubtractBi against the input block -- turing/Search.hpp:1512-1546.  `limit` is the refined list's LimitFullPelMv.
template <typename Sample>
void makeIdealPredictor(MotionTables<Sample> &t, Sample *ideal, const Sample *input, intptr_t inputStride, Plane<Sample> refOther, Mv mvOther,
                        const LimitFullPelMv &limit, int x0, int y0, int w, int h, int bitDepthY)
{
    HAVOC_ALIGN(32, Sample, other[64 * 64]);
    const int xf = mvOther.x & 3, yf = mvOther.y & 3;
    Mv full = shr2(mvOther);
    limit(full);
    auto *f = *havocGetPredUni(&t.predUni, 8, w, h, xf, yf, bitDepthY);
    f(other, 64, refOther.at(x0 + full.x, y0 + full.y), refOther.stride, w, h, xf, yf, bitDepthY);
    constexpr int bitDepth = 6 + 2 * sizeof(Sample);
    t.subtractBi.get()(ideal, 64, other, 64, input, inputStride, w, h, bitDepth);
}

} // namespace havoc_search
