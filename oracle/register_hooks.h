// TEST INFRASTRUCTURE ONLY (see oracle/README.md).  The two calls a host encoder adds to use libhavoc_classic.so's precompute-and-serve layer
// (include/havoc_classic_ext.h), written against the reference encoder's own state so that oracle/Makefile (target `hooked`) can
// build `turing_ref_hooked`: the reference encoder over libhavoc_classic.so WITH its pictures registered.  The calls are inserted into temporary
// copies of turing/TaskEncodeInput.cpp (a picture starts: its input picture is registered as a SOURCE) and turing/TaskSao.cpp (the picture is
// completely reconstructed, deblocked and padded: it is registered as a REFERENCE unless it is a sub-layer non-reference picture) -- see
// hook_points.txt; no reference text is stored here.  tests/test_reference_encoder.py requires the hooked encoder to write the reference's stream
// with served table calls > 0.
#pragma once
// the declarations of include/havoc_classic_ext.h, repeated here because that header pulls in OUR copy of the table headers and this translation unit
// already holds the reference's own (same ABI, but one definition per translation unit)
extern "C" {
#define HAVOC_PICTURE_SOURCE 0
#define HAVOC_PICTURE_REFERENCE 1
int havoc_classic_register_picture(havoc_code code, const void *origin, intptr_t stride, int width, int height, int pad, int S, int bit_depth, int role);
int havoc_classic_unregister_picture(havoc_code code, const void *origin);
}

#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <vector>

namespace havoc_hooked {

struct Registered
{
    const void *origin;
    std::weak_ptr<void> alive;      // the picture object that owns the plane
};
inline std::mutex &mu() { static std::mutex m; return m; }
inline std::vector<Registered> &registered() { static std::vector<Registered> v; return v; }

// planes whose picture object is gone are unregistered BEFORE the memory can meet a table call again (a new picture's planes are allocated before this
// runs, its searches start after)
inline void forgetFreedPictures(havoc_code code)
{
    auto &v = registered();
    for (size_t i = 0; i < v.size();)
        if (v[i].alive.expired())
        {
            havoc_classic_unregister_picture(code, v[i].origin);
            v[i] = v.back();
            v.pop_back();
        }
        else
            ++i;
}

inline void add(havoc_code code, const void *origin, intptr_t stride, int width, int height, int pad, int S, int bitDepth, int role, std::shared_ptr<void> owner)
{
    std::lock_guard<std::mutex> lock(mu());
    forgetFreedPictures(code);
    auto &v = registered();
    for (size_t i = 0; i < v.size(); ++i)
        if (v[i].origin == origin)
        {   // the same plane again (a picture object reused): the old contents go
            havoc_classic_unregister_picture(code, origin);
            v[i] = v.back();
            v.pop_back();
            break;
        }
    const int rc = havoc_classic_register_picture(code, origin, stride, width, height, pad, S, bitDepth, role);
    if (rc)
    {
        std::fprintf(stderr, "havoc_classic_register_picture failed (%d)\n", rc);
        std::abort();
    }
    v.push_back(Registered{origin, owner});
}

template <typename Sample, class H, class Docket>
void pictureStarts(H &h, const Docket &docket)
{
    StateFunctionTables *tables = h;
    auto &input = static_cast<PictureWrap<Sample> &>(*docket->picture);
    // all three planes of the input picture (round 6: the intra candidates of Cb / Cr are measured against their source blocks like luma's)
    for (int cIdx = 0; cIdx < 3; ++cIdx)
        add(tables->code, input[cIdx].p, input[cIdx].stride, input[cIdx].width, input[cIdx].height, 0, int(sizeof(Sample)), cIdx ? h[BitDepthC()] : h[BitDepthY()],
            HAVOC_PICTURE_SOURCE, docket->picture);
}

template <typename Sample, class H>
void pictureReconstructed(H &h)
{
    if (isSubLayerNonReferencePicture(h[nal_unit_type()]) || h[sample_adaptive_offset_enabled_flag()]) return;
    StateFunctionTables *tables = h;
    StateReconstructedPicture<Sample> *rec = h;
    auto &luma = (*rec->picture)[0];
    // 80 of the 96 allocated border samples are filled (turing/TaskDeblock.cpp:151-167: `const int pad = 80`)
    add(tables->code, luma.p, luma.stride, luma.width, luma.height, 80, int(sizeof(Sample)), h[BitDepthY()], HAVOC_PICTURE_REFERENCE, rec->picture);
}

} // namespace havoc_hooked

#define HAVOC_HOOK_PICTURE_STARTS() havoc_hooked::pictureStarts<Sample>(h, docket)
#define HAVOC_HOOK_PICTURE_RECONSTRUCTED() havoc_hooked::pictureReconstructed<Sample>(h)
