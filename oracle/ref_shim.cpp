// TEST INFRASTRUCTURE ONLY -- never linked into, imported by, or called from the product
// (libhavoc_mi355x.so / turingcodec_amd).  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load the library this file is built into.
//
// ref_shim: a thin extern "C" veneer over the *reference's own* havoc library, compiled from the
// sources where they lie under /root/reference/havoc (see oracle/Makefile; output goes to
// oracle/_ref/libhavoc_ref.so, which is git-ignored).  No reference source is copied: this file only
// #includes the reference headers at build time and calls the reference's populate/get API
// (/root/reference/havoc/havoc.h:132-153, sad.h:57-118, ssd.h:32-52, hadamard.h:31-50,
// pred_inter.h:34-104, pred_intra.h:29-60, transform.h:31-148, quantize.h:34-104, diff.h:28-44).
//
// Two table sets are populated:
//   handle 0 : HAVOC_C_REF | HAVOC_C_OPT   -- the plain-C normative functions ("--asm 0")
//   handle 1 : havoc_instruction_set_support() -- the xbyak x86 JIT path ("--asm 1"), used as the
//              cpu_baseline kind "reference" in bench.py.
#include "havoc.h"
#include "sad.h"
#include "ssd.h"
#include "hadamard.h"
#include "pred_inter.h"
#include "pred_intra.h"
#include "transform.h"
#include "quantize.h"
#include "diff.h"

#include <cstdint>
#include <cstring>

namespace {

template <typename Sample>
struct SampleTables
{
    havoc_table_sad<Sample> sad;
    havoc_table_sad_multiref<Sample> sad4;
    havoc_table_ssd<Sample> ssd;
    havoc_table_hadamard_satd<Sample> satd;
    HavocTablePredUni<Sample> predUni;
    HavocTablePredBi<Sample> predBi;
    havoc::TableSubtractBi<Sample> subtractBi;
    havoc::intra::Table<Sample> intra;
    havoc::table_inverse_transform_add<Sample> invAdd;

    void populate(havoc_code code)
    {
        havoc_populate_sad<Sample>(&sad, code);
        havoc_populate_sad_multiref<Sample>(&sad4, code);
        havoc_populate_ssd<Sample>(&ssd, code);
        havoc_populate_hadamard_satd<Sample>(&satd, code);
        havocPopulatePredUni<Sample>(&predUni, code);
        havocPopulatePredBi<Sample>(&predBi, code);
        havoc::populateSubtractBi<Sample>(&subtractBi, code);
        intra.populate(code);
        havoc::populate_inverse_transform_add<Sample>(&invAdd, code, 1);
    }
};

struct Tables
{
    havoc_code code;
    int mask;
    SampleTables<uint8_t> s8;
    SampleTables<uint16_t> s16;
    havoc::table_inverse_transform inv;
    havoc::table_transform<8> fwd8;
    havoc::table_transform<10> fwd10;
    havoc_table_quantize_inverse dequant;
    havoc_table_quantize quant;
    havoc_table_quantize_reconstruct qrec;
    bool ready;
};

Tables g[2];

template <typename Sample> SampleTables<Sample> &st(Tables &t);
template <> SampleTables<uint8_t> &st<uint8_t>(Tables &t) { return t.s8; }
template <> SampleTables<uint16_t> &st<uint16_t>(Tables &t) { return t.s16; }

Tables &tab(int h)
{
    Tables &t = g[h & 1];
    if (!t.ready)
    {
        std::memset(static_cast<void *>(&t), 0, sizeof(t));
        t.mask = (h & 1) ? (int)havoc_instruction_set_support() : (int)(HAVOC_C_REF | HAVOC_C_OPT);
        t.code = havoc_new_code((havoc_instruction_set)t.mask, 16 << 20);
        t.s8.populate(t.code);
        t.s16.populate(t.code);
        havoc::populate_inverse_transform(&t.inv, t.code, 1);
        havoc::populate_transform<8>(&t.fwd8, t.code);
        havoc::populate_transform<10>(&t.fwd10, t.code);
        havoc_populate_quantize_inverse(&t.dequant, t.code);
        havoc_populate_quantize(&t.quant, t.code);
        havoc_populate_quantize_reconstruct(&t.qrec, t.code);
        t.ready = true;
    }
    return t;
}

template <typename Sample>
int sad(int h, const Sample *src, intptr_t ss, const Sample *ref, intptr_t rs, int w, int ht)
{
    auto f = *havoc_get_sad<Sample>(&st<Sample>(tab(h)).sad, w, ht);
    return f ? f(src, ss, ref, rs, HAVOC_RECT(w, ht)) : -1;
}

template <typename Sample>
int sad4(int h, const Sample *src, intptr_t ss, const Sample *r0, const Sample *r1, const Sample *r2, const Sample *r3, intptr_t rs, int *out, int w, int ht)
{
    auto f = *havoc_get_sad_multiref<Sample>(&st<Sample>(tab(h)).sad4, 4, w, ht);
    if (!f) return -1;
    const Sample *refs[4] = { r0, r1, r2, r3 };
    f(src, ss, refs, rs, out, HAVOC_RECT(w, ht));
    return 0;
}

template <typename Sample>
long long ssd(int h, const Sample *a, intptr_t sa, const Sample *b, intptr_t sb, int log2)
{
    auto f = *havoc_get_ssd<Sample>(&st<Sample>(tab(h)).ssd, log2);
    return f ? (long long)f(a, sa, b, sb, 1 << log2, 1 << log2) : -1;
}

template <typename Sample>
int satd(int h, const Sample *a, intptr_t sa, const Sample *b, intptr_t sb, int log2)
{
    auto f = *havoc_get_hadamard_satd<Sample>(&st<Sample>(tab(h)).satd, log2);
    return f ? f(a, sa, b, sb) : -1;
}

template <typename Sample>
int predUni(int h, Sample *dst, intptr_t sd, const Sample *ref, intptr_t sr, int w, int ht, int xf, int yf, int bd, int taps)
{
    auto f = *havocGetPredUni<Sample>(&st<Sample>(tab(h)).predUni, taps, w, ht, xf, yf, bd);
    if (!f) return -1;
    f(dst, sd, ref, sr, w, ht, xf, yf, bd);
    return 0;
}

template <typename Sample>
int predBi(int h, Sample *dst, intptr_t sd, const Sample *r0, const Sample *r1, intptr_t sr, int w, int ht, int xf0, int yf0, int xf1, int yf1, int bd, int taps)
{
    auto f = *havocGetPredBi<Sample>(&st<Sample>(tab(h)).predBi, taps, w, ht, xf0, yf0, xf1, yf1, bd);
    if (!f) return -1;
    f(dst, sd, r0, r1, sr, w, ht, xf0, yf0, xf1, yf1, bd);
    return 0;
}

template <typename Sample>
int subtractBi(int h, Sample *dst, intptr_t sd, const Sample *pred, intptr_t sp, const Sample *src, intptr_t ss, int w, int ht, int bd)
{
    auto f = st<Sample>(tab(h)).subtractBi.get();
    if (!f) return -1;
    f(dst, sd, pred, sp, src, ss, w, ht, bd);
    return 0;
}

template <typename Sample>
int intra(int h, Sample *dst, intptr_t sd, const Sample *neighbours, int cIdx, int bd, int log2, int mode)
{
    auto f = st<Sample>(tab(h)).intra.lookup(cIdx, bd, log2, mode);
    if (!f) return -1;
    f(dst, sd, neighbours, mode);
    return 0;
}

template <typename Sample>
int invAdd(int h, Sample *dst, intptr_t sd, const Sample *pred, intptr_t sp, const int16_t *coeffs, int bd, int trType, int log2)
{
    auto f = *havoc::get_inverse_transform_add<Sample>(&st<Sample>(tab(h)).invAdd, trType, log2);
    if (!f) return -1;
    f(dst, sd, pred, sp, coeffs, bd);
    return 0;
}

} // namespace

extern "C" {

int ref_mask(int h) { return tab(h).mask; }

int ref_sad_u8(int h, const uint8_t *s, intptr_t ss, const uint8_t *r, intptr_t rs, int w, int ht) { return sad<uint8_t>(h, s, ss, r, rs, w, ht); }
int ref_sad_u16(int h, const uint16_t *s, intptr_t ss, const uint16_t *r, intptr_t rs, int w, int ht) { return sad<uint16_t>(h, s, ss, r, rs, w, ht); }

int ref_sad4_u8(int h, const uint8_t *s, intptr_t ss, const uint8_t *r0, const uint8_t *r1, const uint8_t *r2, const uint8_t *r3, intptr_t rs, int *out, int w, int ht) { return sad4<uint8_t>(h, s, ss, r0, r1, r2, r3, rs, out, w, ht); }
int ref_sad4_u16(int h, const uint16_t *s, intptr_t ss, const uint16_t *r0, const uint16_t *r1, const uint16_t *r2, const uint16_t *r3, intptr_t rs, int *out, int w, int ht) { return sad4<uint16_t>(h, s, ss, r0, r1, r2, r3, rs, out, w, ht); }

long long ref_ssd_u8(int h, const uint8_t *a, intptr_t sa, const uint8_t *b, intptr_t sb, int log2) { return ssd<uint8_t>(h, a, sa, b, sb, log2); }
long long ref_ssd_u16(int h, const uint16_t *a, intptr_t sa, const uint16_t *b, intptr_t sb, int log2) { return ssd<uint16_t>(h, a, sa, b, sb, log2); }

int ref_satd_u8(int h, const uint8_t *a, intptr_t sa, const uint8_t *b, intptr_t sb, int log2) { return satd<uint8_t>(h, a, sa, b, sb, log2); }
int ref_satd_u16(int h, const uint16_t *a, intptr_t sa, const uint16_t *b, intptr_t sb, int log2) { return satd<uint16_t>(h, a, sa, b, sb, log2); }

int ref_pred_uni_u8(int h, uint8_t *d, intptr_t sd, const uint8_t *r, intptr_t sr, int w, int ht, int xf, int yf, int bd, int taps) { return predUni<uint8_t>(h, d, sd, r, sr, w, ht, xf, yf, bd, taps); }
int ref_pred_uni_u16(int h, uint16_t *d, intptr_t sd, const uint16_t *r, intptr_t sr, int w, int ht, int xf, int yf, int bd, int taps) { return predUni<uint16_t>(h, d, sd, r, sr, w, ht, xf, yf, bd, taps); }

int ref_pred_bi_u8(int h, uint8_t *d, intptr_t sd, const uint8_t *r0, const uint8_t *r1, intptr_t sr, int w, int ht, int xf0, int yf0, int xf1, int yf1, int bd, int taps) { return predBi<uint8_t>(h, d, sd, r0, r1, sr, w, ht, xf0, yf0, xf1, yf1, bd, taps); }
int ref_pred_bi_u16(int h, uint16_t *d, intptr_t sd, const uint16_t *r0, const uint16_t *r1, intptr_t sr, int w, int ht, int xf0, int yf0, int xf1, int yf1, int bd, int taps) { return predBi<uint16_t>(h, d, sd, r0, r1, sr, w, ht, xf0, yf0, xf1, yf1, bd, taps); }

int ref_subtract_bi_u8(int h, uint8_t *d, intptr_t sd, const uint8_t *p, intptr_t sp, const uint8_t *s, intptr_t ss, int w, int ht, int bd) { return subtractBi<uint8_t>(h, d, sd, p, sp, s, ss, w, ht, bd); }
int ref_subtract_bi_u16(int h, uint16_t *d, intptr_t sd, const uint16_t *p, intptr_t sp, const uint16_t *s, intptr_t ss, int w, int ht, int bd) { return subtractBi<uint16_t>(h, d, sd, p, sp, s, ss, w, ht, bd); }

int ref_intra_u8(int h, uint8_t *d, intptr_t sd, const uint8_t *nb, int cIdx, int bd, int log2, int mode) { return intra<uint8_t>(h, d, sd, nb, cIdx, bd, log2, mode); }
int ref_intra_u16(int h, uint16_t *d, intptr_t sd, const uint16_t *nb, int cIdx, int bd, int log2, int mode) { return intra<uint16_t>(h, d, sd, nb, cIdx, bd, log2, mode); }

int ref_inverse_transform_add_u8(int h, uint8_t *d, intptr_t sd, const uint8_t *p, intptr_t sp, const int16_t *c, int bd, int trType, int log2) { return invAdd<uint8_t>(h, d, sd, p, sp, c, bd, trType, log2); }
int ref_inverse_transform_add_u16(int h, uint16_t *d, intptr_t sd, const uint16_t *p, intptr_t sp, const int16_t *c, int bd, int trType, int log2) { return invAdd<uint16_t>(h, d, sd, p, sp, c, bd, trType, log2); }

int ref_inverse_transform(int h, int16_t *dst, const int16_t *coeffs, int bd, int trType, int log2)
{
    auto f = *havoc::get_inverse_transform(&tab(h).inv, trType, log2);
    if (!f) return -1;
    f(dst, coeffs, bd);
    return 0;
}

int ref_transform(int h, int16_t *coeffs, const int16_t *src, intptr_t stride, int bd, int trType, int log2)
{
    havoc::Transform *f = bd == 8 ? *havoc::get_transform<8>(&tab(h).fwd8, trType, log2) : *havoc::get_transform<10>(&tab(h).fwd10, trType, log2);
    if (!f) return -1;
    f(coeffs, src, stride);
    return 0;
}

int ref_quantize_inverse(int h, int16_t *dst, const int16_t *src, int scale, int shift, int n)
{
    auto f = *havoc_get_quantize_inverse(&tab(h).dequant, scale, shift);
    if (!f) return -1;
    f(dst, src, scale, shift, n);
    return 0;
}

int ref_quantize(int h, int16_t *dst, const int16_t *src, int scale, int shift, int offset, int n)
{
    auto f = *havoc_get_quantize(&tab(h).quant);
    if (!f) return -0x7fffffff;
    return f(dst, src, scale, shift, offset, n);
}

int ref_quantize_reconstruct(int h, uint8_t *rec, intptr_t sr, const uint8_t *pred, intptr_t sp, const int16_t *res, int log2)
{
    auto f = *havoc_get_quantize_reconstruct(&tab(h).qrec, log2);
    if (!f) return -1;
    f(rec, sr, pred, sp, res, 1 << log2);
    return 0;
}

int ref_ssd_linear(int h, const uint8_t *a, const uint8_t *b, int size)
{
    auto f = havoc_get_ssd_linear(size, tab(h).code);
    return f ? f(a, b, size) : -1;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------------------
// Batch runners for bench.py's cpu_baseline leg: the same job tables the GPU consumes
// (include/havoc_mi355x.h layouts, int32 columns), executed call-by-call through the reference's function
// tables.  Jobs [begin, end) so that host threads can split a table.  `h` selects C (0) or JIT (1) tables.
// ---------------------------------------------------------------------------------------------------------
namespace {

template <typename Sample>
void runSad4(int h, const Sample *src, intptr_t ss, const Sample *ref, intptr_t rs, const int32_t *jobs, int b, int e, int32_t *out)
{
    auto &t = st<Sample>(tab(h));
    for (int i = b; i < e; ++i)
    {
        const int32_t *j = jobs + 8 * i;
        auto f = *havoc_get_sad_multiref<Sample>(&t.sad4, 4, j[5], j[6]);
        const Sample *r[4] = { ref + j[1], ref + j[2], ref + j[3], ref + j[4] };
        f(src + j[0], ss, r, rs, out + 4 * i, HAVOC_RECT(j[5], j[6]));
    }
}

template <typename Sample>
void runSad(int h, const Sample *src, intptr_t ss, const Sample *ref, intptr_t rs, const int32_t *jobs, int b, int e, int32_t *out)
{
    auto &t = st<Sample>(tab(h));
    for (int i = b; i < e; ++i)
    {
        const int32_t *j = jobs + 4 * i;
        out[i] = (*havoc_get_sad<Sample>(&t.sad, j[2], j[3]))(src + j[0], ss, ref + j[1], rs, HAVOC_RECT(j[2], j[3]));
    }
}

template <typename Sample>
void runSsd(int h, const Sample *a, intptr_t sa, const Sample *pb, intptr_t sb, const int32_t *jobs, int b, int e, uint32_t *out)
{
    auto &t = st<Sample>(tab(h));
    for (int i = b; i < e; ++i)
    {
        const int32_t *j = jobs + 4 * i;
        int log2 = 0;
        while ((1 << log2) < j[2]) ++log2;
        out[i] = (*havoc_get_ssd<Sample>(&t.ssd, log2))(a + j[0], sa, pb + j[1], sb, j[2], j[3]);
    }
}

// turing/Measure.h:97-135 tiling, through the reference's satd table
template <typename Sample>
void runSatd(int h, const Sample *a, intptr_t sa, const Sample *pb, intptr_t sb, const int32_t *jobs, int b, int e, int32_t *out)
{
    auto &t = st<Sample>(tab(h));
    for (int i = b; i < e; ++i)
    {
        const int32_t *j = jobs + 4 * i;
        const int w = j[2], ht = j[3];
        const int log2 = ((w | ht) & 3) ? 1 : (((w | ht) & 7) ? 2 : 3);
        const int n = 1 << log2;
        auto f = *havoc_get_hadamard_satd<Sample>(&t.satd, log2);
        int s = 0;
        for (int y = 0; y < ht; y += n)
            for (int x = 0; x < w; x += n)
                s += f(a + j[0] + y * sa + x, sa, pb + j[1] + y * sb + x, sb);
        out[i] = s;
    }
}

template <typename Sample>
void runPredUni(int h, int taps, int bd, Sample *dst, intptr_t sd, const Sample *ref, intptr_t sr, const int32_t *jobs, int b, int e)
{
    auto &t = st<Sample>(tab(h));
    for (int i = b; i < e; ++i)
    {
        const int32_t *j = jobs + 8 * i;
        (*havocGetPredUni<Sample>(&t.predUni, taps, j[2], j[3], j[4], j[5], bd))(dst + j[0], sd, ref + j[1], sr, j[2], j[3], j[4], j[5], bd);
    }
}

template <typename Sample>
void runPredBi(int h, int taps, int bd, Sample *dst, intptr_t sd, const Sample *ref, intptr_t sr, const int32_t *jobs, int b, int e)
{
    auto &t = st<Sample>(tab(h));
    for (int i = b; i < e; ++i)
    {
        const int32_t *j = jobs + 12 * i;
        (*havocGetPredBi<Sample>(&t.predBi, taps, j[3], j[4], j[5], j[6], j[7], j[8], bd))(dst + j[0], sd, ref + j[1], ref + j[2], sr, j[3], j[4],
                                                                                           j[5], j[6], j[7], j[8], bd);
    }
}

template <typename Sample>
void runSubtractBi(int h, int bd, Sample *dst, intptr_t sd, const Sample *pred, intptr_t sp, const Sample *src, intptr_t ss, const int32_t *jobs,
                   int b, int e)
{
    auto f = st<Sample>(tab(h)).subtractBi.get();
    for (int i = b; i < e; ++i)
    {
        const int32_t *j = jobs + 8 * i;
        f(dst + j[0], sd, pred + j[1], sp, src + j[2], ss, j[3], j[4], bd);
    }
}

template <typename Sample>
void runIntra(int h, int bd, int log2, Sample *dst, intptr_t sd, const Sample *nb, const int32_t *jobs, int b, int e)
{
    auto &t = st<Sample>(tab(h));
    for (int i = b; i < e; ++i)
    {
        const int32_t *j = jobs + 8 * i;
        t.intra.lookup(j[4] ? 0 : 1, bd, log2, j[3])(dst + j[0], sd, nb + j[1], j[3]);
    }
}

template <typename Sample>
void runResidual(int16_t *res, intptr_t sres, const int32_t *resOff, const Sample *src, intptr_t ss, const Sample *pred, intptr_t sp,
                 const int32_t *jobs, int b, int e)
{
    for (int i = b; i < e; ++i)   // the reference's inline loop, turing/Reconstruct.cpp:258-260
    {
        const int32_t *j = jobs + 4 * i;
        for (int y = 0; y < j[3]; ++y)
            for (int x = 0; x < j[2]; ++x)
                res[resOff[i] + y * sres + x] = src[j[0] + y * ss + x] - pred[j[1] + y * sp + x];
    }
}

template <typename Sample>
void runInvAdd(int h, int bd, int tr, int log2, Sample *dst, intptr_t sd, const Sample *pred, intptr_t sp, const int16_t *coeffs, const int32_t *jobs,
               int b, int e)
{
    auto f = *havoc::get_inverse_transform_add<Sample>(&st<Sample>(tab(h)).invAdd, tr, log2);
    for (int i = b; i < e; ++i)
    {
        const int32_t *j = jobs + 4 * i;
        f(dst + j[3], sd, pred + j[2], sp, coeffs + j[0], bd);
    }
}

} // namespace

namespace {
// the 35-mode SATD stage of one partition the way the reference runs it: 35 x (table intra prediction into a
// 32-stride stack block, then 8x8 / 4x4 Hadamard tiles against the source) -- turing/Reconstruct.cpp:630-701
template <typename Sample>
void runIntraSatd35(int h, int bd, int log2, const Sample *src, intptr_t ss, const Sample *nb, const int32_t *jobs, int b, int e, int32_t *cost)
{
    auto &t = st<Sample>(tab(h));
    const int n = 1 << log2;
    HAVOC_ALIGN(32, Sample, pred[32 * 32]);
    auto satd = *havoc_get_hadamard_satd<Sample>(&t.satd, log2 == 2 ? 2 : 3);
    const int ts = log2 == 2 ? 4 : 8;
    for (int i = b; i < e; ++i)
    {
        const int32_t *j = jobs + 8 * i;
        const uint64_t mask = (uint64_t)(uint32_t)j[3] | ((uint64_t)(uint32_t)j[4] << 32);
        for (int mode = 0; mode < 35; ++mode)
        {
            t.intra.lookup(j[5] ? 0 : 1, bd, log2, mode)(pred, 32, nb + (((mask >> mode) & 1) ? j[2] : j[1]), mode);
            int c = 0;
            for (int y = 0; y < n; y += ts)
                for (int x = 0; x < n; x += ts) c += satd(src + j[0] + y * ss + x, ss, pred + y * 32 + x, 32);
            cost[35 * i + mode] = c;
        }
    }
}
} // namespace

#define SAMPLE_DISPATCH(S, call8, call16) do { if ((S) == 1) { call8; } else { call16; } } while (0)
typedef const uint8_t *cu8;
typedef const uint16_t *cu16;

extern "C" {

void ref_run_sad4(int h, int S, const void *src, intptr_t ss, const void *ref, intptr_t rs, const int32_t *jobs, int b, int e, int32_t *out)
{
    SAMPLE_DISPATCH(S, runSad4<uint8_t>(h, (cu8)src, ss, (cu8)ref, rs, jobs, b, e, out), runSad4<uint16_t>(h, (cu16)src, ss, (cu16)ref, rs, jobs, b, e, out));
}
void ref_run_sad(int h, int S, const void *src, intptr_t ss, const void *ref, intptr_t rs, const int32_t *jobs, int b, int e, int32_t *out)
{
    SAMPLE_DISPATCH(S, runSad<uint8_t>(h, (cu8)src, ss, (cu8)ref, rs, jobs, b, e, out), runSad<uint16_t>(h, (cu16)src, ss, (cu16)ref, rs, jobs, b, e, out));
}
void ref_run_ssd(int h, int S, const void *a, intptr_t sa, const void *pb, intptr_t sb, const int32_t *jobs, int b, int e, uint32_t *out)
{
    SAMPLE_DISPATCH(S, runSsd<uint8_t>(h, (cu8)a, sa, (cu8)pb, sb, jobs, b, e, out), runSsd<uint16_t>(h, (cu16)a, sa, (cu16)pb, sb, jobs, b, e, out));
}
void ref_run_satd(int h, int S, const void *a, intptr_t sa, const void *pb, intptr_t sb, const int32_t *jobs, int b, int e, int32_t *out)
{
    SAMPLE_DISPATCH(S, runSatd<uint8_t>(h, (cu8)a, sa, (cu8)pb, sb, jobs, b, e, out), runSatd<uint16_t>(h, (cu16)a, sa, (cu16)pb, sb, jobs, b, e, out));
}
void ref_run_pred_uni(int h, int S, int taps, int bd, void *dst, intptr_t sd, const void *ref, intptr_t sr, const int32_t *jobs, int b, int e)
{
    SAMPLE_DISPATCH(S, runPredUni<uint8_t>(h, taps, bd, (uint8_t *)dst, sd, (cu8)ref, sr, jobs, b, e),
                    runPredUni<uint16_t>(h, taps, bd, (uint16_t *)dst, sd, (cu16)ref, sr, jobs, b, e));
}
void ref_run_pred_bi(int h, int S, int taps, int bd, void *dst, intptr_t sd, const void *ref, intptr_t sr, const int32_t *jobs, int b, int e)
{
    SAMPLE_DISPATCH(S, runPredBi<uint8_t>(h, taps, bd, (uint8_t *)dst, sd, (cu8)ref, sr, jobs, b, e),
                    runPredBi<uint16_t>(h, taps, bd, (uint16_t *)dst, sd, (cu16)ref, sr, jobs, b, e));
}
void ref_run_subtract_bi(int h, int S, int bd, void *dst, intptr_t sd, const void *pred, intptr_t sp, const void *src, intptr_t ss, const int32_t *jobs,
                         int b, int e)
{
    SAMPLE_DISPATCH(S, runSubtractBi<uint8_t>(h, bd, (uint8_t *)dst, sd, (cu8)pred, sp, (cu8)src, ss, jobs, b, e),
                    runSubtractBi<uint16_t>(h, bd, (uint16_t *)dst, sd, (cu16)pred, sp, (cu16)src, ss, jobs, b, e));
}
void ref_run_intra(int h, int S, int bd, int log2, void *dst, intptr_t sd, const void *nb, const int32_t *jobs, int b, int e)
{
    SAMPLE_DISPATCH(S, runIntra<uint8_t>(h, bd, log2, (uint8_t *)dst, sd, (cu8)nb, jobs, b, e),
                    runIntra<uint16_t>(h, bd, log2, (uint16_t *)dst, sd, (cu16)nb, jobs, b, e));
}
void ref_run_intra_satd35(int h, int S, int bd, int log2, const void *src, intptr_t ss, const void *nb, const int32_t *jobs, int b, int e, int32_t *cost)
{
    SAMPLE_DISPATCH(S, runIntraSatd35<uint8_t>(h, bd, log2, (cu8)src, ss, (cu8)nb, jobs, b, e, cost),
                    runIntraSatd35<uint16_t>(h, bd, log2, (cu16)src, ss, (cu16)nb, jobs, b, e, cost));
}
void ref_run_residual(int S, int16_t *res, intptr_t sres, const int32_t *resOff, const void *src, intptr_t ss, const void *pred, intptr_t sp,
                      const int32_t *jobs, int b, int e)
{
    SAMPLE_DISPATCH(S, runResidual<uint8_t>(res, sres, resOff, (cu8)src, ss, (cu8)pred, sp, jobs, b, e),
                    runResidual<uint16_t>(res, sres, resOff, (cu16)src, ss, (cu16)pred, sp, jobs, b, e));
}
void ref_run_transform(int h, int bd, int tr, int log2, int16_t *coeffs, const int16_t *res, intptr_t sres, const int32_t *jobs, int b, int e)
{
    havoc::Transform *f = bd == 8 ? *havoc::get_transform<8>(&tab(h).fwd8, tr, log2) : *havoc::get_transform<10>(&tab(h).fwd10, tr, log2);
    for (int i = b; i < e; ++i) f(coeffs + jobs[4 * i], res + jobs[4 * i + 1], sres);
}
void ref_run_quantize_inverse(int h, int16_t *dst, const int16_t *src, const int32_t *jobs, int b, int e)
{
    for (int i = b; i < e; ++i)
    {
        const int32_t *j = jobs + 8 * i;
        (*havoc_get_quantize_inverse(&tab(h).dequant, j[3], j[4]))(dst + j[0], src + j[1], j[3], j[4], j[2]);
    }
}
void ref_run_quantize(int h, int16_t *dst, const int16_t *src, const int32_t *jobs, int b, int e, int32_t *cbf)
{
    auto f = *havoc_get_quantize(&tab(h).quant);
    for (int i = b; i < e; ++i)
    {
        const int32_t *j = jobs + 8 * i;
        cbf[i] = f(dst + j[0], src + j[1], j[3], j[4], j[5], j[2]);
    }
}
void ref_run_inverse_transform_add(int h, int S, int bd, int tr, int log2, void *dst, intptr_t sd, const void *pred, intptr_t sp, const int16_t *coeffs,
                                   const int32_t *jobs, int b, int e)
{
    SAMPLE_DISPATCH(S, runInvAdd<uint8_t>(h, bd, tr, log2, (uint8_t *)dst, sd, (cu8)pred, sp, coeffs, jobs, b, e),
                    runInvAdd<uint16_t>(h, bd, tr, log2, (uint16_t *)dst, sd, (cu16)pred, sp, coeffs, jobs, b, e));
}

} // extern "C"
