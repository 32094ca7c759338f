// TEST INFRASTRUCTURE ONLY (see ref_shim.cpp).  The reference's OWN sample-adaptive-offset primitives: the two filters of
// turing/sao.cpp (sao_filter_band / sao_filter_edge, declared in turing/sao.h; compiled from where it lies, oracle/Makefile) and the
// statistics the encoder's SAO decision is made from, the static templates of turing/EncSao.h:62-283 (compiled from the header).
// The RD decision itself (EncSao.h:286-1125, floating point) and the CTU availability rules of LoopFilter.h:886-1008 are
// encoder control and stay outside, like the boundary-strength derivation of the deblocking filter.
#include "turing/StateEncode.h"
#include "turing/EncSao.h"
#include "turing/sao.h"
#include <cstdint>

template <typename Sample>
static void stats(const Sample *src, intptr_t ss, const Sample *rec, intptr_t rs, int w, int h, int shift, int64_t *out)
{
    // out: 4 classes x (E[5], num[5]), then band E[32], num[32], then the band start the luma statistics function returns
    EncSao::edge_offset_stats_class0<Sample>(src, ss, rec, rs, out + 0, out + 5, h, w);
    EncSao::edge_offset_stats_class1<Sample>(src, ss, rec, rs, out + 10, out + 15, h, w);
    EncSao::edge_offset_stats_class2<Sample>(src, ss, rec, rs, out + 20, out + 25, h, w);
    EncSao::edge_offset_stats_class3<Sample>(src, ss, rec, rs, out + 30, out + 35, h, w);
    for (int k = 0; k < 64; ++k) out[40 + k] = 0;
    out[104] = EncSao::band_offset_luma_stats<Sample>(src, ss, rec, rs, out + 40, out + 72, h, w, shift);
}

extern "C" void ref_sao_stats_u8(const uint8_t *src, intptr_t ss, const uint8_t *rec, intptr_t rs, int w, int h, int shift, int64_t *out)
{ stats<uint8_t>(src, ss, rec, rs, w, h, shift, out); }
extern "C" void ref_sao_stats_u16(const uint16_t *src, intptr_t ss, const uint16_t *rec, intptr_t rs, int w, int h, int shift, int64_t *out)
{ stats<uint16_t>(src, ss, rec, rs, w, h, shift, out); }

// U and V together, as the encoder calls it (EncSao.h:62-109): one histogram for both planes
extern "C" int ref_sao_band_chroma_u8(const uint8_t *su, const uint8_t *sv, intptr_t ss, const uint8_t *ru, const uint8_t *rv, intptr_t rs, int w, int h, int shift,
                                      int64_t *E, int64_t *num)
{ return EncSao::band_offset_chroma_stats<uint8_t>(su, sv, ss, ru, rv, rs, E, num, h, w, shift); }

extern "C" int ref_sao_band_chroma_u16(const uint16_t *su, const uint16_t *sv, intptr_t ss, const uint16_t *ru, const uint16_t *rv, intptr_t rs, int w, int h, int shift,
                                       int64_t *E, int64_t *num)
{ return EncSao::band_offset_chroma_stats<uint16_t>(su, sv, ss, ru, rv, rs, E, num, h, w, shift); }

extern "C" void ref_sao_band_u8(uint8_t *dst, intptr_t ds, const uint8_t *src, intptr_t ss, int w, int h, const int16_t *table, int bd)
{ sao_filter_band<uint8_t>(dst, ds, src, ss, w, h, table, bd); }
extern "C" void ref_sao_band_u16(uint16_t *dst, intptr_t ds, const uint16_t *src, intptr_t ss, int w, int h, const int16_t *table, int bd)
{ sao_filter_band<uint16_t>(dst, ds, src, ss, w, h, table, bd); }
extern "C" void ref_sao_edge_u8(uint8_t *dst, intptr_t ds, const uint8_t *src, intptr_t ss, int w, int h, const int16_t *offsets, int eoClass, int bd)
{ sao_filter_edge<uint8_t>(dst, ds, src, ss, w, h, offsets, eoClass, bd); }
extern "C" void ref_sao_edge_u16(uint16_t *dst, intptr_t ds, const uint16_t *src, intptr_t ss, int w, int h, const int16_t *offsets, int eoClass, int bd)
{ sao_filter_edge<uint16_t>(dst, ds, src, ss, w, h, offsets, eoClass, bd); }
