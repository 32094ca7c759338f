/*
 * havoc_tables.hpp -- the classic per-block `havoc` table API of the Turing encoder, re-declared for the MI355X
 * implementation so that code written against the reference headers (turing/StateFunctionTables.h:37-102,
 * Search.hpp, Reconstruct.cpp, Dsp.h, Measure.h, LoopFilter.h, Decode.h) compiles unchanged and links against
 * libhavoc_classic.so + libhavoc_mi355x.so instead of libhavoc.a.
 *
 * What is ABI here: function-pointer types, table struct layouts, the inline `get` accessors, and
 * the names of the populate / new_code / delete_code entry points.  Each block cites the reference declaration it
 * replaces.  Every table entry populated by this library runs on the GPU: a per-block call stages its operands to
 * HBM, launches the matching batch kernel of include/havoc_mi355x.h with one job, and waits.  That is
 * bit-exact and thread-safe but launch-latency bound -- throughput callers use the batch API directly
 * (INTEGRATION.md).  There is no CPU implementation behind any mask bit.
 */
#ifndef HAVOC_TABLES_HPP
#define HAVOC_TABLES_HPP

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

/* ---- havoc/havoc.h:72-100: timestamp + alignment helpers used by turing/Profiler.h and stack buffers ------- */
#if defined(__x86_64__) || defined(__i386__)
typedef uint64_t havoc_timestamp;
static inline havoc_timestamp havoc_get_timestamp(void)
{
    unsigned lo, hi;
    __asm__ __volatile__("rdtsc" : "=a"(lo), "=d"(hi));
    return ((havoc_timestamp)hi << 32) | lo;
}
#else
typedef uint64_t havoc_timestamp;
static inline havoc_timestamp havoc_get_timestamp(void) { return 0; }
#endif
#define HAVOC_ALIGN(n, T, v) T v __attribute__((aligned(n)))
#define HAVOC_RECT(width, height) (((width) << 8) | (height)) /* havoc/havoc.h:156 */

#ifdef __cplusplus
extern "C" {
#endif

/* ---- havoc/havoc.h:107-153.  Bits 0..10 keep the reference's values so existing masks still parse; bit 11 is
 * this implementation.  Whatever mask is passed to havoc_new_code, the populated functions run on the GPU. */
typedef enum
{
    HAVOC_NONE = 0,
    HAVOC_C_REF = 1 << 0,
    HAVOC_C_OPT = 1 << 1,
    HAVOC_SSE2 = 1 << 2,
    HAVOC_SSE3 = 1 << 3,
    HAVOC_SSSE3 = 1 << 4,
    HAVOC_SSE41 = 1 << 5,
    HAVOC_SSE42 = 1 << 6,
    HAVOC_LZCNT = 1 << 7,
    HAVOC_POPCNT = 1 << 8,
    HAVOC_AVX = 1 << 9,
    HAVOC_AVX2 = 1 << 10,
    HAVOC_GFX950 = 1 << 11
} havoc_instruction_set;

havoc_instruction_set havoc_instruction_set_support(void);
void havoc_print_instruction_set_support(FILE *f, havoc_instruction_set mask);

typedef struct
{
    void *implementation; /* here: the process-wide GPU binding (device index, per-thread staging contexts) */
} havoc_code;

havoc_code havoc_new_code(havoc_instruction_set mask, int size); /* aborts loudly if there is no gfx950 device */
void havoc_delete_code(havoc_code);
int havoc_main(int argc, const char *argv[]); /* self-check: every table entry non-null and a smoke call each */

typedef void havoc_test_function(int *error_count, havoc_instruction_set mask);

/* ---- havoc/quantize.h:40-99 ------------------------------------------------------------------------------ */
typedef void havoc_quantize_inverse(int16_t *dst, const int16_t *src, int scale, int shift, int n);
typedef struct { havoc_quantize_inverse *p[2]; } havoc_table_quantize_inverse;
static inline havoc_quantize_inverse **havoc_get_quantize_inverse(havoc_table_quantize_inverse *table, int scale, int shift)
{
    return &table->p[!!(scale & ((1 << shift) - 1))];
}
void havoc_populate_quantize_inverse(havoc_table_quantize_inverse *table, havoc_code code);

typedef int havoc_quantize(int16_t *dst, const int16_t *src, int scale, int shift, int offset, int n);
typedef struct { havoc_quantize *p; } havoc_table_quantize;
static inline havoc_quantize **havoc_get_quantize(havoc_table_quantize *table) { return &table->p; }
void havoc_populate_quantize(havoc_table_quantize *table, havoc_code code);

typedef void havoc_quantize_reconstruct(uint8_t *rec, intptr_t stride_rec, const uint8_t *pred, intptr_t stride_pred, const int16_t *res, int n);
typedef struct { havoc_quantize_reconstruct *p[4]; } havoc_table_quantize_reconstruct;
static inline havoc_quantize_reconstruct **havoc_get_quantize_reconstruct(havoc_table_quantize_reconstruct *table, int log2TrafoSize)
{
    return &table->p[log2TrafoSize - 2];
}
void havoc_populate_quantize_reconstruct(havoc_table_quantize_reconstruct *table, havoc_code code);

/* ---- havoc/diff.h:33-39 ---------------------------------------------------------------------------------- */
typedef int havoc_ssd_linear(const uint8_t *src0, const uint8_t *src1, int size);
havoc_ssd_linear *havoc_get_ssd_linear(int size, havoc_code code);

#ifdef __cplusplus
} /* extern "C" */

/* ---- havoc/sad.h:28-118 ---------------------------------------------------------------------------------- */
template <typename Sample>
using havoc_sad = int(const Sample *src, intptr_t stride_src, const Sample *ref, intptr_t stride_ref, uint32_t rect);

/* member order is ABI: the 23 HEVC PU sizes in the reference's order, then the generic entry */
#define HAVOC_PU_SIZE_LIST(X) \
    X(64, 64) X(64, 48) X(64, 32) X(64, 16) X(48, 64) X(32, 64) X(32, 32) X(32, 24) X(32, 16) X(32, 8) X(24, 32) X(16, 64) \
    X(16, 32) X(16, 16) X(16, 12) X(16, 8) X(16, 4) X(12, 16) X(8, 32) X(8, 16) X(8, 8) X(8, 4) X(4, 8)

template <typename Sample>
struct havoc_table_sad
{
#define HAVOC_DECLARE_ENTRY(w, h) havoc_sad<Sample> *sad##w##x##h;
    HAVOC_PU_SIZE_LIST(HAVOC_DECLARE_ENTRY)
#undef HAVOC_DECLARE_ENTRY
    havoc_sad<Sample> *sadGeneric;
};

template <typename Sample>
static havoc_sad<Sample> **havoc_get_sad(havoc_table_sad<Sample> *table, int width, int height)
{
    switch (HAVOC_RECT(width, height))
    {
#define HAVOC_CASE_ENTRY(w, h) case HAVOC_RECT(w, h): return &table->sad##w##x##h;
        HAVOC_PU_SIZE_LIST(HAVOC_CASE_ENTRY)
#undef HAVOC_CASE_ENTRY
    default: break;
    }
    return &table->sadGeneric;
}
template <typename Sample> void havoc_populate_sad(havoc_table_sad<Sample> *table, havoc_code code);

template <typename Sample>
using havoc_sad_multiref = void(const Sample *src, intptr_t stride_src, const Sample *ref[], intptr_t stride_ref, int sad[], uint32_t rect);
template <typename Sample>
struct havoc_table_sad_multiref
{
    havoc_sad_multiref<Sample> *lookup[16][16];
    havoc_sad_multiref<Sample> *sadGeneric_4;
};
template <typename Sample>
havoc_sad_multiref<Sample> **havoc_get_sad_multiref(havoc_table_sad_multiref<Sample> *table, int ways, int width, int height)
{
    if (ways != 4) return 0;
    return &table->lookup[(width >> 2) - 1][(height >> 2) - 1];
}
template <typename Sample> void havoc_populate_sad_multiref(havoc_table_sad_multiref<Sample> *table, havoc_code code);

/* ---- havoc/ssd.h:32-52 ----------------------------------------------------------------------------------- */
template <typename Sample>
using havoc_ssd = uint32_t(Sample const *srcA, intptr_t stride_srcA, Sample const *srcB, intptr_t stride_srcB, int w, int h);
template <typename Sample> struct havoc_table_ssd { havoc_ssd<Sample> *ssd[5]; };
template <typename Sample>
static havoc_ssd<Sample> **havoc_get_ssd(havoc_table_ssd<Sample> *table, int log2TrafoSize) { return &table->ssd[log2TrafoSize - 2]; }
template <typename Sample> void havoc_populate_ssd(havoc_table_ssd<Sample> *table, havoc_code code);

/* ---- havoc/hadamard.h:31-50 ------------------------------------------------------------------------------ */
template <typename Sample>
using havoc_hadamard_satd = int(Sample const *srcA, intptr_t stride_srcA, Sample const *srcB, intptr_t stride_srcB);
template <typename Sample> struct havoc_table_hadamard_satd { havoc_hadamard_satd<Sample> *satd[3]; };
template <typename Sample>
havoc_hadamard_satd<Sample> **havoc_get_hadamard_satd(havoc_table_hadamard_satd<Sample> *table, int log2TrafoSize)
{
    return &table->satd[log2TrafoSize - 1];
}
template <typename Sample> void havoc_populate_hadamard_satd(havoc_table_hadamard_satd<Sample> *table, havoc_code code);

/* ---- havoc/pred_inter.h:34-104 --------------------------------------------------------------------------- */
template <typename Sample>
using HavocPredUni = void(Sample *dst, intptr_t stride_dst, Sample const *ref, intptr_t stride_ref, int nPbW, int nPbH, int xFrac, int yFrac, int bitDepth);
typedef HavocPredUni<uint8_t> havoc_pred_uni_8to8;
typedef HavocPredUni<uint16_t> havoc_pred_uni_16to16;
template <typename Sample> struct HavocTablePredUni { HavocPredUni<Sample> *p[3][2][17][2][2]; };
template <typename Sample>
static HavocPredUni<Sample> **havocGetPredUni(HavocTablePredUni<Sample> *table, int taps, int w, int h, int xFrac, int yFrac, int bitDepth)
{
    return &table->p[bitDepth - 8][taps / 4 - 1][(w + taps - 1) / taps][xFrac ? 1 : 0][yFrac ? 1 : 0];
}
template <typename Sample> void havocPopulatePredUni(HavocTablePredUni<Sample> *table, havoc_code code);

template <typename Sample>
using HavocPredBi = void(Sample *dst0, intptr_t stride_dst, const Sample *ref0, const Sample *ref1, intptr_t stride_ref, int nPbW, int nPbH,
                         int xFrac0, int yFrac0, int xFrac1, int yFrac1, int bitDepth);
template <typename Sample> struct HavocTablePredBi { HavocPredBi<Sample> *p[3][2][9][2]; };
template <typename Sample>
static HavocPredBi<Sample> **havocGetPredBi(HavocTablePredBi<Sample> *table, int taps, int w, int h, int xFracA, int yFracA, int xFracB, int yFracB,
                                            int bitDepth)
{
    const int frac = xFracA || yFracA || xFracB || yFracB;
    return &table->p[bitDepth - 8][taps / 4 - 1][(w + 2 * taps - 1) / (2 * taps)][frac];
}
template <typename Sample> void havocPopulatePredBi(HavocTablePredBi<Sample> *table, havoc_code code);

namespace havoc {

template <typename Sample>
using SubtractBi = void(Sample *dst0, intptr_t stride_dst, const Sample *ref0, intptr_t stride_ref, const Sample *src, intptr_t stride_src, int nPbW,
                        int nPbH, int bitDepth);
template <typename Sample>
struct TableSubtractBi
{
    SubtractBi<Sample> *p;
    SubtractBi<Sample> *&get() { return this->p; }
};
template <typename Sample> void populateSubtractBi(TableSubtractBi<Sample> *table, havoc_code code, int bitDepth = 0);

/* ---- havoc/pred_intra.h:29-60 ---------------------------------------------------------------------------- */
namespace intra {
template <typename Sample> using Function = void(Sample *dst, intptr_t dstStride, Sample const *neighbours, int predModeIntra);
template <typename Sample>
struct Table
{
    Function<Sample> *entries[3 * sizeof(Sample) - 2][4][38];
    inline Function<Sample> *&lookup(int cIdx, int bitDepth, int log2TrafoSize, int predModeIntra)
    {
        if (cIdx == 0 && log2TrafoSize < 5)
        {   /* luma blocks below 32x32 use the edge-filtered variants of DC, horizontal and vertical */
            if (predModeIntra == 1) predModeIntra = 35;
            else if (predModeIntra == 10) predModeIntra = 36;
            else if (predModeIntra == 26) predModeIntra = 37;
        }
        const int bd = sizeof(Sample) == 2 ? 10 - bitDepth : 0;
        return this->entries[bd][log2TrafoSize - 2][predModeIntra];
    }
    void populate(havoc_code code);
};
} // namespace intra

/* ---- havoc/transform.h:31-148 ---------------------------------------------------------------------------- */
using inverse_transform = void(int16_t dst[], int16_t const coeffs[], int bitDepth);
struct table_inverse_transform
{
    inverse_transform *sine;
    inverse_transform *cosine[4];
};
static inline inverse_transform **get_inverse_transform(table_inverse_transform *table, int trType, int log2TrafoSize)
{
    return trType ? &table->sine : &table->cosine[log2TrafoSize - 2];
}
void populate_inverse_transform(table_inverse_transform *table, havoc_code code, int encoder);

template <typename Sample>
using inverse_transform_add = void(Sample *dst, intptr_t stride_dst, Sample const *pred, intptr_t stride_pred, int16_t const coeffs[], int bitDepth);
template <typename Sample>
struct table_inverse_transform_add
{
    inverse_transform_add<Sample> *sine;
    inverse_transform_add<Sample> *cosine[4];
};
template <typename Sample>
static inline inverse_transform_add<Sample> **get_inverse_transform_add(table_inverse_transform_add<Sample> *table, int trType, int log2TrafoSize)
{
    return trType ? &table->sine : &table->cosine[log2TrafoSize - 2];
}
template <typename Sample> void populate_inverse_transform_add(table_inverse_transform_add<Sample> *table, havoc_code code, int encoder);

static inline int clip(int x, int bit_depth)
{
    const int hi = (1 << bit_depth) - 1;
    return x < 0 ? 0 : (x > hi ? hi : x);
}

/* host-side helper kept for source compatibility with havoc/transform.h:104-114 (transform-skip path of the caller) */
template <typename Sample>
void add_residual(int n, Sample *dst, intptr_t stride_dst, Sample const *pred, intptr_t stride_pred, int16_t *residual, int bitDepth)
{
    for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x) dst[x + y * stride_dst] = (Sample)clip(pred[x + y * stride_pred] + residual[x + y * n], bitDepth);
}

typedef void Transform(int16_t *coeffs, const int16_t *src, intptr_t src_stride);
template <int bitDepth>
struct table_transform
{
    Transform *dst;
    Transform *dct[4];
};
template <int bitDepth>
static Transform **get_transform(table_transform<bitDepth> *table, int trType, int log2TrafoSize)
{
    return trType ? &table->dst : &table->dct[log2TrafoSize - 2];
}
template <int bitDepth> void populate_transform(table_transform<bitDepth> *table, havoc_code code);

} // namespace havoc

#endif /* __cplusplus */
#endif /* HAVOC_TABLES_HPP */
