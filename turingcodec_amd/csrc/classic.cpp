// libhavoc_classic.so -- the reference's per-block table API (include/havoc/havoc_tables.hpp) implemented on top of
// the C ABI of libhavoc_mi355x.so ONLY (no HIP headers here: this file is plain C++ compiled by g++, which is also
// the proof that include/havoc_mi355x.h is sufficient for a host-side integration).
//
// A table entry is a synchronous per-block function with raw HOST pointers and no context argument
// (SURVEY.md 0.2, 8b).  Each call therefore: packs its operands into a per-thread staging buffer, copies it to HBM,
// launches the corresponding batch kernel with ONE job on the thread's private stream, copies the result back and
// returns.  Bit-exact and re-entrant from any number of encoder threads, but bounded by launch latency (tens of
// microseconds per call): it exists so that code written against libhavoc.a links and runs unchanged; the encoder
// integration that wants throughput batches through include/havoc_mi355x.h.
// No CPU implementation is linked: without a gfx950 device havoc_new_code aborts.
#include "../../include/havoc/havoc_tables.hpp"
#include "../../include/havoc_mi355x.h"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

struct Binding
{
    int device;
};

std::atomic<Binding *> g_binding{nullptr};

[[noreturn]] void die(const char *what, int rc)
{
    fprintf(stderr, "libhavoc_classic: %s failed (%d): %s -- there is no CPU fallback\n", what, rc, havoc_mi355x_last_error());
    abort();
}

#define CK(call) do { const int rc_ = (call); if (rc_) die(#call, rc_); } while (0)

// per-thread device context + staging memory (never torn down explicitly: process exit reclaims it, which avoids
// calling into the HIP runtime from thread_local destructors during shutdown)
struct Stage
{
    havoc_mi355x_ctx *ctx = nullptr;
    char *d = nullptr;
    size_t cap = 0;
    std::vector<char> h;
    size_t used = 0;

    void begin()
    {
        if (!ctx)
        {
            Binding *b = g_binding.load();
            if (!b)
            {
                fprintf(stderr, "libhavoc_classic: table function called without a live havoc_code\n");
                abort();
            }
            CK(havoc_mi355x_create(&ctx, b->device, HAVOC_MI355X_NEW_STREAM));
        }
        used = 0;
    }

    size_t reserve(size_t bytes)
    {
        const size_t o = (used + 63) & ~size_t(63);
        used = o + bytes;
        if (h.size() < used + 64) h.resize((used + 64) * 2);
        return o;
    }

    template <typename T>
    size_t pack(const T *p, intptr_t stride, int w, int rows, int pitch)
    {
        const size_t o = reserve(sizeof(T) * size_t(pitch) * rows + 16);
        for (int y = 0; y < rows; ++y) memcpy(&h[o + sizeof(T) * size_t(y) * pitch], p + y * stride, sizeof(T) * w);
        return o;
    }

    void upload()
    {
        if (cap < used + 64)
        {
            if (d) CK(havoc_mi355x_free(ctx, d));
            cap = (used + 64) * 2;
            void *p = nullptr;
            CK(havoc_mi355x_malloc(ctx, &p, cap));
            d = static_cast<char *>(p);
        }
        CK(havoc_mi355x_h2d(ctx, d, h.data(), used));
    }

    void download(size_t off, size_t bytes) { CK(havoc_mi355x_d2h(ctx, &h[off], d + off, bytes)); }

    template <typename T>
    void unpack(T *dst, intptr_t stride, int w, int rows, int pitch, size_t off)
    {
        download(off, sizeof(T) * size_t(pitch) * rows);
        for (int y = 0; y < rows; ++y) memcpy(dst + y * stride, &h[off + sizeof(T) * size_t(y) * pitch], sizeof(T) * w);
    }

    template <typename J> J *job(size_t off) { return reinterpret_cast<J *>(&h[off]); }
    template <typename J> const J *djob(size_t off) const { return reinterpret_cast<const J *>(d + off); }
};

Stage &stage()
{
    static thread_local Stage *s = new Stage();
    s->begin();
    return *s;
}

// ---- distortion metrics ---------------------------------------------------------------------------------------

template <typename Sample>
int sad(const Sample *src, intptr_t ss, const Sample *ref, intptr_t rs, uint32_t rect)
{
    const int w = rect >> 8, h = rect & 0xff;
    Stage &s = stage();
    const size_t j = s.reserve(sizeof(havoc_mi355x_pair_job)), o = s.reserve(4);
    const size_t a = s.pack(src, ss, w, h, w), b = s.pack(ref, rs, w, h, w);
    *s.job<havoc_mi355x_pair_job>(j) = {0, 0, w, h};
    s.upload();
    CK(havoc_mi355x_sad(s.ctx, sizeof(Sample), s.d + a, w, s.d + b, w, s.djob<havoc_mi355x_pair_job>(j), 1, (int32_t *)(s.d + o)));
    s.download(o, 4);
    return *reinterpret_cast<int32_t *>(&s.h[o]);
}

template <typename Sample>
void sad4(const Sample *src, intptr_t ss, const Sample *ref[], intptr_t rs, int out[], uint32_t rect)
{
    const int w = rect >> 8, h = rect & 0xff;
    Stage &s = stage();
    const size_t j = s.reserve(sizeof(havoc_mi355x_sad4_job)), o = s.reserve(16);
    const size_t a = s.pack(src, ss, w, h, w);
    size_t b[4];
    for (int k = 0; k < 4; ++k) b[k] = s.pack(ref[k], rs, w, h, w);
    havoc_mi355x_sad4_job job = {0, {0, 0, 0, 0}, w, h, 0};
    for (int k = 0; k < 4; ++k) job.ref_off[k] = int32_t((b[k] - b[0]) / sizeof(Sample));
    *s.job<havoc_mi355x_sad4_job>(j) = job;
    s.upload();
    CK(havoc_mi355x_sad4(s.ctx, sizeof(Sample), s.d + a, w, s.d + b[0], w, s.djob<havoc_mi355x_sad4_job>(j), 1, (int32_t *)(s.d + o)));
    s.download(o, 16);
    memcpy(out, &s.h[o], 16);
}

template <typename Sample>
uint32_t ssd(const Sample *pa, intptr_t sa, const Sample *pb, intptr_t sb, int w, int h)
{
    Stage &s = stage();
    const size_t j = s.reserve(sizeof(havoc_mi355x_pair_job)), o = s.reserve(4);
    const size_t a = s.pack(pa, sa, w, h, w), b = s.pack(pb, sb, w, h, w);
    *s.job<havoc_mi355x_pair_job>(j) = {0, 0, w, h};
    s.upload();
    CK(havoc_mi355x_ssd(s.ctx, sizeof(Sample), s.d + a, w, s.d + b, w, s.djob<havoc_mi355x_pair_job>(j), 1, (uint32_t *)(s.d + o)));
    s.download(o, 4);
    return *reinterpret_cast<uint32_t *>(&s.h[o]);
}

template <typename Sample, int N>
int satd(const Sample *pa, intptr_t sa, const Sample *pb, intptr_t sb)
{
    Stage &s = stage();
    const size_t j = s.reserve(sizeof(havoc_mi355x_pair_job)), o = s.reserve(4);
    const size_t a = s.pack(pa, sa, N, N, N), b = s.pack(pb, sb, N, N, N);
    *s.job<havoc_mi355x_pair_job>(j) = {0, 0, N, N};
    s.upload();
    CK(havoc_mi355x_satd(s.ctx, sizeof(Sample), N, N, s.d + a, N, s.d + b, N, s.djob<havoc_mi355x_pair_job>(j), 1, (int32_t *)(s.d + o)));
    s.download(o, 4);
    return *reinterpret_cast<int32_t *>(&s.h[o]);
}

int ssdLinear(const uint8_t *a, const uint8_t *b, int size)
{
    Stage &s = stage();
    const size_t o = s.reserve(4);
    const size_t pa = s.pack(a, 0, size, 1, size), pb = s.pack(b, 0, size, 1, size);
    s.upload();
    CK(havoc_mi355x_ssd_linear(s.ctx, (const uint8_t *)(s.d + pa), (const uint8_t *)(s.d + pb), size, (int32_t *)(s.d + o)));
    s.download(o, 4);
    return *reinterpret_cast<int32_t *>(&s.h[o]);
}

// ---- inter prediction -----------------------------------------------------------------------------------------

// packs the (w+taps-1) x (h+taps-1) window around the block (+3 columns the kernel's vector loads may touch; the
// kernel reads the window for every phase, the zero phase included); returns the byte offset and sets *origin to the
// sample offset of the block's integer position
template <typename Sample>
size_t packWindow(Stage &s, const Sample *ref, intptr_t sr, int w, int h, int taps, int *pitch, int *origin)
{
    const int above = taps / 2 - 1, ww = w + taps - 1, wh = h + taps - 1;
    *pitch = ww + 3;
    *origin = above * *pitch + above;
    return s.pack(ref - above * sr - above, sr, ww, wh, *pitch);
}

template <typename Sample, int TAPS>
void predUni(Sample *dst, intptr_t sd, const Sample *ref, intptr_t sr, int w, int h, int xFrac, int yFrac, int bitDepth)
{
    Stage &s = stage();
    const size_t j = s.reserve(sizeof(havoc_mi355x_pred_uni_job));
    int pitch, origin;
    const size_t win = packWindow(s, ref, sr, w, h, TAPS, &pitch, &origin);
    const size_t out = s.reserve(sizeof(Sample) * size_t(w) * h);
    *s.job<havoc_mi355x_pred_uni_job>(j) = {0, origin, w, h, xFrac, yFrac, {0, 0}};
    s.upload();
    CK(havoc_mi355x_pred_uni(s.ctx, sizeof(Sample), TAPS, bitDepth, w, h, s.d + out, w, s.d + win, pitch, s.djob<havoc_mi355x_pred_uni_job>(j), 1));
    s.unpack(dst, sd, w, h, w, out);
}

template <typename Sample, int TAPS>
void predBi(Sample *dst, intptr_t sd, const Sample *ref0, const Sample *ref1, intptr_t sr, int w, int h, int xFrac0, int yFrac0, int xFrac1, int yFrac1,
            int bitDepth)
{
    Stage &s = stage();
    const size_t j = s.reserve(sizeof(havoc_mi355x_pred_bi_job));
    int pitch, origin;
    const size_t w0 = packWindow(s, ref0, sr, w, h, TAPS, &pitch, &origin);
    const size_t w1 = packWindow(s, ref1, sr, w, h, TAPS, &pitch, &origin);
    const size_t out = s.reserve(sizeof(Sample) * size_t(w) * h);
    havoc_mi355x_pred_bi_job job = {0, origin, int32_t((w1 - w0) / sizeof(Sample)) + origin, w, h, xFrac0, yFrac0, xFrac1, yFrac1, {0, 0, 0}};
    *s.job<havoc_mi355x_pred_bi_job>(j) = job;
    s.upload();
    CK(havoc_mi355x_pred_bi(s.ctx, sizeof(Sample), TAPS, bitDepth, w, h, s.d + out, w, s.d + w0, pitch, s.djob<havoc_mi355x_pred_bi_job>(j), 1));
    s.unpack(dst, sd, w, h, w, out);
}

template <typename Sample>
void subtractBi(Sample *dst, intptr_t sd, const Sample *pred, intptr_t sp, const Sample *src, intptr_t ss, int w, int h, int bitDepth)
{
    Stage &s = stage();
    const size_t j = s.reserve(sizeof(havoc_mi355x_subtract_bi_job));
    const size_t p = s.pack(pred, sp, w, h, w), q = s.pack(src, ss, w, h, w);
    const size_t out = s.reserve(sizeof(Sample) * size_t(w) * h);
    *s.job<havoc_mi355x_subtract_bi_job>(j) = {0, 0, 0, w, h, {0, 0, 0}};
    s.upload();
    CK(havoc_mi355x_subtract_bi(s.ctx, sizeof(Sample), bitDepth, s.d + out, w, s.d + p, w, s.d + q, w, s.djob<havoc_mi355x_subtract_bi_job>(j), 1));
    s.unpack(dst, sd, w, h, w, out);
}

// ---- intra prediction -----------------------------------------------------------------------------------------

template <typename Sample, int BITDEPTH, int LOG2, bool EDGE>
void intraPredict(Sample *dst, intptr_t sd, const Sample *neighbours, int mode)
{
    constexpr int n = 1 << LOG2;
    Stage &s = stage();
    const size_t j = s.reserve(sizeof(havoc_mi355x_intra_job));
    const size_t nb = s.pack(neighbours - 2 * n - 1, 0, 4 * n + 1, 1, 4 * n + 1);
    const size_t out = s.reserve(sizeof(Sample) * n * n);
    *s.job<havoc_mi355x_intra_job>(j) = {0, 2 * n + 1, LOG2, mode, EDGE ? 1 : 0, {0, 0, 0}};
    s.upload();
    CK(havoc_mi355x_intra(s.ctx, sizeof(Sample), BITDEPTH, LOG2, s.d + out, n, s.d + nb, s.djob<havoc_mi355x_intra_job>(j), 1));
    s.unpack(dst, sd, n, n, n, out);
}

// ---- transforms and quantisation ------------------------------------------------------------------------------

template <int BITDEPTH, int LOG2, int TR>
void forwardTransform(int16_t *coeffs, const int16_t *src, intptr_t stride)
{
    constexpr int n = 1 << LOG2;
    Stage &s = stage();
    const size_t j = s.reserve(sizeof(havoc_mi355x_tu_job));
    const size_t r = s.pack(src, stride, n, n, n);
    const size_t out = s.reserve(2 * n * n);
    *s.job<havoc_mi355x_tu_job>(j) = {0, 0, 0, 0};
    s.upload();
    CK(havoc_mi355x_transform(s.ctx, BITDEPTH, TR, LOG2, (int16_t *)(s.d + out), (const int16_t *)(s.d + r), n, s.djob<havoc_mi355x_tu_job>(j), 1));
    s.download(out, 2 * n * n);
    memcpy(coeffs, &s.h[out], 2 * n * n);
}

template <int LOG2, int TR>
void inverseTransform(int16_t dst[], int16_t const coeffs[], int bitDepth)
{
    constexpr int n = 1 << LOG2;
    Stage &s = stage();
    const size_t j = s.reserve(sizeof(havoc_mi355x_tu_job));
    const size_t c = s.pack(coeffs, 0, n * n, 1, n * n);
    const size_t out = s.reserve(2 * n * n);
    *s.job<havoc_mi355x_tu_job>(j) = {0, 0, 0, 0};
    s.upload();
    CK(havoc_mi355x_inverse_transform(s.ctx, bitDepth, TR, LOG2, (int16_t *)(s.d + out), (const int16_t *)(s.d + c), s.djob<havoc_mi355x_tu_job>(j), 1));
    s.download(out, 2 * n * n);
    memcpy(dst, &s.h[out], 2 * n * n);
}

template <typename Sample, int LOG2, int TR>
void inverseTransformAdd(Sample *dst, intptr_t sd, Sample const *pred, intptr_t sp, int16_t const coeffs[], int bitDepth)
{
    constexpr int n = 1 << LOG2;
    Stage &s = stage();
    const size_t j = s.reserve(sizeof(havoc_mi355x_tu_job));
    const size_t c = s.pack(coeffs, 0, n * n, 1, n * n);
    const size_t p = s.pack(pred, sp, n, n, n);       // staged before dst is written: pred may alias dst
    const size_t out = s.reserve(sizeof(Sample) * n * n);
    *s.job<havoc_mi355x_tu_job>(j) = {0, 0, 0, 0};
    s.upload();
    CK(havoc_mi355x_inverse_transform_add(s.ctx, sizeof(Sample), bitDepth, TR, LOG2, s.d + out, n, s.d + p, n, (const int16_t *)(s.d + c),
                                          s.djob<havoc_mi355x_tu_job>(j), 1));
    s.unpack(dst, sd, n, n, n, out);
}

void quantizeInverse(int16_t *dst, const int16_t *src, int scale, int shift, int n)
{
    Stage &s = stage();
    const size_t j = s.reserve(sizeof(havoc_mi355x_quant_job));
    const size_t in = s.pack(src, 0, n, 1, n);
    const size_t out = s.reserve(2 * n);
    *s.job<havoc_mi355x_quant_job>(j) = {0, 0, n, scale, shift, 0, {0, 0}};
    s.upload();
    CK(havoc_mi355x_quantize_inverse(s.ctx, (int16_t *)(s.d + out), (const int16_t *)(s.d + in), s.djob<havoc_mi355x_quant_job>(j), 1));
    s.download(out, 2 * n);
    memcpy(dst, &s.h[out], 2 * n);
}

int quantize(int16_t *dst, const int16_t *src, int scale, int shift, int offset, int n)
{
    Stage &s = stage();
    const size_t j = s.reserve(sizeof(havoc_mi355x_quant_job)), cbf = s.reserve(4);
    const size_t in = s.pack(src, 0, n, 1, n);
    const size_t out = s.reserve(2 * n);
    *s.job<havoc_mi355x_quant_job>(j) = {0, 0, n, scale, shift, offset, {0, 0}};
    s.upload();
    CK(havoc_mi355x_quantize(s.ctx, (int16_t *)(s.d + out), (const int16_t *)(s.d + in), s.djob<havoc_mi355x_quant_job>(j), 1, (int32_t *)(s.d + cbf)));
    s.download(out, 2 * n);
    memcpy(dst, &s.h[out], 2 * n);
    s.download(cbf, 4);
    return *reinterpret_cast<int32_t *>(&s.h[cbf]);
}

template <int LOG2>
void quantizeReconstruct(uint8_t *rec, intptr_t sr, const uint8_t *pred, intptr_t sp, const int16_t *res, int n)
{
    Stage &s = stage();
    const size_t j = s.reserve(sizeof(havoc_mi355x_tu_job));
    const size_t p = s.pack(pred, sp, n, n, n), r = s.pack(res, 0, n * n, 1, n * n);
    const size_t out = s.reserve(size_t(n) * n);
    *s.job<havoc_mi355x_tu_job>(j) = {0, 0, 0, 0};
    s.upload();
    CK(havoc_mi355x_quantize_reconstruct(s.ctx, LOG2, (uint8_t *)(s.d + out), n, (const uint8_t *)(s.d + p), n, (const int16_t *)(s.d + r),
                                         s.djob<havoc_mi355x_tu_job>(j), 1));
    s.unpack(rec, sr, n, n, n, out);
}

template <typename Sample, int BD, int LOG2>
void fillIntra(havoc::intra::Function<Sample> *(&row)[38])
{
    for (int m = 0; m < 35; ++m) row[m] = intraPredict<Sample, BD, LOG2, false>;
    for (int m = 35; m < 38; ++m) row[m] = intraPredict<Sample, BD, LOG2, true>;
}

template <typename Sample, int BD>
void fillIntraDepth(havoc::intra::Function<Sample> *(&t)[4][38])
{
    fillIntra<Sample, BD, 2>(t[0]);
    fillIntra<Sample, BD, 3>(t[1]);
    fillIntra<Sample, BD, 4>(t[2]);
    fillIntra<Sample, BD, 5>(t[3]);
}

} // namespace

// ---- library core (havoc/havoc.h:132-153) -------------------------------------------------------------------------

extern "C" {

havoc_instruction_set havoc_instruction_set_support(void)
{
    return (havoc_instruction_set)(HAVOC_C_REF | HAVOC_C_OPT | HAVOC_GFX950);   // C bits kept so mask tests in callers pass
}

void havoc_print_instruction_set_support(FILE *f, havoc_instruction_set mask)
{
    fprintf(f ? f : stdout, "havoc (MI355X build): every table entry runs on gfx950 [%c]; x86 mask bits are accepted and ignored (mask 0x%x)\n",
            (mask & HAVOC_GFX950) ? 'x' : ' ', (unsigned)mask);
}

havoc_code havoc_new_code(havoc_instruction_set mask, int size)
{
    (void)mask;
    (void)size;
    Binding *b = new Binding{0};
    if (const char *e = getenv("HAVOC_MI355X_DEVICE")) b->device = atoi(e);
    havoc_mi355x_ctx *probe = nullptr;
    const int rc = havoc_mi355x_create(&probe, b->device, nullptr);
    if (rc) die("havoc_new_code: havoc_mi355x_create", rc);
    havoc_mi355x_destroy(probe);
    g_binding.store(b);
    havoc_code code;
    code.implementation = b;
    return code;
}

void havoc_delete_code(havoc_code code)
{
    Binding *b = static_cast<Binding *>(code.implementation);
    Binding *cur = b;
    g_binding.compare_exchange_strong(cur, nullptr);
    delete b;
}

void havoc_populate_quantize_inverse(havoc_table_quantize_inverse *table, havoc_code)
{
    table->p[0] = table->p[1] = quantizeInverse;
}

void havoc_populate_quantize(havoc_table_quantize *table, havoc_code) { table->p = quantize; }

void havoc_populate_quantize_reconstruct(havoc_table_quantize_reconstruct *table, havoc_code)
{
    table->p[0] = quantizeReconstruct<2>;
    table->p[1] = quantizeReconstruct<3>;
    table->p[2] = quantizeReconstruct<4>;
    table->p[3] = quantizeReconstruct<5>;
}

havoc_ssd_linear *havoc_get_ssd_linear(int, havoc_code) { return ssdLinear; }

int havoc_main(int, const char *[])
{
    // self-check: populate everything and make one call per family; parity proper lives in tests/
    havoc_code code = havoc_new_code(havoc_instruction_set_support(), 0);
    havoc_table_sad<uint8_t> ts;
    havoc_populate_sad(&ts, code);
    uint8_t a[64 * 64], b[64 * 64];
    for (int i = 0; i < 64 * 64; ++i) { a[i] = uint8_t(i * 7); b[i] = uint8_t(i * 13); }
    int expect = 0;
    for (int y = 0; y < 16; ++y)
        for (int x = 0; x < 16; ++x) expect += abs(int(a[y * 64 + x]) - int(b[y * 64 + x]));
    const int got = (*havoc_get_sad(&ts, 16, 16))(a, 64, b, 64, HAVOC_RECT(16, 16));
    printf("havoc (MI355X) self check: sad16x16 %d (expected %d)\n", got, expect);
    havoc_delete_code(code);
    return got == expect ? 0 : 1;
}

} // extern "C"

// ---- table population (C++ linkage, same names as the reference's explicit instantiations) ----------------------

template <typename Sample>
void havoc_populate_sad(havoc_table_sad<Sample> *table, havoc_code)
{
    for (int h = 4; h <= 64; h += 4)
        for (int w = 4; w <= 64; w += 4) *havoc_get_sad(table, w, h) = sad<Sample>;   // havoc/sad.cpp:494-504
}
template void havoc_populate_sad<uint8_t>(havoc_table_sad<uint8_t> *, havoc_code);
template void havoc_populate_sad<uint16_t>(havoc_table_sad<uint16_t> *, havoc_code);

template <typename Sample>
void havoc_populate_sad_multiref(havoc_table_sad_multiref<Sample> *table, havoc_code)
{
    for (auto &row : table->lookup)
        for (auto &e : row) e = sad4<Sample>;
    table->sadGeneric_4 = sad4<Sample>;
}
template void havoc_populate_sad_multiref<uint8_t>(havoc_table_sad_multiref<uint8_t> *, havoc_code);
template void havoc_populate_sad_multiref<uint16_t>(havoc_table_sad_multiref<uint16_t> *, havoc_code);

template <typename Sample>
void havoc_populate_ssd(havoc_table_ssd<Sample> *table, havoc_code)
{
    for (auto &e : table->ssd) e = ssd<Sample>;
}
template void havoc_populate_ssd<uint8_t>(havoc_table_ssd<uint8_t> *, havoc_code);
template void havoc_populate_ssd<uint16_t>(havoc_table_ssd<uint16_t> *, havoc_code);

template <typename Sample>
void havoc_populate_hadamard_satd(havoc_table_hadamard_satd<Sample> *table, havoc_code)
{
    table->satd[0] = satd<Sample, 2>;
    table->satd[1] = satd<Sample, 4>;
    table->satd[2] = satd<Sample, 8>;
}
template void havoc_populate_hadamard_satd<uint8_t>(havoc_table_hadamard_satd<uint8_t> *, havoc_code);
template void havoc_populate_hadamard_satd<uint16_t>(havoc_table_hadamard_satd<uint16_t> *, havoc_code);

template <typename Sample>
void havocPopulatePredUni(HavocTablePredUni<Sample> *table, havoc_code)
{
    for (auto &bd : table->p)
        for (int t = 0; t < 2; ++t)
            for (auto &wc : bd[t])
                for (auto &xf : wc)
                    for (auto &e : xf) e = t ? predUni<Sample, 8> : predUni<Sample, 4>;
}
template void havocPopulatePredUni<uint8_t>(HavocTablePredUni<uint8_t> *, havoc_code);
template void havocPopulatePredUni<uint16_t>(HavocTablePredUni<uint16_t> *, havoc_code);

template <typename Sample>
void havocPopulatePredBi(HavocTablePredBi<Sample> *table, havoc_code)
{
    for (auto &bd : table->p)
        for (int t = 0; t < 2; ++t)
            for (auto &wc : bd[t])
                for (auto &e : wc) e = t ? predBi<Sample, 8> : predBi<Sample, 4>;
}
template void havocPopulatePredBi<uint8_t>(HavocTablePredBi<uint8_t> *, havoc_code);
template void havocPopulatePredBi<uint16_t>(HavocTablePredBi<uint16_t> *, havoc_code);

namespace havoc {

template <typename Sample>
void populateSubtractBi(TableSubtractBi<Sample> *table, havoc_code, int)
{
    table->get() = subtractBi<Sample>;
}
template void populateSubtractBi<uint8_t>(TableSubtractBi<uint8_t> *, havoc_code, int);
template void populateSubtractBi<uint16_t>(TableSubtractBi<uint16_t> *, havoc_code, int);

namespace intra {
template <> void Table<uint8_t>::populate(havoc_code) { fillIntraDepth<uint8_t, 8>(this->entries[0]); }
template <> void Table<uint16_t>::populate(havoc_code)
{   // entries[10 - bitDepth]: [0] = 10-bit, [1] = 9-bit, [2] = 8-bit (havoc/pred_intra.h:49-50)
    fillIntraDepth<uint16_t, 10>(this->entries[0]);
    fillIntraDepth<uint16_t, 9>(this->entries[1]);
    fillIntraDepth<uint16_t, 8>(this->entries[2]);
}
} // namespace intra

void populate_inverse_transform(table_inverse_transform *table, havoc_code, int)
{
    table->sine = inverseTransform<2, 1>;
    table->cosine[0] = inverseTransform<2, 0>;
    table->cosine[1] = inverseTransform<3, 0>;
    table->cosine[2] = inverseTransform<4, 0>;
    table->cosine[3] = inverseTransform<5, 0>;
}

template <typename Sample>
void populate_inverse_transform_add(table_inverse_transform_add<Sample> *table, havoc_code, int)
{
    table->sine = inverseTransformAdd<Sample, 2, 1>;
    table->cosine[0] = inverseTransformAdd<Sample, 2, 0>;
    table->cosine[1] = inverseTransformAdd<Sample, 3, 0>;
    table->cosine[2] = inverseTransformAdd<Sample, 4, 0>;
    table->cosine[3] = inverseTransformAdd<Sample, 5, 0>;
}
template void populate_inverse_transform_add<uint8_t>(table_inverse_transform_add<uint8_t> *, havoc_code, int);
template void populate_inverse_transform_add<uint16_t>(table_inverse_transform_add<uint16_t> *, havoc_code, int);

template <int bitDepth>
void populate_transform(table_transform<bitDepth> *table, havoc_code)
{
    table->dst = forwardTransform<bitDepth, 2, 1>;
    table->dct[0] = forwardTransform<bitDepth, 2, 0>;
    table->dct[1] = forwardTransform<bitDepth, 3, 0>;
    table->dct[2] = forwardTransform<bitDepth, 4, 0>;
    table->dct[3] = forwardTransform<bitDepth, 5, 0>;
}
template void populate_transform<8>(table_transform<8> *, havoc_code);
template void populate_transform<10>(table_transform<10> *, havoc_code);

} // namespace havoc
