// Device-side picture store + input upload: pictures resident in HBM in the reference's padded
// layout, filled from the on-disk format the reference reads.
//
//   turing/Picture.cpp:91-125      Picture<Sample>(width, height, chromaFormat, paddingX, paddingY, alignment): per plane
//                                  stride = padding + width + padding rounded up to `alignment` bytes, rows = padding +
//                                  height + padding; chroma planes halve width, height and padding (4:2:0)
//   turing/StatePictures.h:155-156 reconstructed pictures: padding 96, alignment 32
//   turing/encode.cpp:377-460      input frames: planar Y, U, V, tightly packed, 8-bit bytes or 16-bit little-endian words;
//                                  8-bit input on the 16-bit path is pre-shifted << 2 (encode.cpp:397)
//   turing/Padding.h:33-57         padImage: border replication of every plane
//
// One HBM allocation per picture (Y | Cb | Cr back to back, each plane 256-byte aligned) so that a job table can address
// any plane of a picture with one base pointer; the 16 fractional-sample luma planes of a reference picture
// (havoc_mi355x_interp_planes) are made on first request and kept with the picture.
#include "ctx.h"

namespace havoc_gpu {
hipError_t launch_pad_block(hipStream_t, int S, void *, long, int, int, long, int, int, int, int, int);
hipError_t launch_interp_planes(hipStream_t, int S, int bd, void *, long, const void *, long, int, int, int, int);

// dst plane (T samples) <- src rows of U samples, value << shift
template <typename T, typename U>
__global__ __launch_bounds__(256) void k_upload_plane(T *__restrict__ dst, long dstStride, const U *__restrict__ src, long srcStride, int w, int h, int shift)
{
    const long n = (long)w * h;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    {
        const int y = (int)(i / w), x = (int)(i - (long)y * w);
        dst[(long)y * dstStride + x] = (T)((unsigned)src[(long)y * srcStride + x] << shift);
    }
}
} // namespace havoc_gpu

using namespace havoc_gpu;

struct havoc_mi355x_picture
{
    int S, bitDepth, width, height, pad, alignment;
    struct Plane
    {
        int width, height, pad;
        long stride;        // samples
        long base;          // first sample of the plane's allocation, in samples from d_base
        long origin;        // sample (0, 0) of the picture proper, in samples from d_base
        long elems;         // samples allocated for the plane
    } plane[3];
    char *d_base;
    size_t bytes;
    char *d_phase;          // 16 luma phase planes (slot 0 = a copy of the luma plane), each plane[0].elems samples
    bool phaseValid;
    char *d_stage;          // staging for host input (tight planar frame)
    size_t stageBytes;
};

#define REQUIRE_S() REQUIRE(S == 1 || S == 2, "S (bytes per sample) must be 1 or 2")

template <typename T, typename U>
static void launch_upload(hipStream_t st, void *dst, long ds, const void *src, long ss, int w, int h, int shift)
{
    const long n = (long)w * h;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL((k_upload_plane<T, U>), dim3(blocks), dim3(256), 0, st, (T *)dst, ds, (const U *)src, ss, w, h, shift);
}


extern "C" {

// ---- pinned, device-visible host memory: results a host loop polls right after a sync without a copy ------------
int havoc_mi355x_host_alloc(havoc_mi355x_ctx *ctx, size_t bytes, void **h_ptr, void **d_ptr)
{
    REQUIRE_CTX();
    REQUIRE(h_ptr != nullptr && d_ptr != nullptr && bytes > 0, "null pointer / zero size");
    int rc = check(hipHostMalloc(h_ptr, bytes, hipHostMallocMapped), "hipHostMalloc");
    if (rc) return rc;
    if ((rc = check(hipHostGetDevicePointer(d_ptr, *h_ptr, 0), "hipHostGetDevicePointer")))
    {
        (void)hipHostFree(*h_ptr);
        *h_ptr = nullptr;
    }
    return rc;
}

int havoc_mi355x_host_free(havoc_mi355x_ctx *ctx, void *h_ptr)
{
    REQUIRE_CTX();
    return check(hipHostFree(h_ptr), "hipHostFree");
}

// asynchronous copies on the context's stream (pair with havoc_mi355x_sync)
int havoc_mi355x_h2d_async(havoc_mi355x_ctx *ctx, void *d_dst, const void *h_src, size_t bytes)
{
    REQUIRE_CTX();
    return check(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, LS(ctx)), "hipMemcpyAsync h2d");
}

int havoc_mi355x_d2h_async(havoc_mi355x_ctx *ctx, void *h_dst, const void *d_src, size_t bytes)
{
    REQUIRE_CTX();
    return check(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, LS(ctx)), "hipMemcpyAsync d2h");
}

// rows of `row_bytes` bytes between pitched buffers, either direction given by `to_device`
int havoc_mi355x_copy_2d(havoc_mi355x_ctx *ctx, void *dst, size_t dst_pitch, const void *src, size_t src_pitch, size_t row_bytes, size_t rows,
                         int to_device)
{
    REQUIRE_CTX();
    if (row_bytes == 0 || rows == 0) return 0;
    return check(hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, row_bytes, rows, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, LS(ctx)),
                 "hipMemcpy2DAsync");
}

// ---- picture store ------------------------------------------------------------------------------------------------

int havoc_mi355x_picture_create(havoc_mi355x_ctx *ctx, int S, int bit_depth, int width, int height, int pad, int alignment,
                                havoc_mi355x_picture **out)
{
    REQUIRE_CTX(); REQUIRE_S();
    REQUIRE(out != nullptr, "null out pointer");
    *out = nullptr;
    REQUIRE(bit_depth >= 8 && bit_depth <= (S == 1 ? 8 : 10), "bit depth must be 8 (S=1) or 8..10 (S=2)");
    REQUIRE(width > 0 && height > 0 && (width & 1) == 0 && (height & 1) == 0, "width / height must be positive and even (4:2:0)");
    REQUIRE(pad >= 0 && (pad & 1) == 0, "pad must be even and >= 0");
    REQUIRE(alignment >= S && alignment <= 256 && (alignment & (alignment - 1)) == 0, "alignment must be a power of two <= 256 bytes");
    havoc_mi355x_picture *p = new havoc_mi355x_picture();
    p->S = S; p->bitDepth = bit_depth; p->width = width; p->height = height; p->pad = pad; p->alignment = alignment;
    const int n = alignment / S;
    long at = 0;
    int w = width, h = height, pd = pad;
    for (int c = 0; c < 3; ++c)
    {
        if (c == 1) { w /= 2; h /= 2; pd /= 2; }               // Picture.cpp:98-104 (4:2:0)
        long stride = pd + w + pd;
        if (stride % n) stride += n - stride % n;               // Picture.cpp:111-116: extra padding on the right
        const long front = (pd % n) ? n - pd % n : 0;           // Picture.cpp:120: first picture sample aligned
        havoc_mi355x_picture::Plane &q = p->plane[c];
        q.width = w; q.height = h; q.pad = pd; q.stride = stride;
        q.base = at;
        q.origin = at + front + (long)pd * stride + pd;
        q.elems = front + stride * (pd + h + pd);
        at += (q.elems + 255) & ~255L;                          // next plane 256-sample aligned
    }
    p->bytes = (size_t)at * S + 256;                            // slack: interpolation kernels read <= 3 samples past a window
    int rc = check(hipMalloc((void **)&p->d_base, p->bytes), "hipMalloc(picture)");
    if (rc) { delete p; return rc; }
    if ((rc = check(hipMemsetAsync(p->d_base, 0, p->bytes, LS(ctx)), "hipMemsetAsync"))) { (void)hipFree(p->d_base); delete p; return rc; }
    *out = p;
    return 0;
}

// ctx must be the context the picture was created with (its device owns the allocations; a NULL ctx cannot free them and is refused
// loudly rather than leaking device memory in silence)
void havoc_mi355x_picture_destroy(havoc_mi355x_ctx *ctx, havoc_mi355x_picture *pic)
{
    if (!pic) return;
    if (!ctx) (void)fail(HAVOC_MI355X_EINVAL, "havoc_mi355x_picture_destroy: NULL context, the picture's device memory is NOT freed");
    if (ctx)
    {
        DeviceGuard g(ctx->device);
        // uploads, pads and phase-plane launches go to the current lane, which may be a forked one: wait for all of them
        (void)havoc_mi355x_sync(ctx);
        if (pic->d_phase) (void)hipFree(pic->d_phase);
        if (pic->d_stage) (void)hipFree(pic->d_stage);
        (void)hipFree(pic->d_base);
    }
    delete pic;
}

// geometry of plane cIdx: *d_base = the picture's allocation (the same for the three planes), *origin_off = sample offset
// of sample (0, 0) from d_base, *stride in samples.  Any pointer may be NULL.
int havoc_mi355x_picture_plane(havoc_mi355x_picture *pic, int cIdx, void **d_base, int64_t *origin_off, intptr_t *stride, int *width, int *height,
                               int *pad)
{
    if (!pic || cIdx < 0 || cIdx > 2) return fail(HAVOC_MI355X_EINVAL, "bad picture / plane index");
    const havoc_mi355x_picture::Plane &q = pic->plane[cIdx];
    if (d_base) *d_base = pic->d_base;
    if (origin_off) *origin_off = q.origin;
    if (stride) *stride = q.stride;
    if (width) *width = q.width;
    if (height) *height = q.height;
    if (pad) *pad = q.pad;
    return 0;
}

static int ensure_stage(havoc_mi355x_picture *pic, size_t bytes)
{
    if (pic->stageBytes >= bytes) return 0;
    if (pic->d_stage) (void)hipFree(pic->d_stage);
    pic->d_stage = nullptr;
    pic->stageBytes = 0;
    const int rc = check(hipMalloc((void **)&pic->d_stage, bytes), "hipMalloc(stage)");
    if (!rc) pic->stageBytes = bytes;
    return rc;
}

// One input frame as the reference reads it (turing/encode.cpp:600-640): planar Y, U, V, tightly packed, `src_S` bytes per
// sample (16-bit = little-endian words).  Samples are stored << shift (encode.cpp:397: 8-bit input on the 16-bit path uses
// shift 2).  The borders are then replicated (Padding::padImage) when pad_after != 0.
int havoc_mi355x_picture_upload_yuv(havoc_mi355x_ctx *ctx, havoc_mi355x_picture *pic, const void *h_yuv, int src_S, int shift, int pad_after)
{
    REQUIRE_CTX();
    REQUIRE(pic != nullptr && h_yuv != nullptr, "null picture / input");
    REQUIRE(src_S == 1 || src_S == 2, "src_S must be 1 or 2");
    REQUIRE(src_S <= pic->S, "16-bit input needs a 16-bit picture");
    REQUIRE(shift >= 0 && shift <= 8, "shift out of range");
    const size_t frame = (size_t)pic->width * pic->height * 3 / 2 * src_S;
    int rc = ensure_stage(pic, frame);
    if (rc) return rc;
    hipStream_t st = LS(ctx);
    if ((rc = check(hipMemcpyAsync(pic->d_stage, h_yuv, frame, hipMemcpyHostToDevice, st), "hipMemcpyAsync(frame)"))) return rc;
    size_t at = 0;
    for (int c = 0; c < 3; ++c)
    {
        const havoc_mi355x_picture::Plane &q = pic->plane[c];
        void *dst = pic->d_base + q.origin * pic->S;
        const void *src = pic->d_stage + at;
        if (pic->S == 1) launch_upload<uint8_t, uint8_t>(st, dst, q.stride, src, q.width, q.width, q.height, shift);
        else if (src_S == 1) launch_upload<uint16_t, uint8_t>(st, dst, q.stride, src, q.width, q.width, q.height, shift);
        else launch_upload<uint16_t, uint16_t>(st, dst, q.stride, src, q.width, q.width, q.height, shift);
        at += (size_t)q.width * q.height * src_S;
    }
    if ((rc = check(hipGetLastError(), "upload_yuv"))) return rc;
    pic->phaseValid = false;
    if (pad_after)
        for (int c = 0; c < 3; ++c)
        {
            const havoc_mi355x_picture::Plane &q = pic->plane[c];
            if ((rc = check(launch_pad_block(st, pic->S, pic->d_base, q.origin, q.width, q.height, q.stride, q.pad, 1, 1, 1, 1), "pad"))) return rc;
        }
    return 0;
}

// One plane from / to a host Picture plane (pointer to its sample (0, 0), stride in samples): the picture proper when
// with_padding == 0, else the plane including its `pad` border (the host plane must have one at least as wide).
int havoc_mi355x_picture_upload_plane(havoc_mi355x_ctx *ctx, havoc_mi355x_picture *pic, int cIdx, const void *h_origin, intptr_t h_stride,
                                      int with_padding)
{
    REQUIRE_CTX();
    REQUIRE(pic != nullptr && h_origin != nullptr && cIdx >= 0 && cIdx <= 2, "bad picture / plane");
    const havoc_mi355x_picture::Plane &q = pic->plane[cIdx];
    const int pd = with_padding ? q.pad : 0;
    const int S = pic->S;
    pic->phaseValid = pic->phaseValid && cIdx != 0;
    return check(hipMemcpy2DAsync(pic->d_base + (q.origin - (long)pd * q.stride - pd) * S, q.stride * S,
                                  (const char *)h_origin - ((long)pd * h_stride + pd) * S, h_stride * S, (size_t)(q.width + 2 * pd) * S,
                                  q.height + 2 * pd, hipMemcpyHostToDevice, LS(ctx)),
                 "hipMemcpy2DAsync(upload plane)");
}

int havoc_mi355x_picture_download_plane(havoc_mi355x_ctx *ctx, havoc_mi355x_picture *pic, int cIdx, void *h_origin, intptr_t h_stride,
                                        int with_padding)
{
    REQUIRE_CTX();
    REQUIRE(pic != nullptr && h_origin != nullptr && cIdx >= 0 && cIdx <= 2, "bad picture / plane");
    const havoc_mi355x_picture::Plane &q = pic->plane[cIdx];
    const int pd = with_padding ? q.pad : 0;
    const int S = pic->S;
    int rc = check(hipMemcpy2DAsync((char *)h_origin - ((long)pd * h_stride + pd) * S, h_stride * S,
                                    pic->d_base + (q.origin - (long)pd * q.stride - pd) * S, q.stride * S, (size_t)(q.width + 2 * pd) * S,
                                    q.height + 2 * pd, hipMemcpyDeviceToHost, LS(ctx)),
                   "hipMemcpy2DAsync(download plane)");
    return rc ? rc : check(hipStreamSynchronize(LS(ctx)), "hipStreamSynchronize");
}

// Padding::padImage on the three planes (turing/Padding.h:33-57)
int havoc_mi355x_picture_pad(havoc_mi355x_ctx *ctx, havoc_mi355x_picture *pic)
{
    REQUIRE_CTX();
    REQUIRE(pic != nullptr, "null picture");
    for (int c = 0; c < 3; ++c)
    {
        const havoc_mi355x_picture::Plane &q = pic->plane[c];
        const int rc = check(launch_pad_block(LS(ctx), pic->S, pic->d_base, q.origin, q.width, q.height, q.stride, q.pad, 1, 1, 1, 1), "pad");
        if (rc) return rc;
    }
    pic->phaseValid = false;
    return 0;
}

// The 16 fractional-sample luma planes of the picture (havoc_mi355x_interp_planes; slot 0 = the luma plane itself),
// computed on first request after the picture changed.  Plane k starts at d_planes + k * plane_elems samples and has the
// luma plane's geometry: sample (x, y) of phase k at index (origin - base) + y * stride + x with the luma plane's
// origin / stride (*luma_first = luma plane's first allocated sample relative to d_base, so that index = origin_off -
// *luma_first + ...).  Covers every position whose 8-tap window stays inside the padded plane.
int havoc_mi355x_picture_phase_planes(havoc_mi355x_ctx *ctx, havoc_mi355x_picture *pic, void **d_planes, intptr_t *plane_elems, int64_t *luma_first)
{
    REQUIRE_CTX();
    REQUIRE(pic != nullptr && d_planes != nullptr && plane_elems != nullptr, "null pointer");
    const havoc_mi355x_picture::Plane &q = pic->plane[0];
    REQUIRE(q.pad >= 16, "phase planes need a padded picture (pad >= 16)");
    const long pe = (q.elems + 63) & ~63L;
    int rc;
    if (!pic->d_phase && (rc = check(hipMalloc((void **)&pic->d_phase, (size_t)pe * 16 * pic->S + 256), "hipMalloc(phase planes)"))) return rc;
    if (!pic->phaseValid)
    {
        hipStream_t st = LS(ctx);
        const char *luma = pic->d_base + q.base * pic->S;
        if ((rc = check(hipMemcpyAsync(pic->d_phase, luma, (size_t)q.elems * pic->S, hipMemcpyDeviceToDevice, st), "copy plane 0"))) return rc;
        // rectangle: the padded plane minus a frame of 12 samples / 4 rows (what the filter taps and the vector loads reach)
        const long first = q.origin - q.base - (long)q.pad * q.stride - q.pad;      // index of the padded plane's top-left sample
        const int fx = (int)(first % q.stride), fy = (int)(first / q.stride);
        const int x0 = fx + 12, y0 = fy + 4, wdt = q.width + 2 * q.pad - 24, hgt = q.height + 2 * q.pad - 8;
        if ((rc = check(launch_interp_planes(st, pic->S, pic->bitDepth, pic->d_phase, pe, luma, q.stride, x0, y0, wdt, hgt), "interp_planes"))) return rc;
        pic->phaseValid = true;
    }
    *d_planes = pic->d_phase;
    *plane_elems = pe;
    if (luma_first) *luma_first = q.base;
    return 0;
}

} // extern "C"
