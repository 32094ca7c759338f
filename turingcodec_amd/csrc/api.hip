// C ABI of libhavoc_mi355x.so (include/havoc_mi355x.h): context management, argument validation, launches.
// There is no CPU path in this library: without a gfx950 device havoc_mi355x_create fails with ENODEV.
#include "common.h"

#include <cstdio>
#include <cstring>

namespace havoc_gpu {
hipError_t launch_sad(hipStream_t, int S, int ways, const void *, long, const void *, long, const void *, int, int32_t *);
hipError_t launch_sad_surface(hipStream_t, int S, int range, int maxw, int maxh, const void *, long, const void *, long, const void *, int, int32_t *);
hipError_t launch_ssd(hipStream_t, int S, const void *, long, const void *, long, const void *, int, uint32_t *);
hipError_t launch_satd(hipStream_t, int S, int maxw, int maxh, const void *, long, const void *, long, const void *, int, int32_t *);
hipError_t launch_satd_multi(hipStream_t, int S, int maxw, int maxh, const void *, long, const void *, long, const void *, int, int32_t *);
hipError_t launch_pad_block(hipStream_t, int S, void *, long, int, int, long, int, int, int, int, int);
hipError_t launch_ssd_linear(hipStream_t, const uint8_t *, const uint8_t *, int, int32_t *);
hipError_t launch_derive_bs(hipStream_t, const void *, long, int, int, int8_t *, uint8_t *);
hipError_t launch_deblock(hipStream_t, int S, int bd, void *, long, void *, void *, long, int, int, const int8_t *, const uint8_t *, int, int, int, int);
hipError_t launch_pred_uni(hipStream_t, int S, int taps, int bd, int maxw, int maxh, void *, long, const void *, long, const void *, int);
hipError_t launch_pred_bi(hipStream_t, int S, int taps, int bd, int maxw, int maxh, void *, long, const void *, long, const void *, int);
hipError_t launch_subtract_bi(hipStream_t, int S, int bd, void *, long, const void *, long, const void *, long, const void *, int);
hipError_t launch_pred_classes(hipStream_t, int bi, int S, int taps, int bd, void *, long, const void *, long, const void *, const int count[4]);
hipError_t launch_intra(hipStream_t, int S, int log2, int bd, void *, long, const void *, const void *, int);
hipError_t launch_intra_satd35(hipStream_t, int S, int log2, int bd, const void *, long, const void *, const void *, int, int32_t *);
hipError_t launch_sad4_runs(hipStream_t, int S, const void *, long, const void *, long, const void *, int, const void *, int, int32_t *);
hipError_t launch_interp_planes(hipStream_t, int S, int bd, void *, long, const void *, long, int, int, int, int);
hipError_t launch_subpel_satd(hipStream_t, int S, int taps, int bd, int maxw, int maxh, const void *, long, const void *, long, const void *, int,
                              int32_t *);
hipError_t launch_transform(hipStream_t, int bd, int log2, int tr, int16_t *, const int16_t *, long, const void *, int);
hipError_t launch_inverse_transform(hipStream_t, int mode, int bd, int log2, int tr, void *, long, const void *, long, int16_t *, const int16_t *,
                                    const void *, int);
hipError_t launch_tu_forward(hipStream_t, int S, int bd, int log2, int tr, int16_t *, const void *, long, const void *, long, const void *, int);
hipError_t launch_intra_measure(hipStream_t, int S, int bd, int log2, int16_t *, int16_t *, int32_t *, void *, uint32_t *, const void *, long, const void *, long, const void *, int, int);
hipError_t launch_tu_forward_scan(hipStream_t, int S, int bd, int log2, int16_t *, const void *, long, const void *, long, const void *, int, const void *, int16_t *,
                                  void *);
hipError_t launch_rdoq_prescanned(hipStream_t, int bitDepth, int log2, int16_t *, const int16_t *, const uint8_t *, const void *, int, int32_t *, void *);
hipError_t launch_tu_reconstruct(hipStream_t, int S, int bd, int log2, int tr, int scale, int shift, void *, long, const void *, long, const void *,
                                 long, const int16_t *, const void *, int, uint32_t *);
hipError_t launch_quantize(hipStream_t, int16_t *, const int16_t *, const void *, int, int32_t *);
hipError_t launch_level_stats(hipStream_t, const int16_t *, const void *, int, int32_t *);
hipError_t launch_intra_order(hipStream_t, const int32_t *, const void *, int, int32_t, int32_t *, int32_t *, int32_t *, int32_t *);
hipError_t launch_intra_expand(hipStream_t, const void *, const int32_t *, const int32_t *, const int32_t *, const int32_t *, int, int, int, int, int, int, int, int, void *, void *,
                               void *, int32_t *, int32_t *);
hipError_t launch_intra_decide(hipStream_t, const void *, const int32_t *, const int32_t *, const int32_t *, const int32_t *, const uint32_t *, const int32_t *, const void *, int,
                               int, int32_t, void *, void *);
hipError_t launch_merge_jobs(hipStream_t, const void *, const int16_t *, const int32_t *, const int32_t *, int, int, void *, void *, void *, int16_t *);
hipError_t launch_pred_jobs(hipStream_t, const void *, const int16_t *, int, const int32_t *, const int32_t *, int, int, int, const int32_t *, void *);
hipError_t launch_rqt_decide(hipStream_t, const void *, int, const int32_t *, const int32_t *, const void *, long, int, int, int32_t, void *);
hipError_t launch_block_cells(hipStream_t, int, int, int, int, const int16_t *, const void *, const void *, int, void *, bool);
hipError_t launch_search_wait_rows(hipStream_t, const void *, int, int, int, int *);
hipError_t launch_intra_gather(hipStream_t, int, const void *, const void *, const int32_t *, const uint8_t *, const void *, int, const void *, void *, void *);
hipError_t launch_intra_commit(hipStream_t, int, const void *, void *, uint8_t *, const void *, int, const void *, const void *, int);
hipError_t launch_intra_fill_spare(hipStream_t, const int32_t *, int, int, void *, void *, void *, int32_t *, int32_t *);
hipError_t launch_merge_decide(hipStream_t, const int32_t *, const int32_t *, const int32_t *, int, int64_t, int64_t *, int32_t *);
size_t search_workspace_bytes(int width, int height);
hipError_t launch_search_list(hipStream_t, int S, const havoc_mi355x_search_params *, const void *, long, long, const void *, long, long, const void *, long, long, const void *,
                              int, void *);
hipError_t launch_search_bi_list(hipStream_t, int S, const havoc_mi355x_search_params *, const void *, long, long, const void *, long, long, const void *, long, long, const void *,
                                 long, const void *, const int16_t *, int, void *);
hipError_t launch_search_picture_uni(hipStream_t, int S, const havoc_mi355x_search_params *, const int64_t *, const void *, long, long, const void *, const long *, long,
                                     const void *, long, const long *, const void *, const int32_t *, int, int, int, void *, void *, int16_t *, void *, int, const int32_t *);
hipError_t launch_rdoq(hipStream_t, int bd, int log2, int16_t *, const int16_t *, const uint8_t *, const void *, int, int32_t *, void *);
size_t rdoq_workspace_bytes(int njobs);
hipError_t launch_sao_stats(hipStream_t, int S, int bd, const void *, long, const void *, long, const void *, int, int64_t *);
hipError_t launch_sao_band_chroma(hipStream_t, int S, int bd, const void *, long, const void *, long, const void *, int, int64_t *);
hipError_t launch_sao_filter(hipStream_t, int S, int bd, void *, long, const void *, long, const void *, int);
hipError_t launch_quantize_inverse(hipStream_t, int16_t *, const int16_t *, const void *, int);
hipError_t launch_quantize_reconstruct(hipStream_t, int log2, uint8_t *, long, const uint8_t *, long, const int16_t *, const void *, int);
hipError_t launch_residual(hipStream_t, int S, int16_t *, long, const int32_t *, const void *, long, const void *, long, const void *, int);
} // namespace havoc_gpu

using namespace havoc_gpu;

static_assert(sizeof(havoc_mi355x_pair_job) == 16, "job ABI");
static_assert(sizeof(havoc_mi355x_sad4_job) == 32, "job ABI");
static_assert(sizeof(havoc_mi355x_surface_job) == 32, "job ABI");
static_assert(sizeof(havoc_mi355x_satd_multi_job) == 80, "job ABI");
static_assert(sizeof(havoc_mi355x_pred_uni_job) == 32, "job ABI");
static_assert(sizeof(havoc_mi355x_pred_bi_job) == 48, "job ABI");
static_assert(sizeof(havoc_mi355x_subtract_bi_job) == 32, "job ABI");
static_assert(sizeof(havoc_mi355x_intra_job) == 32, "job ABI");
static_assert(sizeof(havoc_mi355x_tu_job) == 16, "job ABI");
static_assert(sizeof(havoc_mi355x_intra_search_job) == 32, "job ABI");
static_assert(sizeof(havoc_mi355x_tu_fused_job) == 16, "job ABI");
static_assert(sizeof(havoc_mi355x_quant_job) == 32, "job ABI");
static_assert(sizeof(havoc_mi355x_intra_mpm) == 40 && sizeof(havoc_mi355x_intra_choice) == 40, "job ABI");

#include "ctx.h"

static thread_local char g_err_storage[256] = "";
char *havoc_err_buf() { return g_err_storage; }

#define REQUIRE_S() REQUIRE(S == 1 || S == 2, "S (bytes per sample) must be 1 or 2")
#define REQUIRE_BD() REQUIRE(bitDepth >= 8 && bitDepth <= (S == 1 ? 8 : 10), "bitDepth must be 8 (S=1) or 8..10 (S=2)")

extern "C" {

const char *havoc_mi355x_last_error(void) { return g_err; }
const char *havoc_mi355x_version(void) { return "havoc_mi355x 0.1 (gfx950)"; }

int havoc_mi355x_create(havoc_mi355x_ctx **out, int device, void *stream)
{
    REQUIRE(out != nullptr, "null out pointer");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) return fail(HAVOC_MI355X_ENODEV, "no HIP device: libhavoc_mi355x has no CPU path");
    REQUIRE(device >= 0 && device < count, "device index out of range");
    DeviceGuard device_guard_(device);   // the caller's current device is put back on return
    havoc_mi355x_ctx *c = new havoc_mi355x_ctx();
    c->device = device;
    c->ownsStream = stream == HAVOC_MI355X_NEW_STREAM;
    c->stream = c->ownsStream ? nullptr : (hipStream_t)stream;
    auto bail = [&](int rc) {   // undo whatever was created so far
        if (c->ev0) (void)hipEventDestroy(c->ev0);
        if (c->ev1) (void)hipEventDestroy(c->ev1);
        if (c->ownsStream && c->stream) (void)hipStreamDestroy(c->stream);
        delete c;
        return rc;
    };
    int rc;
    if ((rc = check(hipGetDeviceProperties(&c->prop, device), "hipGetDeviceProperties"))) return bail(rc);
    if (strncmp(c->prop.gcnArchName, "gfx950", 6) != 0)
    {
        snprintf(g_err, 256, "device %d is %s; this library contains gfx950 code only", device, c->prop.gcnArchName);
        return bail(HAVOC_MI355X_ENODEV);
    }
    if ((rc = check(hipEventCreate(&c->ev0), "hipEventCreate")) || (rc = check(hipEventCreate(&c->ev1), "hipEventCreate"))) return bail(rc);
    if (c->ownsStream && (rc = check(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking), "hipStreamCreate"))) return bail(rc);
    *out = c;
    return 0;
}

void havoc_mi355x_destroy(havoc_mi355x_ctx *ctx)
{
    if (!ctx) return;
    DeviceGuard device_guard_(ctx->device);
    for (int k = 1; k < havoc_mi355x_ctx::kMaxLanes; ++k)
        if (ctx->lanes[k])
        {
            (void)hipStreamDestroy(ctx->lanes[k]);
            (void)hipEventDestroy(ctx->laneEv[k]);
        }
    if (ctx->forkEv) (void)hipEventDestroy(ctx->forkEv);
    if (ctx->flagH) (void)hipHostFree(const_cast<uint32_t *>(ctx->flagH));
    if (ctx->ownsStream) (void)hipStreamDestroy(ctx->stream);
    (void)hipEventDestroy(ctx->ev0);
    (void)hipEventDestroy(ctx->ev1);
    delete ctx;
}

int havoc_mi355x_set_stream(havoc_mi355x_ctx *ctx, void *stream)
{
    REQUIRE_CTX();
    REQUIRE(!ctx->ownsStream, "context owns its stream");
    ctx->stream = (hipStream_t)stream;
    return 0;
}

int havoc_mi355x_sync(havoc_mi355x_ctx *ctx)
{
    REQUIRE_CTX();
    for (int k = 1; k < ctx->nlanes; ++k)   // lanes forked and not yet joined
    {
        const int rc = check(hipStreamSynchronize(ctx->lanes[k]), "hipStreamSynchronize(lane)");
        if (rc) return rc;
    }
    return check(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
}

// The wait of a caller that launched a few microseconds of work and needs the result NOW (libhavoc_classic.so's table calls): the stream writes a sequence number
// into pinned memory behind the work queued so far and the calling thread polls that word -- hipStreamSynchronize costs ~10 us of its own on this stack, a polled
// word is seen ~1-2 us after the kernel ends.  Bounded: after ~2 ms of polling (or if the write cannot be queued) it falls back to hipStreamSynchronize, which is
// also what reports a device fault.
int havoc_mi355x_sync_spin(havoc_mi355x_ctx *ctx)
{
    REQUIRE_CTX();
    if (ctx->nlanes > 1) return havoc_mi355x_sync(ctx);
    if (!ctx->flagH)
    {
        void *h = nullptr, *d = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&d, h, 0) != hipSuccess)
        {
            (void)hipGetLastError();
            return check(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
        }
        memset(h, 0, 64);
        ctx->flagH = static_cast<volatile uint32_t *>(h);
        ctx->flagD = d;
    }
    const uint32_t want = ++ctx->flagSeq;
    if (hipStreamWriteValue32(ctx->stream, ctx->flagD, want, 0) != hipSuccess)
    {
        (void)hipGetLastError();
        return check(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    }
    for (int spins = 0; spins < (1 << 18); ++spins)
    {
        if (*ctx->flagH == want)
        {
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            return 0;
        }
        __builtin_ia32_pause();
    }
    return check(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
}

int havoc_mi355x_device_info(havoc_mi355x_ctx *ctx, int64_t info[8])
{
    REQUIRE_CTX();
    const hipDeviceProp_t &p = ctx->prop;
    info[0] = p.multiProcessorCount;
    info[1] = p.clockRate;
    info[2] = p.memoryClockRate;
    info[3] = p.memoryBusWidth;
    info[4] = p.l2CacheSize;
    info[5] = p.warpSize;
    info[6] = (int64_t)p.sharedMemPerBlock;
    info[7] = (int64_t)(p.totalGlobalMem >> 20);
    return 0;
}

int havoc_mi355x_malloc(havoc_mi355x_ctx *ctx, void **d_ptr, size_t bytes)
{
    REQUIRE_CTX();
    return check(hipMalloc(d_ptr, bytes), "hipMalloc");
}

int havoc_mi355x_free(havoc_mi355x_ctx *ctx, void *d_ptr)
{
    REQUIRE_CTX();
    return check(hipFree(d_ptr), "hipFree");
}

int havoc_mi355x_h2d(havoc_mi355x_ctx *ctx, void *d_dst, const void *h_src, size_t bytes)
{
    REQUIRE_CTX();
    int rc = check(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync h2d");
    return rc ? rc : check(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
}

int havoc_mi355x_d2h(havoc_mi355x_ctx *ctx, void *h_dst, const void *d_src, size_t bytes)
{
    REQUIRE_CTX();
    int rc = check(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync d2h");
    return rc ? rc : check(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
}

int havoc_mi355x_timer_start(havoc_mi355x_ctx *ctx)
{
    REQUIRE_CTX();
    return check(hipEventRecord(ctx->ev0, ctx->stream), "hipEventRecord");
}

int havoc_mi355x_timer_stop_ms(havoc_mi355x_ctx *ctx, float *ms)
{
    REQUIRE_CTX();
    int rc = check(hipEventRecord(ctx->ev1, ctx->stream), "hipEventRecord");
    if (rc) return rc;
    if ((rc = check(hipEventSynchronize(ctx->ev1), "hipEventSynchronize"))) return rc;
    return check(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1), "hipEventElapsedTime");
}

// ---- HIP graphs ----------------------------------------------------------------------------------------------

struct havoc_mi355x_graph
{
    hipGraph_t graph;
    hipGraphExec_t exec;
};

// side streams + their join events, created on first need and kept for the context's life
static int ensure_lanes(havoc_mi355x_ctx *ctx, int nlanes)
{
    int rc;
    if (!ctx->forkEv && (rc = check(hipEventCreateWithFlags(&ctx->forkEv, hipEventDisableTiming), "hipEventCreate"))) return rc;
    for (int k = 1; k < nlanes; ++k)
        if (!ctx->lanes[k])
        {
            if ((rc = check(hipStreamCreateWithFlags(&ctx->lanes[k], hipStreamNonBlocking), "hipStreamCreate"))) return rc;
            if ((rc = check(hipEventCreateWithFlags(&ctx->laneEv[k], hipEventDisableTiming), "hipEventCreate"))) return rc;
        }
    return 0;
}

int havoc_mi355x_graph_begin(havoc_mi355x_ctx *ctx)
{
    REQUIRE_CTX();
    REQUIRE(ctx->stream != nullptr, "graph capture needs a non-default stream (create the context with HAVOC_MI355X_NEW_STREAM)");
    int rc = ensure_lanes(ctx, havoc_mi355x_ctx::kMaxLanes);   // a fork() inside the capture must not create streams / events
    if (rc) return rc;
    return check(hipStreamBeginCapture(LS(ctx),hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture");
}

int havoc_mi355x_graph_end(havoc_mi355x_ctx *ctx, havoc_mi355x_graph **graph)
{
    REQUIRE_CTX();
    REQUIRE(graph != nullptr, "null graph pointer");
    *graph = nullptr;
    hipGraph_t g = nullptr;
    int rc = check(hipStreamEndCapture(LS(ctx),&g), "hipStreamEndCapture");
    if (rc) return rc;
    hipGraphExec_t e = nullptr;
    if ((rc = check(hipGraphInstantiate(&e, g, nullptr, nullptr, 0), "hipGraphInstantiate")))
    {
        (void)hipGraphDestroy(g);
        return rc;
    }
    *graph = new havoc_mi355x_graph{g, e};
    return 0;
}

int havoc_mi355x_graph_launch(havoc_mi355x_ctx *ctx, havoc_mi355x_graph *graph)
{
    REQUIRE_CTX();
    REQUIRE(graph != nullptr, "null graph");
    return check(hipGraphLaunch(graph->exec, ctx->stream), "hipGraphLaunch");
}

void havoc_mi355x_graph_destroy(havoc_mi355x_graph *graph)
{
    if (!graph) return;
    (void)hipDeviceSynchronize();   // the executable graph must be idle on every internal stream before it goes
    (void)hipGraphExecDestroy(graph->exec);
    (void)hipGraphDestroy(graph->graph);
    delete graph;
}

// ---- fork / join: overlap independent launch chains ---------------------------------------------------------

int havoc_mi355x_fork(havoc_mi355x_ctx *ctx, int nlanes)
{
    REQUIRE_CTX();
    REQUIRE(nlanes >= 1 && nlanes <= havoc_mi355x_ctx::kMaxLanes, "nlanes must be 1..8");
    REQUIRE(ctx->nlanes == 0, "already forked");
    int rc;
    if ((rc = ensure_lanes(ctx, nlanes))) return rc;
    if ((rc = check(hipEventRecord(ctx->forkEv, ctx->stream), "hipEventRecord"))) return rc;
    for (int k = 1; k < nlanes; ++k)
        if ((rc = check(hipStreamWaitEvent(ctx->lanes[k], ctx->forkEv, 0), "hipStreamWaitEvent"))) return rc;
    ctx->nlanes = nlanes;
    ctx->cur = 0;
    return 0;
}

int havoc_mi355x_lane(havoc_mi355x_ctx *ctx, int lane)
{
    REQUIRE_CTX();
    REQUIRE(lane >= 0 && lane < (ctx->nlanes ? ctx->nlanes : 1), "lane out of range (fork first)");
    ctx->cur = lane;
    return 0;
}

int havoc_mi355x_join(havoc_mi355x_ctx *ctx)
{
    REQUIRE_CTX();
    int rc;
    for (int k = 1; k < ctx->nlanes; ++k)
    {
        if ((rc = check(hipEventRecord(ctx->laneEv[k], ctx->lanes[k]), "hipEventRecord"))) return rc;
        if ((rc = check(hipStreamWaitEvent(ctx->stream, ctx->laneEv[k], 0), "hipStreamWaitEvent"))) return rc;
    }
    ctx->nlanes = 0;
    ctx->cur = 0;
    return 0;
}

// ---- distortion metrics -----------------------------------------------------------------------------------

int havoc_mi355x_sad(havoc_mi355x_ctx *ctx, int S, const void *d_src, intptr_t stride_src, const void *d_ref, intptr_t stride_ref,
                     const havoc_mi355x_pair_job *d_jobs, int njobs, int32_t *d_out)
{
    REQUIRE_CTX(); REQUIRE_S(); REQUIRE(njobs >= 0, "njobs < 0");
    return check(launch_sad(LS(ctx),S, 1, d_src, stride_src, d_ref, stride_ref, d_jobs, njobs, d_out), "sad");
}

int havoc_mi355x_sad4(havoc_mi355x_ctx *ctx, int S, const void *d_src, intptr_t stride_src, const void *d_ref, intptr_t stride_ref,
                      const havoc_mi355x_sad4_job *d_jobs, int njobs, int32_t *d_out)
{
    REQUIRE_CTX(); REQUIRE_S(); REQUIRE(njobs >= 0, "njobs < 0");
    return check(launch_sad(LS(ctx),S, 4, d_src, stride_src, d_ref, stride_ref, d_jobs, njobs, d_out), "sad4");
}

int havoc_mi355x_sad4_runs(havoc_mi355x_ctx *ctx, int S, const void *d_src, intptr_t stride_src, const void *d_ref, intptr_t stride_ref,
                           const havoc_mi355x_sad4_job *d_jobs, int njobs, const havoc_mi355x_sad4_run *d_runs, int nruns, int32_t *d_out)
{
    REQUIRE_CTX(); REQUIRE_S(); REQUIRE(njobs >= 0 && nruns >= 0, "njobs / nruns < 0");
    REQUIRE(stride_ref >= 64 && stride_ref < (1 << 22), "sad4_runs: reference stride must be 64 .. 2^22 - 1 samples");
    REQUIRE(njobs == 0 || nruns == 0 || (d_src && d_ref && d_jobs && d_runs && d_out), "sad4_runs: null device pointer");
    return check(launch_sad4_runs(LS(ctx), S, d_src, stride_src, d_ref, stride_ref, d_jobs, njobs, d_runs, nruns, d_out), "sad4_runs");
}

int havoc_mi355x_sad4_make_runs(const havoc_mi355x_sad4_job *jobs, int njobs, int max_run, intptr_t stride_ref, int S, havoc_mi355x_sad4_run *runs)
{
    if (!jobs || !runs || njobs < 0 || (S != 1 && S != 2)) return -1;
    const long st = (long)stride_ref, half = st >> 1;
    const long budget = (S == 1 ? 16 : 32) * 1024 - 64;      // the kernel's window in bytes (csrc/kernels_metric.hip: k_sad4r), less the dwords it keeps spare
    int n = 0;
    for (int i = 0; i < njobs;)
    {
        const havoc_mi355x_sad4_job &a = jobs[i];
        const long area = (long)a.w * a.h;
        // calls per run by block size (what keeps the workgroups' work even: a 64x64 call is sixteen 16x16 calls' worth); HAVOC_SAD4_CAPS="big,middle,small" overrides (experiments)
        static const struct Caps { int big = 16, mid = 48, small = 128; Caps() { if (const char *e = getenv("HAVOC_SAD4_CAPS")) sscanf(e, "%d,%d,%d", &big, &mid, &small); } } caps;
        const int byArea = area >= 4096 ? caps.big : area >= 1024 ? caps.mid : caps.small;
        const int cap = max_run >= 1 ? (max_run > 128 ? 128 : max_run) : (byArea < 1 ? 1 : byArea > 128 ? 128 : byArea);
        // the box grows call by call, in displacements (dx, dy) from the run's first candidate: |dx| <= stride / 2 makes the split of an offset unique
        long mnx = 0, mxx = 0, mny = 0, mxy = 0;
        bool boxed = st >= 64 && a.w > 0 && a.h > 0 && a.w <= 64 && a.h <= 64;
        int e = i;
        while (e < njobs && e - i < cap && jobs[e].src_off == a.src_off && jobs[e].w == a.w && jobs[e].h == a.h)
        {
            long nx0 = mnx, nx1 = mxx, ny0 = mny, ny1 = mxy;
            if (boxed)
                for (int k = 0; k < 4; ++k)
                {
                    const long delta = (long)jobs[e].ref_off[k] - a.ref_off[0];
                    long q = (delta + half >= 0 ? (delta + half) / st : -((-(delta + half) + st - 1) / st)), r = delta - q * st;
                    nx0 = r < nx0 ? r : nx0; nx1 = r > nx1 ? r : nx1; ny0 = q < ny0 ? q : ny0; ny1 = q > ny1 ? q : ny1;
                }
            // bytes of the staged box: rows at a 16-byte granularity (+ up to 15 bytes of alignment lead + the odd dword of the LDS pitch)
            const long pitch = ((15 + (nx1 - nx0 + a.w) * S + 15) / 16) * 16 + 4, bytes = pitch * (ny1 - ny0 + a.h);
            const bool holds = boxed && nx1 - nx0 + a.w < st && bytes <= budget;
            if (boxed && !holds && e > i) break;      // this call would burst the box: it opens the next run
            if (!holds && st >= 64)                   // a single call whose own four candidates do not fit: a run of one, without a box (the kernel takes it call by call)
            {
                boxed = false;
                ++e;
                break;
            }
            if (!holds) boxed = false;                // (no stride given: runs without boxes, cut by source block and length only)
            mnx = nx0; mxx = nx1; mny = ny0; mxy = ny1;
            ++e;
        }
        const long off = (long)a.ref_off[0] + mny * st + mnx;
        havoc_mi355x_sad4_run &r = runs[n++];
        r = havoc_mi355x_sad4_run();
        r.first_job = i;
        r.count = e - i;
        if (boxed && off >= 0 && off < (1l << 31))
        {
            r.box_off = (int32_t)off;
            r.box_w = (int32_t)(mxx - mnx + a.w);
            r.box_h = (int32_t)(mxy - mny + a.h);
        }
        i = e;
    }
    return n;
}

int havoc_mi355x_sad_surface(havoc_mi355x_ctx *ctx, int S, int range, int max_w, int max_h, const void *d_src, intptr_t stride_src,
                             const void *d_ref, intptr_t stride_ref, const havoc_mi355x_surface_job *d_jobs, int njobs, int32_t *d_out)
{
    REQUIRE_CTX(); REQUIRE_S(); REQUIRE(njobs >= 0, "njobs < 0");
    REQUIRE(range >= 0 && range <= 96, "range must be 0..96");
    REQUIRE(max_w >= 4 && max_w <= 64 && (max_w & 3) == 0 && max_h >= 1 && max_h <= 64, "max_w must be 4..64 and a multiple of 4, max_h 1..64");
    return check(launch_sad_surface(LS(ctx), S, range, max_w, max_h, d_src, stride_src, d_ref, stride_ref, d_jobs, njobs, d_out), "sad_surface");
}

int havoc_mi355x_ssd(havoc_mi355x_ctx *ctx, int S, const void *d_a, intptr_t stride_a, const void *d_b, intptr_t stride_b,
                     const havoc_mi355x_pair_job *d_jobs, int njobs, uint32_t *d_out)
{
    REQUIRE_CTX(); REQUIRE_S(); REQUIRE(njobs >= 0, "njobs < 0");
    return check(launch_ssd(LS(ctx),S, d_a, stride_a, d_b, stride_b, d_jobs, njobs, d_out), "ssd");
}

int havoc_mi355x_satd(havoc_mi355x_ctx *ctx, int S, int max_w, int max_h, const void *d_a, intptr_t stride_a, const void *d_b, intptr_t stride_b,
                      const havoc_mi355x_pair_job *d_jobs, int njobs, int32_t *d_out)
{
    REQUIRE_CTX(); REQUIRE_S(); REQUIRE(njobs >= 0, "njobs < 0");
    REQUIRE(max_w >= 2 && max_w <= 64 && max_h >= 2 && max_h <= 64, "max_w / max_h must be 2..64");
    return check(launch_satd(LS(ctx),S, max_w, max_h, d_a, stride_a, d_b, stride_b, d_jobs, njobs, d_out), "satd");
}

int havoc_mi355x_satd_multi(havoc_mi355x_ctx *ctx, int S, int max_w, int max_h, const void *d_a, intptr_t stride_a, const void *d_b,
                            intptr_t stride_b, const havoc_mi355x_satd_multi_job *d_jobs, int njobs, int32_t *d_out)
{
    REQUIRE_CTX(); REQUIRE_S(); REQUIRE(njobs >= 0, "njobs < 0");
    REQUIRE(max_w >= 2 && max_w <= 64 && max_h >= 2 && max_h <= 64, "max_w / max_h must be 2..64");
    return check(launch_satd_multi(LS(ctx), S, max_w, max_h, d_a, stride_a, d_b, stride_b, d_jobs, njobs, d_out), "satd_multi");
}

int havoc_mi355x_pad_block(havoc_mi355x_ctx *ctx, int S, void *d_plane, int64_t origin_off, int width, int height, intptr_t stride, int pad, int top,
                           int bottom, int left, int right)
{
    REQUIRE_CTX(); REQUIRE_S();
    REQUIRE(width > 0 && height > 0 && pad >= 0, "width / height must be positive, pad >= 0");
    return check(launch_pad_block(LS(ctx), S, d_plane, (long)origin_off, width, height, stride, pad, top, bottom, left, right), "pad_block");
}

int havoc_mi355x_derive_bs(havoc_mi355x_ctx *ctx, const havoc_mi355x_cell *d_cells, intptr_t cells_stride, int width, int height, int8_t *d_block_data,
                           uint8_t *d_block_bs)
{
    REQUIRE_CTX();
    REQUIRE(d_cells && d_block_data && d_block_bs, "null pointer");
    REQUIRE(width > 0 && height > 0 && (width & 7) == 0 && (height & 7) == 0 && cells_stride >= width / 4, "picture size must be a multiple of 8 (minimum coding unit)");
    return check(launch_derive_bs(LS(ctx), d_cells, (long)cells_stride, width, height, d_block_data, d_block_bs), "derive_bs");
}

int havoc_mi355x_deblock(havoc_mi355x_ctx *ctx, int S, int bitDepth, void *d_luma, intptr_t stride_luma, void *d_cb, void *d_cr, intptr_t stride_chroma,
                         int width, int height, const int8_t *d_block_data, const uint8_t *d_block_bs, int tc_offset_div2, int beta_offset_div2,
                         int cb_qp_offset, int cr_qp_offset)
{
    REQUIRE_CTX(); REQUIRE_S(); REQUIRE_BD();
    REQUIRE(width >= 8 && height >= 8 && (width & 7) == 0 && (height & 7) == 0, "width / height must be multiples of 8 (minimum coding block)");
    REQUIRE(d_luma && d_cb && d_cr && d_block_data && d_block_bs, "null plane / block map");
    REQUIRE(tc_offset_div2 >= -6 && tc_offset_div2 <= 6 && beta_offset_div2 >= -6 && beta_offset_div2 <= 6, "slice offsets must be -6..6");
    return check(launch_deblock(LS(ctx), S, bitDepth, d_luma, stride_luma, d_cb, d_cr, stride_chroma, width, height, d_block_data, d_block_bs,
                                tc_offset_div2, beta_offset_div2, cb_qp_offset, cr_qp_offset),
                 "deblock");
}

int havoc_mi355x_ssd_linear(havoc_mi355x_ctx *ctx, const uint8_t *d_a, const uint8_t *d_b, int size, int32_t *d_out)
{
    REQUIRE_CTX(); REQUIRE(size >= 0, "size < 0");
    return check(launch_ssd_linear(LS(ctx),d_a, d_b, size, d_out), "ssd_linear");
}

// ---- inter prediction -------------------------------------------------------------------------------------

#define REQUIRE_MAXWH() REQUIRE(max_w >= 2 && max_w <= 64 && max_h >= 2 && max_h <= 64, "max_w / max_h must be 2..64")

int havoc_mi355x_pred_uni(havoc_mi355x_ctx *ctx, int S, int taps, int bitDepth, int max_w, int max_h, void *d_dst, intptr_t stride_dst,
                          const void *d_ref, intptr_t stride_ref, const havoc_mi355x_pred_uni_job *d_jobs, int njobs)
{
    REQUIRE_CTX(); REQUIRE_S(); REQUIRE_BD(); REQUIRE(taps == 8 || taps == 4, "taps must be 8 or 4"); REQUIRE(njobs >= 0, "njobs < 0");
    REQUIRE_MAXWH();
    return check(launch_pred_uni(LS(ctx),S, taps, bitDepth, max_w, max_h, d_dst, stride_dst, d_ref, stride_ref, d_jobs, njobs), "pred_uni");
}

int havoc_mi355x_pred_bi(havoc_mi355x_ctx *ctx, int S, int taps, int bitDepth, int max_w, int max_h, void *d_dst, intptr_t stride_dst,
                         const void *d_ref, intptr_t stride_ref, const havoc_mi355x_pred_bi_job *d_jobs, int njobs)
{
    REQUIRE_CTX(); REQUIRE_S(); REQUIRE_BD(); REQUIRE(taps == 8 || taps == 4, "taps must be 8 or 4"); REQUIRE(njobs >= 0, "njobs < 0");
    REQUIRE_MAXWH();
    return check(launch_pred_bi(LS(ctx),S, taps, bitDepth, max_w, max_h, d_dst, stride_dst, d_ref, stride_ref, d_jobs, njobs), "pred_bi");
}

int havoc_mi355x_pred_uni_classes(havoc_mi355x_ctx *ctx, int S, int taps, int bitDepth, void *d_dst, intptr_t stride_dst, const void *d_ref, intptr_t stride_ref,
                                  const havoc_mi355x_pred_uni_job *d_jobs, const int32_t count[4])
{
    REQUIRE_CTX(); REQUIRE_S(); REQUIRE_BD(); REQUIRE(taps == 8 || taps == 4, "taps must be 8 or 4"); REQUIRE(count != nullptr, "null count");
    const int c[4] = {count[0], count[1], count[2], count[3]};
    return check(launch_pred_classes(LS(ctx), 0, S, taps, bitDepth, d_dst, stride_dst, d_ref, stride_ref, d_jobs, c), "pred_uni_classes");
}

int havoc_mi355x_pred_bi_classes(havoc_mi355x_ctx *ctx, int S, int taps, int bitDepth, void *d_dst, intptr_t stride_dst, const void *d_ref, intptr_t stride_ref,
                                 const havoc_mi355x_pred_bi_job *d_jobs, const int32_t count[4])
{
    REQUIRE_CTX(); REQUIRE_S(); REQUIRE_BD(); REQUIRE(taps == 8 || taps == 4, "taps must be 8 or 4"); REQUIRE(count != nullptr, "null count");
    const int c[4] = {count[0], count[1], count[2], count[3]};
    return check(launch_pred_classes(LS(ctx), 1, S, taps, bitDepth, d_dst, stride_dst, d_ref, stride_ref, d_jobs, c), "pred_bi_classes");
}

int havoc_mi355x_subtract_bi(havoc_mi355x_ctx *ctx, int S, int bitDepth, void *d_dst, intptr_t stride_dst, const void *d_pred, intptr_t stride_pred,
                             const void *d_src, intptr_t stride_src, const havoc_mi355x_subtract_bi_job *d_jobs, int njobs)
{
    REQUIRE_CTX(); REQUIRE_S(); REQUIRE_BD(); REQUIRE(njobs >= 0, "njobs < 0");
    return check(launch_subtract_bi(LS(ctx),S, bitDepth, d_dst, stride_dst, d_pred, stride_pred, d_src, stride_src, d_jobs, njobs), "subtract_bi");
}

int havoc_mi355x_interp_planes(havoc_mi355x_ctx *ctx, int S, int bitDepth, void *d_planes, intptr_t plane_elems, const void *d_ref, intptr_t stride,
                               int x0, int y0, int width, int height)
{
    REQUIRE_CTX(); REQUIRE_S(); REQUIRE_BD(); REQUIRE(x0 >= 3 && y0 >= 3 && width >= 0 && height >= 0, "rectangle must start >= 3 samples in");
    REQUIRE(x0 + width + 12 <= stride, "rectangle must end >= 12 samples before the row end");
    return check(launch_interp_planes(LS(ctx), S, bitDepth, d_planes, plane_elems, d_ref, stride, x0, y0, width, height), "interp_planes");
}

int havoc_mi355x_subpel_satd(havoc_mi355x_ctx *ctx, int S, int taps, int bitDepth, int max_w, int max_h, const void *d_src, intptr_t stride_src,
                             const void *d_ref, intptr_t stride_ref, const havoc_mi355x_pred_uni_job *d_jobs, int njobs, int32_t *d_cost)
{
    REQUIRE_CTX(); REQUIRE_S(); REQUIRE_BD(); REQUIRE(taps == 8 || taps == 4, "taps must be 8 or 4"); REQUIRE(njobs >= 0, "njobs < 0");
    REQUIRE(max_w >= 2 && max_w <= 64 && max_h >= 2 && max_h <= 64, "max_w / max_h must be 2..64");
    return check(launch_subpel_satd(LS(ctx),S, taps, bitDepth, max_w, max_h, d_src, stride_src, d_ref, stride_ref, d_jobs, njobs, d_cost),
                 "subpel_satd");
}

// ---- intra prediction -------------------------------------------------------------------------------------

int havoc_mi355x_intra(havoc_mi355x_ctx *ctx, int S, int bitDepth, int log2TrafoSize, void *d_dst, intptr_t stride_dst, const void *d_neighbours,
                       const havoc_mi355x_intra_job *d_jobs, int njobs)
{
    REQUIRE_CTX(); REQUIRE_S(); REQUIRE_BD(); REQUIRE(log2TrafoSize >= 2 && log2TrafoSize <= 5, "log2TrafoSize must be 2..5");
    REQUIRE(njobs >= 0, "njobs < 0");
    return check(launch_intra(LS(ctx),S, log2TrafoSize, bitDepth, d_dst, stride_dst, d_neighbours, d_jobs, njobs), "intra");
}

int havoc_mi355x_intra_satd35(havoc_mi355x_ctx *ctx, int S, int bitDepth, int log2TrafoSize, const void *d_src, intptr_t stride_src,
                              const void *d_neighbours, const havoc_mi355x_intra_search_job *d_jobs, int njobs, int32_t *d_cost)
{
    REQUIRE_CTX(); REQUIRE_S(); REQUIRE_BD(); REQUIRE(log2TrafoSize >= 2 && log2TrafoSize <= 5, "log2TrafoSize must be 2..5");
    REQUIRE(njobs >= 0, "njobs < 0");
    return check(launch_intra_satd35(LS(ctx),S, log2TrafoSize, bitDepth, d_src, stride_src, d_neighbours, d_jobs, njobs, d_cost), "intra_satd35");
}

// ---- residual, transforms, quantisation -------------------------------------------------------------------

int havoc_mi355x_residual(havoc_mi355x_ctx *ctx, int S, int16_t *d_res, intptr_t stride_res, const int32_t *d_res_off, const void *d_src,
                          intptr_t stride_src, const void *d_pred, intptr_t stride_pred, const havoc_mi355x_pair_job *d_jobs, int njobs)
{
    REQUIRE_CTX(); REQUIRE_S(); REQUIRE(njobs >= 0, "njobs < 0");
    return check(launch_residual(LS(ctx),S, d_res, stride_res, d_res_off, d_src, stride_src, d_pred, stride_pred, d_jobs, njobs), "residual");
}

#define REQUIRE_TR() \
    REQUIRE(log2TrafoSize >= 2 && log2TrafoSize <= 5, "log2TrafoSize must be 2..5"); \
    REQUIRE(trType == 0 || (trType == 1 && log2TrafoSize == 2), "trType 1 (DST) requires log2TrafoSize 2")

int havoc_mi355x_transform(havoc_mi355x_ctx *ctx, int bitDepth, int trType, int log2TrafoSize, int16_t *d_coeffs, const int16_t *d_res,
                           intptr_t stride_res, const havoc_mi355x_tu_job *d_jobs, int njobs)
{
    REQUIRE_CTX(); REQUIRE_TR(); REQUIRE(bitDepth >= 8 && bitDepth <= 10, "bitDepth must be 8..10"); REQUIRE(njobs >= 0, "njobs < 0");
    return check(launch_transform(LS(ctx),bitDepth, log2TrafoSize, trType, d_coeffs, d_res, stride_res, d_jobs, njobs), "transform");
}

int havoc_mi355x_inverse_transform(havoc_mi355x_ctx *ctx, int bitDepth, int trType, int log2TrafoSize, int16_t *d_res, const int16_t *d_coeffs,
                                   const havoc_mi355x_tu_job *d_jobs, int njobs)
{
    REQUIRE_CTX(); REQUIRE_TR(); REQUIRE(bitDepth >= 8 && bitDepth <= 10, "bitDepth must be 8..10"); REQUIRE(njobs >= 0, "njobs < 0");
    return check(launch_inverse_transform(LS(ctx),0, bitDepth, log2TrafoSize, trType, nullptr, 0, nullptr, 0, d_res, d_coeffs, d_jobs, njobs),
                 "inverse_transform");
}

int havoc_mi355x_inverse_transform_add(havoc_mi355x_ctx *ctx, int S, int bitDepth, int trType, int log2TrafoSize, void *d_dst, intptr_t stride_dst,
                                       const void *d_pred, intptr_t stride_pred, const int16_t *d_coeffs, const havoc_mi355x_tu_job *d_jobs,
                                       int njobs)
{
    REQUIRE_CTX(); REQUIRE_S(); REQUIRE_BD(); REQUIRE_TR(); REQUIRE(njobs >= 0, "njobs < 0");
    return check(launch_inverse_transform(LS(ctx),S, bitDepth, log2TrafoSize, trType, d_dst, stride_dst, d_pred, stride_pred, nullptr, d_coeffs,
                                          d_jobs, njobs),
                 "inverse_transform_add");
}

int havoc_mi355x_tu_forward(havoc_mi355x_ctx *ctx, int S, int bitDepth, int trType, int log2TrafoSize, int16_t *d_coeffs, const void *d_src,
                            intptr_t stride_src, const void *d_pred, intptr_t stride_pred, const havoc_mi355x_tu_fused_job *d_jobs, int njobs)
{
    REQUIRE_CTX(); REQUIRE_S(); REQUIRE_BD(); REQUIRE_TR(); REQUIRE(njobs >= 0, "njobs < 0");
    return check(launch_tu_forward(LS(ctx), S, bitDepth, log2TrafoSize, trType, d_coeffs, d_src, stride_src, d_pred, stride_pred, d_jobs, njobs),
                 "tu_forward");
}

int havoc_mi355x_intra_measure(havoc_mi355x_ctx *ctx, int S, int bitDepth, int log2TrafoSize, int16_t *d_coeffs, int16_t *d_coeffs_dct, int32_t *d_satd, void *d_rec0,
                               uint32_t *d_ssd0, const void *d_src, intptr_t stride_src, const void *d_pred, intptr_t stride_pred, const havoc_mi355x_tu_fused_job *d_jobs,
                               int njobs, int with_satd)
{
    REQUIRE_CTX(); REQUIRE_S(); REQUIRE(log2TrafoSize >= 2 && log2TrafoSize <= 5, "log2TrafoSize must be 2..5"); REQUIRE(njobs >= 0, "njobs < 0");
    REQUIRE(bitDepth >= 8 && bitDepth <= (S == 1 ? 8 : 10), "bit depth");
    REQUIRE(njobs == 0 || (d_coeffs && d_rec0 && d_ssd0 && d_src && d_pred && d_jobs && (log2TrafoSize != 2 || d_coeffs_dct) && (!with_satd || d_satd)), "intra_measure: null device pointer");
    return check(launch_intra_measure(LS(ctx), S, bitDepth, log2TrafoSize, d_coeffs, d_coeffs_dct, d_satd, d_rec0, d_ssd0, d_src, stride_src, d_pred, stride_pred, d_jobs, njobs,
                                      with_satd),
                 "intra_measure");
}

int havoc_mi355x_tu_reconstruct(havoc_mi355x_ctx *ctx, int S, int bitDepth, int trType, int log2TrafoSize, int scale, int shift, void *d_rec,
                                intptr_t stride_rec, const void *d_pred, intptr_t stride_pred, const void *d_src, intptr_t stride_src,
                                const int16_t *d_levels, const havoc_mi355x_tu_fused_job *d_jobs, int njobs, uint32_t *d_ssd)
{
    REQUIRE_CTX(); REQUIRE_S(); REQUIRE_BD(); REQUIRE_TR(); REQUIRE(njobs >= 0, "njobs < 0");
    REQUIRE(shift >= 1 && shift <= 30 && scale > 0, "scale / shift out of range");
    return check(launch_tu_reconstruct(LS(ctx), S, bitDepth, log2TrafoSize, trType, scale, shift, d_rec, stride_rec, d_pred, stride_pred, d_src,
                                       stride_src, d_levels, d_jobs, njobs, d_ssd),
                 "tu_reconstruct");
}

int havoc_mi355x_level_stats(havoc_mi355x_ctx *ctx, const int16_t *d_levels, const int32_t *d_jobs, int njobs, int32_t *d_out)
{
    REQUIRE_CTX(); REQUIRE(njobs >= 0, "njobs < 0");
    return check(launch_level_stats(LS(ctx), d_levels, d_jobs, njobs, d_out), "level_stats");
}

int havoc_mi355x_search_motion_uni(havoc_mi355x_ctx *ctx, int S, const havoc_mi355x_search_params *params, const void *d_src, int64_t src_origin, intptr_t src_stride,
                                   const void *d_ref, int64_t ref_origin, intptr_t ref_stride, const void *d_phase, intptr_t plane_elems, int64_t phase_origin,
                                   const void *d_pus, int n, void *d_out)
{
    REQUIRE_CTX(); REQUIRE_S();
    REQUIRE(params, "null argument"); REQUIRE(n >= 0, "n < 0");
    REQUIRE(n == 0 || (d_src && d_ref && d_phase && d_pus && d_out), "null device pointer");
    REQUIRE(params->bit_depth >= 8 && params->bit_depth <= (S == 1 ? 8 : 10), "bit_depth must be 8 (S=1) or 8..10 (S=2)");
    return check(launch_search_list(LS(ctx), S, params, d_src, (long)src_origin, src_stride, d_ref, (long)ref_origin, ref_stride, d_phase, plane_elems, (long)phase_origin,
                                    d_pus, n, d_out),
                 "search_motion_uni");
}

int havoc_mi355x_search_motion_bi(havoc_mi355x_ctx *ctx, int S, const havoc_mi355x_search_params *params, const void *d_src, int64_t src_origin, intptr_t src_stride,
                                  const void *d_ref, int64_t ref_origin, intptr_t ref_stride, const void *d_phase, intptr_t plane_elems, int64_t phase_origin,
                                  const void *d_phase_other, int64_t phase_other_origin, const void *d_pus, const int16_t *d_start, int n, void *d_out)
{
    REQUIRE_CTX(); REQUIRE_S();
    REQUIRE(params, "null argument"); REQUIRE(n >= 0, "n < 0");
    REQUIRE(n == 0 || (d_src && d_ref && d_phase && d_phase_other && d_pus && d_start && d_out), "null device pointer");
    REQUIRE(params->bit_depth >= 8 && params->bit_depth <= (S == 1 ? 8 : 10), "bit_depth must be 8 (S=1) or 8..10 (S=2)");
    return check(launch_search_bi_list(LS(ctx), S, params, d_src, (long)src_origin, src_stride, d_ref, (long)ref_origin, ref_stride, d_phase, plane_elems, (long)phase_origin,
                                       d_phase_other, (long)phase_other_origin, d_pus, d_start, n, d_out),
                 "search_motion_bi");
}

size_t havoc_mi355x_search_workspace(int width, int height) { return width > 0 && height > 0 ? search_workspace_bytes(width, height) : 0; }

int havoc_mi355x_search_picture_uni(havoc_mi355x_ctx *ctx, int S, const havoc_mi355x_search_params *params, const int64_t mvp_rate[2], const void *d_src,
                                    int64_t src_origin, intptr_t src_stride, const void *d_ref, const int64_t ref_origin[2], intptr_t ref_stride, const void *d_phase,
                                    intptr_t plane_elems, const int64_t phase_origin[2], const void *d_pus, const int32_t *d_ctu_first, int ctus_x, int ctus_y,
                                    int n_pus, void *d_out, void *d_out_bi, int16_t *d_field, void *d_work, int step_launches)
{
    REQUIRE_CTX(); REQUIRE_S();
    REQUIRE(n_pus >= 0, "n_pus < 0");
    REQUIRE(params && mvp_rate && ref_origin && phase_origin, "null argument");
    REQUIRE(d_src && d_ref && d_phase && d_pus && d_ctu_first && d_out && d_field && d_work, "null device pointer");
    REQUIRE(((uintptr_t)d_field & 3) == 0 && ((uintptr_t)d_work & 15) == 0, "d_field must be 4-byte aligned, d_work 16-byte aligned");
    REQUIRE(mvp_rate[0] >= 0 && mvp_rate[1] >= 0, "mvp_rate must not be negative (the device compares costs as non-negative numbers)");
    REQUIRE(params->ctb_size == 64, "ctb_size must be 64");
    REQUIRE(params->pic_width > 0 && params->pic_height > 0 && params->pic_width % 8 == 0 && params->pic_height % 8 == 0, "picture size must be a positive multiple of 8");
    REQUIRE(ctus_x == (params->pic_width + 63) / 64 && ctus_y == (params->pic_height + 63) / 64, "ctus_x / ctus_y do not match the picture size");
    REQUIRE(params->bit_depth >= 8 && params->bit_depth <= (S == 1 ? 8 : 10), "bit_depth must be 8 (S=1) or 8..10 (S=2)");
    REQUIRE(!ctx->searchGate || (params->concurrent_frames > 1 && !step_launches),
            "search gate: needs concurrent_frames > 1 (only then are the vectors of a CTU row limited to rows the gate can promise) and the one-launch form");
    const long ro[2] = {(long)ref_origin[0], (long)ref_origin[1]}, po[2] = {(long)phase_origin[0], (long)phase_origin[1]};
    return check(launch_search_picture_uni(LS(ctx), S, params, mvp_rate, d_src, (long)src_origin, src_stride, d_ref, ro, ref_stride, d_phase, plane_elems, po, d_pus,
                                           d_ctu_first, ctus_x, ctus_y, n_pus, d_out, d_out_bi, d_field, d_work, step_launches, ctx->searchGate),
                 "search_picture_uni");
}

int havoc_mi355x_search_gate(havoc_mi355x_ctx *ctx, const int32_t *d_rows_ready)
{
    REQUIRE_CTX();
    REQUIRE(((uintptr_t)d_rows_ready & 3) == 0, "d_rows_ready must be 4-byte aligned");
    ctx->searchGate = d_rows_ready;
    return 0;
}

static bool layout_ok(const havoc_mi355x_field_layout *l)
{
    return l && l->pic_width > 0 && l->pic_height > 0 && l->range >= 0 && l->field_cw >= (l->pic_width + 3) / 4 && l->luma_stride >= l->pic_width + 2 * l->luma_pad &&
           l->chroma_stride >= l->pic_width / 2 + 2 * l->chroma_pad && l->luma_pad >= l->range + 4 && 2 * l->chroma_pad >= l->range + 4;
}

int havoc_mi355x_merge_jobs(havoc_mi355x_ctx *ctx, const havoc_mi355x_field_layout *layout, const int16_t *d_field, const int32_t *d_x0, const int32_t *d_y0, int n, int log2_size,
                            havoc_mi355x_pred_bi_job *d_luma_jobs, havoc_mi355x_pred_bi_job *d_cb_jobs, havoc_mi355x_pred_bi_job *d_cr_jobs, int16_t *d_vectors)
{
    REQUIRE_CTX(); REQUIRE(n >= 0, "n < 0"); REQUIRE(log2_size >= 3 && log2_size <= 6, "log2_size must be 3..6");
    REQUIRE(layout_ok(layout), "field layout: sizes, strides or borders do not fit");
    REQUIRE(n == 0 || (d_field && d_x0 && d_y0 && d_luma_jobs && d_cb_jobs && d_cr_jobs && d_vectors), "null device pointer");
    REQUIRE(((uintptr_t)d_field & 3) == 0 && ((uintptr_t)d_vectors & 3) == 0, "d_field and d_vectors must be 4-byte aligned");
    return check(launch_merge_jobs(LS(ctx), layout, d_field, d_x0, d_y0, n, log2_size, d_luma_jobs, d_cb_jobs, d_cr_jobs, d_vectors), "merge_jobs");
}

int havoc_mi355x_rqt_decide(havoc_mi355x_ctx *ctx, const havoc_mi355x_rqt_unit *d_units, int n, const int32_t *d_zero_at, const int32_t *d_one_at, const havoc_mi355x_rqt_size sizes[4],
                            int64_t rec_origin, intptr_t rec_stride, int32_t dump_off, int32_t reciprocal_lambda_q16, havoc_mi355x_rqt_choice *d_out)
{
    REQUIRE_CTX(); REQUIRE(n >= 0, "n < 0"); REQUIRE(sizes != nullptr, "null sizes"); REQUIRE(rec_stride > 0 && rec_stride < (1 << 24), "rec_stride out of range");
    REQUIRE(n == 0 || (d_units && d_zero_at && d_one_at && d_out), "null device pointer");
    REQUIRE(reciprocal_lambda_q16 >= 0, "reciprocal_lambda_q16 < 0"); REQUIRE(dump_off >= 0, "dump_off < 0");
    if (n > 0)      // units of 8x8 .. 32x32 read the tables of their own size and of half of it: sizes 4 .. 32 = sizes[0 .. 3]; a table nobody can need may be null,
        for (int k = 0; k < 4; ++k)      // a partly filled one is a caller's mistake
        {
            const havoc_mi355x_rqt_size &z = sizes[k];
            const bool all = z.d_cbf && z.d_ssd && z.d_stats && z.d_jobs && z.d_final, none = !z.d_cbf && !z.d_ssd && !z.d_stats && !z.d_jobs && !z.d_final;
            REQUIRE(all || none, "rqt_decide: a size table with some null pointers");
        }
    return check(launch_rqt_decide(LS(ctx), d_units, n, d_zero_at, d_one_at, sizes, (long)rec_origin, (int)rec_stride, dump_off, reciprocal_lambda_q16, d_out), "rqt_decide");
}

int havoc_mi355x_block_cells(havoc_mi355x_ctx *ctx, int width, int height, int qp, int dpb_index0, const int16_t *d_field, const havoc_mi355x_rqt_unit *d_units,
                             const havoc_mi355x_rqt_choice *d_decisions, int n, havoc_mi355x_cell *d_cells)
{
    REQUIRE_CTX(); REQUIRE(n >= 0, "n < 0"); REQUIRE(width > 0 && height > 0 && !(width & 3) && !(height & 3), "width / height must be positive multiples of 4");
    REQUIRE(d_cells && (n == 0 || (d_field && d_units && d_decisions)), "null device pointer"); REQUIRE(((uintptr_t)d_field & 3) == 0, "d_field must be 4-byte aligned");
    return check(launch_block_cells(LS(ctx), width, height, qp, dpb_index0, d_field, d_units, d_decisions, n, d_cells, true), "block_cells");
}

int havoc_mi355x_block_cells_add(havoc_mi355x_ctx *ctx, int width, int height, int qp, int dpb_index0, const int16_t *d_field, const havoc_mi355x_rqt_unit *d_units,
                                 const havoc_mi355x_rqt_choice *d_decisions, int n, havoc_mi355x_cell *d_cells)
{
    REQUIRE_CTX(); REQUIRE(n >= 0, "n < 0"); REQUIRE(width > 0 && height > 0 && !(width & 3) && !(height & 3), "width / height must be positive multiples of 4");
    REQUIRE(d_cells && (n == 0 || (d_field && d_units && d_decisions)), "null device pointer"); REQUIRE(((uintptr_t)d_field & 3) == 0, "d_field must be 4-byte aligned");
    return check(launch_block_cells(LS(ctx), width, height, qp, dpb_index0, d_field, d_units, d_decisions, n, d_cells, false), "block_cells_add");
}

int havoc_mi355x_search_wait_rows(havoc_mi355x_ctx *ctx, const void *d_work, int pic_width, int pic_height, int ctu_row, int32_t *d_gave_up)
{
    REQUIRE_CTX();
    REQUIRE(d_work && d_gave_up && ((uintptr_t)d_work & 15) == 0 && ((uintptr_t)d_gave_up & 3) == 0, "search_wait_rows: null or misaligned device pointer");
    REQUIRE(pic_width > 0 && pic_height > 0 && ctu_row >= 0, "search_wait_rows: sizes");
    return check(launch_search_wait_rows(LS(ctx), d_work, pic_width, pic_height, ctu_row, d_gave_up), "search_wait_rows");
}

static bool chain_layout_ok(const havoc_mi355x_intra_chain_layout *l)
{
    return l && l->pic_width > 0 && l->pic_height > 0 && l->pad >= 1 && l->stride >= l->pic_width + 2 * l->pad && l->cells_per_row >= (l->pic_width + 3) / 4 &&
           l->bit_depth >= 8 && l->bit_depth <= 16 && l->ctb_log2 >= 4 && l->ctb_log2 <= 6;
}

int havoc_mi355x_intra_gather(havoc_mi355x_ctx *ctx, int S, const havoc_mi355x_intra_chain_layout *layout, const void *d_rec, const int32_t *d_owner, const uint8_t *d_modes,
                              const havoc_mi355x_intra_chain_part *d_parts, int n, const havoc_mi355x_intra_search_job *d_jobs, void *d_neighbours, havoc_mi355x_intra_mpm *d_mpm)
{
    REQUIRE_CTX(); REQUIRE_S(); REQUIRE(n >= 0, "n < 0"); REQUIRE(chain_layout_ok(layout), "chain layout: sizes, stride or border do not fit");
    REQUIRE(n == 0 || (d_rec && d_owner && d_modes && d_parts && d_jobs && d_neighbours && d_mpm), "null device pointer");
    return check(launch_intra_gather(LS(ctx), S, layout, d_rec, d_owner, d_modes, d_parts, n, d_jobs, d_neighbours, d_mpm), "intra_gather");
}

int havoc_mi355x_intra_commit(havoc_mi355x_ctx *ctx, int S, const havoc_mi355x_intra_chain_layout *layout, void *d_rec, uint8_t *d_modes, const havoc_mi355x_intra_chain_part *d_parts,
                              int n, const void *d_blocks, const int32_t *d_mode, int mode_stride)
{
    REQUIRE_CTX(); REQUIRE_S(); REQUIRE(n >= 0, "n < 0"); REQUIRE(chain_layout_ok(layout), "chain layout: sizes, stride or border do not fit");
    REQUIRE(n == 0 || (d_rec && d_modes && d_parts && d_blocks && d_mode), "null device pointer"); REQUIRE(mode_stride >= 1, "mode_stride < 1");
    return check(launch_intra_commit(LS(ctx), S, layout, d_rec, d_modes, d_parts, n, d_blocks, d_mode, mode_stride), "intra_commit");
}

int havoc_mi355x_intra_fill_spare(havoc_mi355x_ctx *ctx, const int32_t *d_total, int capacity, int log2TrafoSize, havoc_mi355x_intra_job *d_intra_jobs,
                                  havoc_mi355x_tu_fused_job *d_tu_jobs, havoc_mi355x_rdoq_job *d_rdoq_jobs, int32_t *d_stat_jobs, int32_t *d_owner)
{
    REQUIRE_CTX(); REQUIRE(capacity >= 0, "capacity < 0"); REQUIRE(log2TrafoSize >= 2 && log2TrafoSize <= 5, "log2TrafoSize must be 2..5");
    REQUIRE(capacity == 0 || (d_total && d_intra_jobs && d_tu_jobs && d_rdoq_jobs && d_stat_jobs && d_owner), "null device pointer");
    return check(launch_intra_fill_spare(LS(ctx), d_total, capacity, log2TrafoSize, d_intra_jobs, d_tu_jobs, d_rdoq_jobs, d_stat_jobs, d_owner), "intra_fill_spare");
}

int havoc_mi355x_merge_decide(havoc_mi355x_ctx *ctx, const int32_t *d_satd_y, const int32_t *d_satd_cb, const int32_t *d_satd_cr, int n, int64_t reciprocal_sqrt_lambda_q16,
                              int64_t *d_cost, int32_t *d_best)
{
    REQUIRE_CTX(); REQUIRE(n >= 0, "n < 0"); REQUIRE(reciprocal_sqrt_lambda_q16 >= 0, "reciprocal_sqrt_lambda_q16 < 0");
    REQUIRE(n == 0 || (d_satd_y && d_satd_cb && d_satd_cr && d_cost && d_best), "null device pointer");
    REQUIRE(((uintptr_t)d_cost & 7) == 0, "d_cost must be 8-byte aligned");
    return check(launch_merge_decide(LS(ctx), d_satd_y, d_satd_cb, d_satd_cr, n, reciprocal_sqrt_lambda_q16, d_cost, d_best), "merge_decide");
}

int havoc_mi355x_pred_jobs(havoc_mi355x_ctx *ctx, const havoc_mi355x_field_layout *layout, const int16_t *d_field, int list, const int32_t *d_x0, const int32_t *d_y0, int n,
                           int log2_size, int plane, const int32_t *d_dst_off, havoc_mi355x_pred_uni_job *d_jobs)
{
    REQUIRE_CTX(); REQUIRE(n >= 0, "n < 0"); REQUIRE(log2_size >= 3 && log2_size <= 6, "log2_size must be 3..6");
    REQUIRE(list == 0 || list == 1, "list must be 0 or 1"); REQUIRE(plane >= 0 && plane <= 2, "plane must be 0..2");
    REQUIRE(layout_ok(layout), "field layout: sizes, strides or borders do not fit");
    REQUIRE(n == 0 || (d_field && d_x0 && d_y0 && d_dst_off && d_jobs), "null device pointer");
    REQUIRE(((uintptr_t)d_field & 3) == 0, "d_field must be 4-byte aligned");
    return check(launch_pred_jobs(LS(ctx), layout, d_field, list, d_x0, d_y0, n, log2_size, plane, d_dst_off, d_jobs), "pred_jobs");
}

int havoc_mi355x_intra_order(havoc_mi355x_ctx *ctx, const int32_t *d_satd35, const havoc_mi355x_intra_mpm *d_mpm, int n, int32_t lambda_q16, int32_t *d_order,
                             int32_t *d_count, int32_t *d_slot, int32_t *d_total)
{
    REQUIRE_CTX(); REQUIRE(n >= 0, "n < 0");
    return check(launch_intra_order(LS(ctx), d_satd35, d_mpm, n, lambda_q16, d_order, d_count, d_slot, d_total), "intra_order");
}

int havoc_mi355x_intra_expand(havoc_mi355x_ctx *ctx, const havoc_mi355x_intra_search_job *d_parts, const int32_t *d_order, const int32_t *d_count, const int32_t *d_slot,
                              const int32_t *d_ctx_index, int n, int log2TrafoSize, int quant_scale, int quant_shift, int inv_scale, int lambda_q16, int sdh_factor, int sdh,
                              havoc_mi355x_intra_job *d_intra_jobs, havoc_mi355x_tu_fused_job *d_tu_jobs, havoc_mi355x_rdoq_job *d_rdoq_jobs, int32_t *d_stat_jobs,
                              int32_t *d_owner)
{
    REQUIRE_CTX(); REQUIRE(n >= 0, "n < 0"); REQUIRE(log2TrafoSize >= 2 && log2TrafoSize <= 5, "log2TrafoSize must be 2..5");
    return check(launch_intra_expand(LS(ctx), d_parts, d_order, d_count, d_slot, d_ctx_index, n, log2TrafoSize, quant_scale, quant_shift, inv_scale, lambda_q16, sdh_factor, sdh,
                                     d_intra_jobs, d_tu_jobs, d_rdoq_jobs, d_stat_jobs, d_owner),
                 "intra_expand");
}

int havoc_mi355x_intra_decide(havoc_mi355x_ctx *ctx, const havoc_mi355x_intra_mpm *d_mpm, const int32_t *d_order, const int32_t *d_count, const int32_t *d_slot,
                              const int32_t *d_cbf, const uint32_t *d_ssd, const int32_t *d_stats, const havoc_mi355x_tu_fused_job *d_tu_jobs, int n, int log2TrafoSize,
                              int32_t reciprocal_lambda_q16, havoc_mi355x_intra_choice *d_out, havoc_mi355x_tu_fused_job *d_final)
{
    REQUIRE_CTX(); REQUIRE(n >= 0, "n < 0"); REQUIRE(log2TrafoSize >= 2 && log2TrafoSize <= 5, "log2TrafoSize must be 2..5");
    return check(launch_intra_decide(LS(ctx), d_mpm, d_order, d_count, d_slot, d_cbf, d_ssd, d_stats, d_tu_jobs, n, log2TrafoSize, reciprocal_lambda_q16, d_out, d_final),
                 "intra_decide");
}

int havoc_mi355x_quantize(havoc_mi355x_ctx *ctx, int16_t *d_dst, const int16_t *d_src, const havoc_mi355x_quant_job *d_jobs, int njobs, int32_t *d_cbf)
{
    REQUIRE_CTX(); REQUIRE(njobs >= 0, "njobs < 0");
    return check(launch_quantize(LS(ctx),d_dst, d_src, d_jobs, njobs, d_cbf), "quantize");
}

int havoc_mi355x_quantize_inverse(havoc_mi355x_ctx *ctx, int16_t *d_dst, const int16_t *d_src, const havoc_mi355x_quant_job *d_jobs, int njobs)
{
    REQUIRE_CTX(); REQUIRE(njobs >= 0, "njobs < 0");
    return check(launch_quantize_inverse(LS(ctx),d_dst, d_src, d_jobs, njobs), "quantize_inverse");
}

int havoc_mi355x_quantize_reconstruct(havoc_mi355x_ctx *ctx, int log2TrafoSize, uint8_t *d_rec, intptr_t stride_rec, const uint8_t *d_pred,
                                      intptr_t stride_pred, const int16_t *d_res, const havoc_mi355x_tu_job *d_jobs, int njobs)
{
    REQUIRE_CTX(); REQUIRE(log2TrafoSize >= 2 && log2TrafoSize <= 5, "log2TrafoSize must be 2..5"); REQUIRE(njobs >= 0, "njobs < 0");
    return check(launch_quantize_reconstruct(LS(ctx),log2TrafoSize, d_rec, stride_rec, d_pred, stride_pred, d_res, d_jobs, njobs),
                 "quantize_reconstruct");
}

// the two integers turing/Rdoq.h:163-167 derives from the floating-point lambda: FixedPoint<int32_t, 16>::set(double)
// (turing/FixedPoint.h:47-50) and m_shdRdFactor
void havoc_mi355x_rdoq_lambda(double lambda, int inv_scale, int32_t *lambda_q16, int32_t *sdh_factor)
{
    if (lambda_q16) *lambda_q16 = static_cast<int32_t>(lambda * (1 << 16) + 0.5);
    if (sdh_factor) *sdh_factor = (int)(inv_scale * inv_scale / lambda / 16 + 0.5);
}

int havoc_mi355x_sao_stats(havoc_mi355x_ctx *ctx, int S, int bitDepth, const void *d_src, intptr_t stride_src, const void *d_rec, intptr_t stride_rec,
                           const havoc_mi355x_sao_stats_job *d_jobs, int njobs, int64_t *d_out)
{
    REQUIRE_CTX(); REQUIRE(S == 1 || S == 2, "S must be 1 or 2"); REQUIRE_BD(); REQUIRE(njobs >= 0, "njobs < 0");
    return check(launch_sao_stats(LS(ctx), S, bitDepth, d_src, stride_src, d_rec, stride_rec, d_jobs, njobs, d_out), "sao_stats");
}

int havoc_mi355x_sao_band_chroma(havoc_mi355x_ctx *ctx, int S, int bitDepth, const void *d_src, intptr_t stride_src, const void *d_rec, intptr_t stride_rec,
                                 const havoc_mi355x_sao_chroma_job *d_jobs, int njobs, int64_t *d_out)
{
    REQUIRE_CTX(); REQUIRE(S == 1 || S == 2, "S must be 1 or 2"); REQUIRE_BD(); REQUIRE(njobs >= 0, "njobs < 0");
    return check(launch_sao_band_chroma(LS(ctx), S, bitDepth, d_src, stride_src, d_rec, stride_rec, d_jobs, njobs, d_out), "sao_band_chroma");
}

int havoc_mi355x_sao_filter(havoc_mi355x_ctx *ctx, int S, int bitDepth, void *d_dst, intptr_t stride_dst, const void *d_src, intptr_t stride_src,
                            const havoc_mi355x_sao_job *d_jobs, int njobs)
{
    REQUIRE_CTX(); REQUIRE(S == 1 || S == 2, "S must be 1 or 2"); REQUIRE_BD(); REQUIRE(njobs >= 0, "njobs < 0");
    REQUIRE(d_dst != d_src, "sao_filter: the filtered picture and the deblocked picture must be different buffers");
    return check(launch_sao_filter(LS(ctx), S, bitDepth, d_dst, stride_dst, d_src, stride_src, d_jobs, njobs), "sao_filter");
}

int havoc_mi355x_tu_forward_scan(havoc_mi355x_ctx *ctx, int S, int bitDepth, int log2TrafoSize, int16_t *d_coeffs, const void *d_src, intptr_t stride_src,
                                 const void *d_pred, intptr_t stride_pred, const havoc_mi355x_tu_fused_job *d_jobs, int njobs, const havoc_mi355x_rdoq_job *d_rdoq_jobs,
                                 int16_t *d_levels, void *d_work, size_t work_bytes)
{
    REQUIRE_CTX(); REQUIRE_S(); REQUIRE_BD(); REQUIRE(njobs >= 0, "njobs < 0");
    REQUIRE(log2TrafoSize == 4 || log2TrafoSize == 5, "tu_forward_scan: 16x16 and 32x32 blocks (smaller ones are scanned inside the RDOQ walk)");
    REQUIRE(njobs == 0 || (d_rdoq_jobs && d_levels && d_work && work_bytes >= rdoq_workspace_bytes(njobs) && (reinterpret_cast<uintptr_t>(d_work) & 15) == 0),
            "tu_forward_scan: rdoq jobs / level buffer / workspace missing, misaligned or smaller than havoc_mi355x_rdoq_workspace(njobs)");
    return check(launch_tu_forward_scan(LS(ctx), S, bitDepth, log2TrafoSize, d_coeffs, d_src, stride_src, d_pred, stride_pred, d_jobs, njobs, d_rdoq_jobs, d_levels, d_work),
                 "tu_forward_scan");
}

int havoc_mi355x_rdoq_prescanned(havoc_mi355x_ctx *ctx, int bitDepth, int log2TrafoSize, int16_t *d_dst, const int16_t *d_src, const uint8_t *d_states,
                                 const havoc_mi355x_rdoq_job *d_jobs, int njobs, int32_t *d_cbf, void *d_work, size_t work_bytes)
{
    REQUIRE_CTX(); REQUIRE(log2TrafoSize == 4 || log2TrafoSize == 5, "rdoq_prescanned: 16x16 and 32x32 blocks"); REQUIRE(njobs >= 0, "njobs < 0");
    REQUIRE(bitDepth >= 8 && bitDepth <= 12, "bitDepth must be 8..12");
    REQUIRE(d_dst != d_src, "rdoq: d_dst and d_src must be different buffers");
    REQUIRE(njobs == 0 || (d_work && work_bytes >= rdoq_workspace_bytes(njobs) && (reinterpret_cast<uintptr_t>(d_work) & 15) == 0),
            "rdoq: workspace missing, misaligned or smaller than havoc_mi355x_rdoq_workspace(njobs)");
    return check(launch_rdoq_prescanned(LS(ctx), bitDepth, log2TrafoSize, d_dst, d_src, d_states, d_jobs, njobs, d_cbf, d_work), "rdoq_prescanned");
}

size_t havoc_mi355x_rdoq_workspace(int njobs) { return rdoq_workspace_bytes(njobs); }

int havoc_mi355x_rdoq(havoc_mi355x_ctx *ctx, int bitDepth, int log2TrafoSize, int16_t *d_dst, const int16_t *d_src, const uint8_t *d_states,
                      const havoc_mi355x_rdoq_job *d_jobs, int njobs, int32_t *d_cbf, void *d_work, size_t work_bytes)
{
    REQUIRE_CTX(); REQUIRE(log2TrafoSize >= 2 && log2TrafoSize <= 5, "log2TrafoSize must be 2..5"); REQUIRE(njobs >= 0, "njobs < 0");
    REQUIRE(bitDepth >= 8 && bitDepth <= 12, "bitDepth must be 8..12");
    REQUIRE(d_dst != d_src, "rdoq: d_dst and d_src must be different buffers");
    // 8x8 and 4x4 blocks are walked in job order with the scan inside the walk kernel: no workspace is read or written for them
    REQUIRE(njobs == 0 || log2TrafoSize <= 3 || (d_work && work_bytes >= rdoq_workspace_bytes(njobs) && (reinterpret_cast<uintptr_t>(d_work) & 15) == 0),
            "rdoq: workspace missing, misaligned or smaller than havoc_mi355x_rdoq_workspace(njobs)");
    return check(launch_rdoq(LS(ctx), bitDepth, log2TrafoSize, d_dst, d_src, d_states, d_jobs, njobs, d_cbf, d_work), "rdoq");
}

} // extern "C"
