// Private to libhavoc_mi355x.so: the context object behind havoc_mi355x_ctx and the argument / error helpers every
// C-ABI translation unit (api.hip, picture_store.hip) uses.
#pragma once

#include "common.h"

#include <cstdio>
#include <cstring>

struct havoc_mi355x_ctx
{
    int device;
    hipStream_t stream;
    hipEvent_t ev0, ev1;
    hipDeviceProp_t prop;
    bool ownsStream;
    // fork/join lanes: independent launch chains issued on side streams so that they overlap on the GPU
    static constexpr int kMaxLanes = 8;
    hipStream_t lanes[kMaxLanes];
    hipEvent_t laneEv[kMaxLanes];
    hipEvent_t forkEv;
    int nlanes;   // 0 = not forked
    int cur;      // lane the next launch goes to (0 = the context's main stream)
    // havoc_mi355x_sync_spin: a pinned word the stream writes a sequence number to (hipStreamWriteValue32) and the host polls
    volatile uint32_t *flagH = nullptr;
    void *flagD = nullptr;
    uint32_t flagSeq = 0;
    const int32_t *searchGate = nullptr;      // havoc_mi355x_search_gate: rows of the reference pictures that have arrived, [2] device ints (nullptr: references complete)
};

// the stream the next launch is issued on
static inline hipStream_t LS(havoc_mi355x_ctx *ctx) { return ctx->cur == 0 ? ctx->stream : ctx->lanes[ctx->cur]; }

char *havoc_err_buf();   // thread-local, 256 bytes (api.hip)
#define g_err (havoc_err_buf())

static int fail(int code, const char *what)
{
    snprintf(g_err, 256, "%s", what);
    return code;
}

static int check(hipError_t e, const char *where)
{
    if (e == hipSuccess) return 0;
    snprintf(g_err, 256, "%s: %s", where, hipGetErrorString(e));
    return -(int)e;
}

#define REQUIRE(cond, what) \
    do { if (!(cond)) return fail(HAVOC_MI355X_EINVAL, what); } while (0)

// Every entry point runs with the context's device current on the calling thread and puts the caller's device back on
// return (a process may hold contexts on several GPUs; hipMalloc, NULL-stream launches and event calls act on whatever
// device is current).  hipSetDevice is only issued when the current device differs.
struct DeviceGuard
{
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int device)
    {
        if (hipGetDevice(&prev) == hipSuccess && prev != device) switched = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceGuard()
    {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define REQUIRE_CTX() \
    REQUIRE(ctx != nullptr, "null context"); \
    DeviceGuard device_guard_(ctx->device)
