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
#include "../../include/havoc_classic_ext.h"
#include "../../include/havoc_mi355x.h"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <memory>
#include <mutex>
#include <vector>

namespace {

[[noreturn]] void die(const char *what, int rc)
{
    fprintf(stderr, "libhavoc_classic: %s failed (%d): %s -- there is no CPU fallback\n", what, rc, havoc_mi355x_last_error());
    abort();
}

#define CK(call) do { const int rc_ = (call); if (rc_) die(#call, rc_); } while (0)

// ---- registered pictures (havoc_classic_ext.h) -------------------------------------------------------------------------
struct Pic
{
    int id;
    const char *lo, *hi;        // host byte range of the padded plane
    const char *origin;         // host pointer of sample (0, 0)
    intptr_t stride;            // samples
    int w, h, pad, S, bd, role;
    char *d_plane;              // device copy of [lo, hi)
    char *d_phase;              // reference: 16 planes of `pe` samples each (slot 0 = copy of the plane), device
    char *h_phase;              // reference: pinned host mirror of the 16 planes
    long pe;                    // samples per phase plane
    // rectangle of the padded plane the phase planes are valid in (plane coordinates: x in [-pad, w + pad))
    int vx0, vy0, vx1, vy1;
    long first() const { return -((long)pad * stride + pad); }   // sample index of `lo` relative to the origin
};
// an immutable snapshot of the registered pictures.  `serial` is process-wide and never reused: per-thread caches name a snapshot by it,
// not by its address (a freed snapshot's address may come back for another encoder instance's list)
struct PicList : std::vector<std::shared_ptr<Pic>>
{
    uint64_t serial = 0;
};
std::atomic<int> g_nextPicId{1};                     // picture ids are unique in the process, across bindings (per-thread caches compare them)
std::atomic<uint64_t> g_nextSerial{1};

// The process-wide binding behind every havoc_code of this library: reference-counted (havoc_new_code / havoc_delete_code may
// be called any number of times, from any thread; table entries stay callable while at least one code is alive).
struct Binding
{
    int device = 0;
    int refs = 0;
    std::mutex mu;                                   // registration / new / delete
    std::atomic<const PicList *> pics{nullptr};      // immutable snapshots: table calls read without a lock
    std::vector<const PicList *> retired;            // old snapshots, freed with the binding
    havoc_mi355x_ctx *ctx = nullptr;                 // registration work (uploads, interpolation)
    std::atomic<int64_t> stat[12];                   // [0..7]: havoc_classic_stats; [8] / [9]: 35-mode stages measured at a guessed position / guesses that were right;
                                                     // [10]: waits (a call that launched and waited: what a table call's time is made of), [11]: nanoseconds spent in them (HAVOC_CLASSIC_REPORT only)
    std::atomic<int64_t> oneJob[16];                 // one-job launches by entry point (HAVOC_CLASSIC_REPORT): see kOneJobNames
    std::atomic<int64_t> waitsBy[16], launchesBy[16]; // waits / launches by the entry point the calling thread was in (HAVOC_CLASSIC_REPORT)
    // havoc_quantize_inverse of EVERY int16 level for a (scale, shift) pair, made on the device the first time the pair is seen (one launch, 65 536 values) and kept in
    // pinned memory: the de-quantiser is element-wise (havoc/quantize.cpp:37-46), so a call is answered by looking its levels up (round 6; a picture uses a handful of pairs)
    struct DeqTab { int scale, shift; int16_t *tab; DeqTab *next; };
    std::atomic<DeqTab *> deq{nullptr};              // append-only list; built under mu, read without a lock
    int16_t *deqRampH = nullptr, *deqRampD = nullptr;   // the 65 536 levels in index order (pinned), the tables' common input
    uint64_t generation = 0;                         // never reused in the process (a later binding may be allocated where this one was: per-thread caches compare this, not the address)
    Binding() { for (auto &c : stat) c = 0; for (auto &c : oneJob) c = 0; for (auto &c : waitsBy) c = 0; for (auto &c : launchesBy) c = 0; }
};

Binding *g_binding = nullptr;                        // guarded by g_mu for creation / destruction
std::mutex g_mu;
std::atomic<Binding *> g_live{nullptr};              // what table entries see

enum { kSad, kSad4, kSsd, kSatd, kSsdLinear, kPredUni, kPredBi, kSubtractBi, kIntra, kTransform, kInverse, kInverseAdd, kDequant, kQuant, kQuantRec };
thread_local int t_site = 15;                        // the table entry point this thread is in (kSad ...; 15 = none): what a wait / launch is tallied under
struct Site
{
    int before;
    explicit Site(int k) : before(t_site) { t_site = k; }
    ~Site() { t_site = before; }
};
inline void bump(int k, int64_t n = 1)
{
    if (Binding *b = g_live.load(std::memory_order_relaxed))
    {
        b->stat[k].fetch_add(n, std::memory_order_relaxed);
        if (k == 2) b->launchesBy[t_site & 15].fetch_add(n, std::memory_order_relaxed);
        if (k == 10) b->waitsBy[t_site & 15].fetch_add(n, std::memory_order_relaxed);
    }
}
const char *const kOneJobNames[15] = {"sad", "sad4", "ssd", "satd", "ssd_linear", "pred_uni", "pred_bi", "subtract_bi", "intra", "transform", "inverse_transform",
                                      "inverse_transform_add", "quantize_inverse", "quantize", "quantize_reconstruct"};
// a table call that took the one-job launch path
inline void oneJob(int kind)
{
    if (Binding *b = g_live.load(std::memory_order_relaxed))
    {
        b->stat[1].fetch_add(1, std::memory_order_relaxed);
        b->stat[2].fetch_add(1, std::memory_order_relaxed);
        b->launchesBy[t_site & 15].fetch_add(1, std::memory_order_relaxed);
        b->oneJob[kind].fetch_add(1, std::memory_order_relaxed);
    }
}

// picture containing host pointer p (any byte of the padded plane), or null
inline const Pic *findPic(const void *p)
{
    Binding *b = g_live.load(std::memory_order_acquire);
    if (!b) return nullptr;
    const PicList *l = b->pics.load(std::memory_order_acquire);
    if (!l) return nullptr;
    static thread_local const Pic *last[2] = {nullptr, nullptr};
    static thread_local uint64_t lastSerial = 0;
    const char *c = static_cast<const char *>(p);
    if (lastSerial == l->serial)
        for (const Pic *q : last)
            if (q && c >= q->lo && c < q->hi) return q;
    if (lastSerial != l->serial) last[0] = last[1] = nullptr;   // pictures of another snapshot may be gone
    lastSerial = l->serial;
    for (const auto &q : *l)
        if (c >= q->lo && c < q->hi)
        {
            last[1] = last[0];
            last[0] = q.get();
            return q.get();
        }
    return nullptr;
}

// (x, y) of host pointer p in picture coordinates
inline void locate(const Pic *q, const void *p, int *x, int *y)
{
    const long off = (static_cast<const char *>(p) - q->lo) / q->S;
    const long row = off / q->stride;
    *y = int(row) - q->pad;
    *x = int(off - row * q->stride) - q->pad;
}

// every wait of a table call for the device goes through here: waits are what a table call's time is made of (HAVOC_CLASSIC_REPORT)
inline void waitFor(havoc_mi355x_ctx *ctx)
{
    static const bool timed = getenv("HAVOC_CLASSIC_REPORT") != nullptr;
    if (timed)
    {
        timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        CK(havoc_mi355x_sync_spin(ctx));
        clock_gettime(CLOCK_MONOTONIC, &t1);
        bump(11, (t1.tv_sec - t0.tv_sec) * 1000000000ll + (t1.tv_nsec - t0.tv_nsec));
    }
    else
        CK(havoc_mi355x_sync_spin(ctx));
    bump(10);
}

// per-thread device context + staging memory (never torn down explicitly: process exit reclaims it, which avoids
// calling into the HIP runtime from thread_local destructors during shutdown)
//
// Round 6: the staging buffer is PINNED host memory the device addresses directly (havoc_mi355x_host_alloc): a one-job call packs its operands there, launches,
// and waits ONCE -- the kernel reads the operands and writes its results through the host mapping.  (Before, every such call was a pageable hipMemcpy + wait up, the
// launch, and a hipMemcpy + wait down: three round trips for a few hundred bytes.)
struct Stage
{
    havoc_mi355x_ctx *ctx = nullptr;
    char *d = nullptr;                 // device view of the staging buffer
    struct Host                        // host view: what the calls below index as they did the std::vector it replaces
    {
        char *p = nullptr;
        size_t n = 0;
        char &operator[](size_t i) { return p[i]; }
        char *data() { return p; }
        size_t size() const { return n; }
    } h;
    size_t used = 0;
    bool inFlight = false;             // a launch of this call has not been waited for yet

    void begin()
    {
        if (!ctx)
        {
            Binding *b = g_live.load();
            if (!b)
            {
                fprintf(stderr, "libhavoc_classic: table function called without a live havoc_code\n");
                abort();
            }
            CK(havoc_mi355x_create(&ctx, b->device, HAVOC_MI355X_NEW_STREAM));
        }
        used = 0;
    }

    // pinned, device-visible scratch of this thread (job tables the kernels read and results they write: no copies)
    char *hp = nullptr, *dp = nullptr;
    size_t pcap = 0;
    void pinned(size_t bytes)
    {
        if (pcap >= bytes) return;
        if (hp) CK(havoc_mi355x_host_free(ctx, hp));
        void *h_ = nullptr, *d_ = nullptr;
        pcap = bytes;
        CK(havoc_mi355x_host_alloc(ctx, pcap, &h_, &d_));
        hp = static_cast<char *>(h_);
        dp = static_cast<char *>(d_);
    }

    void grow(size_t bytes)            // (between calls nothing of this thread is in flight: every call waits for what it launched)
    {
        wait();
        void *h_ = nullptr, *d_ = nullptr;
        CK(havoc_mi355x_host_alloc(ctx, bytes, &h_, &d_));
        if (h.p)
        {
            memcpy(h_, h.p, used < h.n ? used : h.n);
            CK(havoc_mi355x_host_free(ctx, h.p));
        }
        h.p = static_cast<char *>(h_);
        h.n = bytes;
        d = static_cast<char *>(d_);
    }

    size_t reserve(size_t bytes)
    {
        const size_t o = (used + 63) & ~size_t(63);
        const size_t before = used;
        used = o + bytes;
        if (h.n < used + 64)
        {
            const size_t keep = used;
            used = before;             // what grow() carries over
            grow((keep + 64) * 2 < (size_t(1) << 18) ? (size_t(1) << 18) : (keep + 64) * 2);
            used = keep;
        }
        return o;
    }

    template <typename T>
    size_t pack(const T *p, intptr_t stride, int w, int rows, int pitch)
    {
        const size_t o = reserve(sizeof(T) * size_t(pitch) * rows + 16);
        for (int y = 0; y < rows; ++y) memcpy(&h[o + sizeof(T) * size_t(y) * pitch], p + y * stride, sizeof(T) * w);
        return o;
    }

    // the operands are where the device reads them: nothing to copy; what follows is a launch this call will wait for
    void upload() { inFlight = true; }

    void wait()
    {
        if (!inFlight) return;
        waitFor(ctx);
        inFlight = false;
    }

    void download(size_t, size_t) { wait(); }      // the results were written through the host mapping

    template <typename T>
    void unpack(T *dst, intptr_t stride, int w, int rows, int pitch, size_t off)
    {
        wait();
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

// ---- precompute and serve (havoc_classic_ext.h) -------------------------------------------------------
//
// Per THREAD (the reference calls the tables from every worker thread, each working on its own PU): a few SAD surfaces, the
// memo of the last prediction this thread copied out of the phase planes, and a few tile-SATD sets.  All results live in
// pinned host memory the kernels write directly; a served call is a pointer look-up plus a few comparisons.

constexpr int kSurfR = 64;                              // surface half-width: the star search's window (Search.hpp:2100)
constexpr int kSurfSide = 2 * kSurfR + 1;
constexpr int kSurfaces = 4, kSatdSets = 2;
constexpr int kSubPel = 3, kSubPelSide = 2 * kSubPel + 1, kSubPelCands = kSubPelSide * kSubPelSide;   // quarter-sample positions

struct SrcKey                                           // the source block of a search
{
    const Pic *pic = nullptr;                           // registered input picture (used while the key is being built) ...
    int picId = 0;                                      // ... compared by its never-reused id
    int x = 0, y = 0;                                   // ... and the block's position in it
    const void *ptr = nullptr;                          // or an unregistered block (bi search: the ideal second predictor)
    intptr_t stride = 0;
    bool same(const SrcKey &o) const { return picId == o.picId && x == o.x && y == o.y && ptr == o.ptr && stride == o.stride; }
};

struct Surface
{
    bool valid = false;
    SrcKey src;
    int refId = 0;                                      // Pic::id of the reference (ids are never reused; pointers may be)
    int w = 0, h = 0, cx = 0, cy = 0;                   // candidates (cx + dx, cy + dy), |dx|, |dy| <= kSurfR
    uint64_t stamp = 0;
    int32_t *res = nullptr;                             // pinned: kSurfSide^2 values
    char *blockCopy = nullptr;                          // pinned copy of an unregistered source block (the kernel's operand
                                                        // and the reference for "is this still the same block")
};

struct PredMemo                                         // what this thread last copied out of a reference's phase planes
{
    bool valid = false;
    const void *dst = nullptr;
    intptr_t sd = 0;
    const Pic *ref = nullptr;
    int x = 0, y = 0, xf = 0, yf = 0, w = 0, h = 0;     // block = plane[4*yf+xf] at (x, y)
};

struct SatdSet
{
    bool valid = false;
    SrcKey src;                                         // the PU's source block (origin of the PU)
    int refId = 0;
    int w = 0, h = 0, n = 0;                            // PU size, tile size
    int cqx = 0, cqy = 0;                               // centre of the 7 x 7 quarter-sample positions (absolute: 4 * x + xFrac)
    uint64_t stamp = 0;
    int32_t *res = nullptr;                             // pinned: [cand][tile]; -1 = position outside the phase planes
    char *blockCopy = nullptr;
};

// ---- intra (round 5; turing/Search.hpp:113-142 -> Reconstruct.cpp:630-701, and the RD candidates' chain Reconstruct.cpp:230-353) ----
// A partition's 35 predictIntraLuma calls read the same 4n + 1 reference samples (two arrays: unfiltered / filtered) and measure against the same source
// block; its RD candidates predict from them again and run transform -> [RDOQ on the host] -> de-quantise -> inverse transform + add -> SSD.  Keyed on the
// CONTENT of the array a call names: the first call predicts every mode from that array in one launch (35 modes + the edge-filtered forms of DC / 10 / 26);
// the first SATD call of the partition measures every mode's tiles against the source block AND makes every mode's forward transform in one wait; the
// de-quantiser call of a candidate also reconstructs it and takes its SSD (prediction and source are known on the device).  Every later call is a look-up
// VALIDATED by content (the residual the transform is given, the coefficients and prediction the inverse transform is given, the reconstruction the SSD
// is given): whatever the caller does differently simply misses and takes the one-job path.
constexpr int kIntraSets = 4, kIntraSlots = 38;

struct IntraSet
{
    bool valid = false;
    int log2 = 0, bd = 0, S = 0, nslots = 0;
    uint64_t stamp = 0;
    char *nb = nullptr;                                 // pinned: the 4n + 1 samples the predictions were made from (the key)
    char *pred = nullptr;                               // pinned: slot k = mode k (k < 35), 35 / 36 / 37 = modes 1 / 10 / 26 with the edge filter; n x n each
    bool measured = false;                              // the 35-mode stage ran for this source block:
    const char *srcHost = nullptr;                      // ... its host address and stride (the caller's picture)
    intptr_t srcStride = 0;
    int srcPicId = 0;
    long srcOff = 0;                                    // ... and its sample offset in the picture's device plane
    const char *srcDev = nullptr;
    int32_t *satd = nullptr;                            // pinned [slot][tile]
    int16_t *coef = nullptr;                            // pinned [slot][n * n]: forward transform of (source - prediction); 4x4: DST-VII (luma, Reconstruct.cpp:263)
    int16_t *coefDct = nullptr;                         // pinned [slot][16]: 4x4 sets only -- the DCT of the same residuals (a 4x4 CHROMA block; the set cannot tell)
    // round 6: what a candidate whose quantised levels are ALL ZERO reconstructs to (two thirds of the reference encoder's de-quantiser calls carry no level at QP 32,
    // profiles/r04_reference_call_mix_1080p.json): havoc_mi355x_tu_reconstruct of every mode on a zero level block, made with the 35-mode stage
    char *rec0 = nullptr;                               // pinned [slot][n * n]
    uint32_t *ssd0 = nullptr;                           // pinned [slot]: havoc_ssd(source, rec0)
    // round 6: a set whose partition never met a SATD call (the Cb / Cr candidates of a unit, Reconstruct.cpp:244-353: predict -> residual -> transform) is measured at
    // its first `transform` call against the source block that call implies -- residual + prediction, exact: the encoder subtracted them -- kept here
    bool ownSource = false;
    char *srcCopy = nullptr;                            // pinned: n x n
    const void *nbPtr = nullptr;                        // where the encoder held the reference samples the set was made from (its per-thread arrays: luma's and chroma's differ)
    bool guessed = false;                               // measured at a GUESSED position (the partition after the last one of this size): valid only if the first SATD call names it
};

// where this thread's last partition of each size was measured: the next one of that size is looked for at its successor in coding order
struct LastPartition
{
    bool valid = false;
    int picId = 0, x = 0, y = 0;
    const void *nbPtr = nullptr;      // the reference-sample array of the set measured there last ...
    int sets = 0;                     // ... and how many sets have been: a luma partition above 4x4 is predicted from TWO arrays (unfiltered / filtered), each a set of its own
};

struct IntraMemo                                        // what this thread last wrote as an intra prediction
{
    bool valid = false;
    const void *dst = nullptr;
    intptr_t sd = 0;
    IntraSet *set = nullptr;
    int slot = 0;
    int tr = 0;                                         // transform type of the candidate: the luma rule until a `transform` call of the candidate says otherwise
};

struct ChainMemo                                        // what the de-quantiser call of an intra candidate computed ahead
{
    bool valid = false;
    IntraSet *set = nullptr;
    int slot = 0, tr = 0;
    int16_t *deq = nullptr;                             // pinned: the de-quantised coefficients it returned
    char *rec = nullptr;                                // pinned: prediction + inverse transform of them, n x n
    uint32_t *ssd = nullptr;                            // pinned: havoc_ssd(source, rec)
    const char *recAt = nullptr;                        // what is served: `rec` / `ssd`, or the set's zero-level reconstruction of the slot
    const uint32_t *ssdAt = nullptr;
    const void *recDst = nullptr;                       // where inverse_transform_add was asked to put it (then an SSD call may follow)
    intptr_t recSd = 0;
};

// the last inter prediction this thread made through a one-job launch (bi-prediction, chroma, unregistered references): its PU-SATD is asked for tile by
// tile right after (Measure.h:97-135) -- the first tile call measures every tile of the block in one launch
struct LastPred
{
    bool valid = false, measured = false;
    const void *dst = nullptr;
    intptr_t sd = 0;
    int w = 0, h = 0, S = 0, n = 0;
    const void *srcBlock = nullptr;                     // the source block the tiles were measured against
    intptr_t srcStride = 0;
    std::vector<char> copy;                             // the prediction as it was returned
    std::vector<int32_t> tiles;
};

// the prediction of the last inverse_transform_add that took the launch path: an inter block's SSD(source, reconstruction) is followed by SSD(source, prediction)
// (Reconstruct.cpp:849-856) -- both in the first one's launch
struct SsdPair
{
    bool havePred = false, valid = false;
    int n = 0, S = 0;
    const void *pa = nullptr;
    intptr_t sa = 0;
    uint32_t value = 0;
    std::vector<char> pred, src;
};

// An inter block's chain (Reconstruct.cpp:766-856): transform(residual) -> [RDOQ] -> de-quantise -> inverse_transform_add(pred) -> SSD(source, rec) -> SSD(source, pred).
// The residual the transform was given is source - prediction, so when the inverse transform arrives with the prediction the source block is residual + prediction:
// both SSDs are measured in the inverse transform's wait, and served when the blocks the SSD calls name hold exactly those samples.
struct InterAhead
{
    bool haveRes = false, valid = false;
    int resN = 0, n = 0, S = 0;
    std::vector<int16_t> res;
    std::vector<char> src, rec, pred;
    uint32_t ssdRec = 0, ssdPred = 0;
};

// Round 6 -- the forward transforms of an inter unit in ONE wait.  reconstructInter subtracts the prediction from the whole unit, all three components, BEFORE it walks
// the transform tree (Reconstruct.cpp:1246-1285), so when the first block's `transform` call arrives the residuals of the blocks that follow (the other luma blocks of a
// 64 x 64 unit, Cb, Cr) are already in the encoder's residual buffer -- a member of its per-thread state, at the same addresses unit after unit.  The library LEARNS, per
// calling thread, which blocks followed a first block (same pointer, stride, size, type) the last time, reads those blocks when the first block comes again, transforms
// them all in the first block's launch, and answers the calls that follow when the residual they name holds EXACTLY the samples that were transformed (compared on the
// host, like every served answer); anything else takes the one-job path and is learnt for the next unit.  A run ends at the first call that is not part of a unit's
// transform chain (a prediction, a SAD / SATD: the encoder is at another candidate).  The blocks read ahead were operands of earlier calls of this thread in the same
// place -- the one assumption made: a residual buffer the encoder passed to `transform` stays readable while it keeps encoding.
constexpr int kFwdFollowers = 12, kFwdHeads = 8;
struct FwdKey
{
    const int16_t *ptr = nullptr;
    intptr_t stride = 0;
    int log2 = 0, tr = 0, bd = 0;
    bool same(const FwdKey &o) const { return ptr == o.ptr && stride == o.stride && log2 == o.log2 && tr == o.tr && bd == o.bd; }
};
struct FwdHead
{
    bool valid = false;
    FwdKey key;
    int nf = 0;
    FwdKey f[kFwdFollowers];
    uint64_t stamp = 0;
};
struct FwdSpec
{
    bool valid = false;
    FwdKey key;
    int16_t *res = nullptr, *coef = nullptr;            // pinned: the residual block as it was read (n x n, contiguous), its coefficients
};
struct FwdState
{
    bool groupOpen = false;
    int cur = -1;                                       // the head whose followers are being recorded
    FwdHead heads[kFwdHeads];
    FwdSpec spec[kFwdFollowers];
    int16_t *headRes = nullptr, *headCoef = nullptr;    // pinned: the first block itself
    char *jobsH = nullptr;                              // pinned: tu jobs of the launches
};

// Round 6.  measurePuCost predicts a PU's luma, Cb and Cr into the reconstructed picture and THEN measures the three SATDs tile by tile against the source picture at the
// same coordinates (Search.hpp:1656-1683, Measure.h:139-163).  The first measured tile of a plane tells where the reconstructed plane's sample (0, 0) lies relative to
// the prediction's address, and which registered source plane it is measured against; from then on a prediction that takes the launch path into that plane measures its
// tiles against the block at the same coordinates in the same wait.  Used only if the first tile call names exactly that source block.
struct PlaneMap
{
    bool valid = false;
    const char *recOrigin = nullptr;                    // address of the destination plane's sample (0, 0) (arithmetic only: never read)
    intptr_t sd = 0;
    const char *srcOrigin = nullptr;                    // the registered source plane's sample (0, 0)
    intptr_t ss = 0;
    int S = 0, rows = 0;                                // rows: the source plane's height (a destination address further down belongs to another plane)
    uint64_t stamp = 0;
};
constexpr int kLastPreds = 3, kPlaneMaps = 4;

struct Serve
{
    bool ready = false;
    LastPred last[kLastPreds];                          // luma, Cb, Cr of a PU are predicted before any of them is measured
    int lastAt = 0;
    PlaneMap maps[kPlaneMaps];
    SsdPair pair;
    InterAhead inter;
    Surface surf[kSurfaces];
    SatdSet sets[kSatdSets];
    PredMemo memo;
    IntraSet intra[kIntraSets];
    IntraMemo imemo;
    ChainMemo chain;
    FwdState fwd;
    LastPartition lastPart[4];                          // by log2 size - 2
    const void *noGuess[8] = {};                        // reference-sample arrays whose sets ended up measured from a residual (chroma): no position is guessed for them
    char *zeroLevels = nullptr;                         // pinned: a 32 x 32 block of zero levels
    uint64_t clock = 0;
    char *jobsH = nullptr, *jobsD = nullptr;            // pinned job tables
    int32_t *denseH = nullptr;                          // pinned: results of a tile-SATD batch in job order
    char *baseH = nullptr, *baseD = nullptr;            // the one pinned arena (host / device view)
    static constexpr size_t kSurfBytes = size_t(kSurfSide) * kSurfSide * 4;
    static constexpr size_t kBlockBytes = 64 * 64 * 2;
    static constexpr size_t kSetBytes = size_t(kSubPelCands) * 256 * 4;            // <= 256 tiles of 4x4 in a 64x64 PU
    static constexpr size_t kJobBytes = size_t(kSubPelCands) * 256 * 16 + 64;
    void init(Stage &s)
    {
        if (ready) return;
        constexpr size_t kIntraPred = size_t(kIntraSlots) * 32 * 32 * 2, kIntraSatd = size_t(kIntraSlots) * 16 * 4, kIntraNb = 512;
        const size_t total = kSurfaces * (kSurfBytes + kBlockBytes) + kSatdSets * (kSetBytes + kBlockBytes) + kJobBytes + kSetBytes + 8192 +
                             kIntraSets * (3 * kIntraPred + kIntraSatd + kIntraNb + 4096 + 32 * 32 * 2) + 4 * 32 * 32 * 2 + 4096 + (kFwdFollowers + 1) * 2 * (32 * 32 * 2 + 256) + 1024;
        void *h_ = nullptr, *d_ = nullptr;
        CK(havoc_mi355x_host_alloc(s.ctx, total, &h_, &d_));
        baseH = static_cast<char *>(h_);
        baseD = static_cast<char *>(d_);
        size_t at = 0;
        auto take = [&](size_t n) { char *p = baseH + at; at += (n + 255) & ~size_t(255); return p; };
        for (auto &f : surf) { f.res = reinterpret_cast<int32_t *>(take(kSurfBytes)); f.blockCopy = take(kBlockBytes); }
        for (auto &f : sets) { f.res = reinterpret_cast<int32_t *>(take(kSetBytes)); f.blockCopy = take(kBlockBytes); }
        jobsH = take(kJobBytes);
        jobsD = dev(jobsH);
        denseH = reinterpret_cast<int32_t *>(take(kSetBytes));
        for (auto &f : intra)
        {
            f.nb = take(kIntraNb);
            f.pred = take(kIntraPred);
            f.coef = reinterpret_cast<int16_t *>(take(kIntraPred));
            f.satd = reinterpret_cast<int32_t *>(take(kIntraSatd));
            f.coefDct = reinterpret_cast<int16_t *>(take(kIntraSlots * 16 * 2));
            f.srcCopy = take(32 * 32 * 2);
            f.rec0 = take(kIntraPred);
            f.ssd0 = reinterpret_cast<uint32_t *>(take(kIntraSlots * 4));
        }
        zeroLevels = take(32 * 32 * 2);
        memset(zeroLevels, 0, 32 * 32 * 2);
        for (auto &e : fwd.spec)
        {
            e.res = reinterpret_cast<int16_t *>(take(32 * 32 * 2));
            e.coef = reinterpret_cast<int16_t *>(take(32 * 32 * 2));
        }
        fwd.headRes = reinterpret_cast<int16_t *>(take(32 * 32 * 2));
        fwd.headCoef = reinterpret_cast<int16_t *>(take(32 * 32 * 2));
        fwd.jobsH = take((kFwdFollowers + 1) * sizeof(havoc_mi355x_tu_job));
        chain.deq = reinterpret_cast<int16_t *>(take(32 * 32 * 2));
        chain.rec = take(32 * 32 * 2);
        chain.ssd = reinterpret_cast<uint32_t *>(take(64));
        ready = true;
    }
    char *dev(const void *hostPtr) const { return baseD + (static_cast<const char *>(hostPtr) - baseH); }
};

Serve &serve(Stage &s)
{
    static thread_local Serve *v = new Serve();
    v->init(s);
    return *v;
}

template <typename Sample>
bool sameBlock(const Sample *p, intptr_t stride, const char *copy, int w, int h)
{
    for (int y = 0; y < h; ++y)
        if (memcmp(p + y * stride, copy + sizeof(Sample) * size_t(y) * w, sizeof(Sample) * w)) return false;
    return true;
}

template <typename Sample>
SrcKey keyOf(const Sample *src, intptr_t ss)
{
    SrcKey k;
    const Pic *q = findPic(src);
    if (q && q->S == int(sizeof(Sample)) && q->stride == ss)
    {
        k.pic = q;
        k.picId = q->id;
        locate(q, src, &k.x, &k.y);
    }
    else
    {
        k.ptr = src;
        k.stride = ss;
    }
    return k;
}

// ---- SAD: one full-pel surface per (source block, reference) -- Search.hpp:1447-1482, 2060-2336
template <typename Sample>
Surface *surfaceFor(Stage &s, Serve &v, const SrcKey &key, const Sample *src, intptr_t ss, const Pic *ref, int rx, int ry, int w, int h)
{
    for (auto &f : v.surf)
        if (f.valid && f.refId == ref->id && f.w == w && f.h == h && f.src.same(key) && abs(rx - f.cx) <= kSurfR && abs(ry - f.cy) <= kSurfR)
        {
            if (key.ptr && !sameBlock(src, ss, f.blockCopy, w, h)) { f.valid = false; continue; }   // the buffer was rewritten
            f.stamp = ++v.clock;
            return &f;
        }
    // miss: a new surface centred on this candidate, moved inwards where the window would leave the padded plane
    const int margin = 4;   // the kernel reads <= 3 bytes past a candidate row
    const int loX = -ref->pad + kSurfR, hiX = ref->w + ref->pad - w - kSurfR - margin;
    const int loY = -ref->pad + kSurfR, hiY = ref->h + ref->pad - h - kSurfR;
    if (loX > hiX || loY > hiY) return nullptr;         // plane too small for a surface of this size
    const int cx = rx < loX ? loX : (rx > hiX ? hiX : rx), cy = ry < loY ? loY : (ry > hiY ? hiY : ry);
    if (abs(rx - cx) > kSurfR || abs(ry - cy) > kSurfR) return nullptr;
    Surface *f = &v.surf[0];
    for (auto &g : v.surf)
        if (!g.valid) { f = &g; break; }
        else if (g.stamp < f->stamp) f = &g;
    f->valid = false;
    const void *dSrc;
    intptr_t dStride;
    int32_t srcOff;
    if (key.pic)
    {
        dSrc = key.pic->d_plane;
        dStride = key.pic->stride;
        srcOff = int32_t((long)(key.y + key.pic->pad) * key.pic->stride + key.x + key.pic->pad);
    }
    else
    {
        for (int y = 0; y < h; ++y) memcpy(f->blockCopy + sizeof(Sample) * size_t(y) * w, src + y * ss, sizeof(Sample) * w);
        dSrc = v.dev(f->blockCopy);
        dStride = w;
        srcOff = 0;
    }
    havoc_mi355x_surface_job *job = reinterpret_cast<havoc_mi355x_surface_job *>(v.jobsH);
    *job = {srcOff, int32_t((long)(cy + ref->pad) * ref->stride + cx + ref->pad), w, h, 0, {0, 0, 0}};
    CK(havoc_mi355x_sad_surface(s.ctx, sizeof(Sample), kSurfR, (w + 3) & ~3, h, dSrc, dStride, ref->d_plane, ref->stride,
                                reinterpret_cast<const havoc_mi355x_surface_job *>(v.jobsD), 1, reinterpret_cast<int32_t *>(v.dev(f->res))));
    waitFor(s.ctx);
    bump(2);
    bump(3);
    f->valid = true;
    f->src = key;
    f->refId = ref->id;
    f->w = w; f->h = h; f->cx = cx; f->cy = cy;
    f->stamp = ++v.clock;
    return f;
}

// true and *out set when the call can be answered from a surface
template <typename Sample>
bool serveSad(const Sample *src, intptr_t ss, const Sample *const *refs, int nrefs, intptr_t rs, int w, int h, int *out)
{
    if ((w & 3) || w > 64 || h > 64) return false;
    const Pic *ref = findPic(refs[0]);
    if (!ref || ref->role != HAVOC_PICTURE_REFERENCE || ref->S != int(sizeof(Sample)) || ref->stride != rs) return false;
    Stage &s = stage();
    Serve &v = serve(s);
    const SrcKey key = keyOf(src, ss);
    for (int i = 0; i < nrefs; ++i)
    {
        if (i && findPic(refs[i]) != ref) return false;
        int rx, ry;
        locate(ref, refs[i], &rx, &ry);
        Surface *f = surfaceFor(s, v, key, src, ss, ref, rx, ry, w, h);
        if (!f) return false;
        out[i] = f->res[(ry - f->cy + kSurfR) * kSurfSide + (rx - f->cx + kSurfR)];
    }
    bump(0);
    return true;
}

// ---- HavocPredUni: a strided copy out of the mirrored phase planes -- Search.hpp:1976-1979
template <typename Sample>
bool servePredUni(Sample *dst, intptr_t sd, const Sample *refp, intptr_t sr, int w, int h, int xFrac, int yFrac, int bitDepth)
{
    Stage &s = stage();
    Serve &v = serve(s);
    v.memo.valid = false;
    const Pic *ref = findPic(refp);
    if (!ref || ref->role != HAVOC_PICTURE_REFERENCE || !ref->h_phase || ref->S != int(sizeof(Sample)) || ref->stride != sr || ref->bd != bitDepth)
        return false;
    int x, y;
    locate(ref, refp, &x, &y);
    if (x < ref->vx0 || y < ref->vy0 || x + w > ref->vx1 || y + h > ref->vy1) return false;
    const Sample *from = reinterpret_cast<const Sample *>(ref->h_phase) + (long)(4 * yFrac + xFrac) * ref->pe + (long)(y + ref->pad) * ref->stride + x + ref->pad;
    for (int r = 0; r < h; ++r) memcpy(dst + r * sd, from + r * ref->stride, sizeof(Sample) * w);
    v.memo = PredMemo{true, dst, sd, ref, x, y, xFrac, yFrac, w, h};
    bump(0);
    return true;
}

// ---- havoc_hadamard_satd of (source tile, tile of the prediction this thread just made): the tile SATDs of all 49
// quarter-sample positions around the vector, one launch per (PU, list) -- Search.hpp:1963-2061, Measure.h:97-135
template <typename Sample, int N>
bool serveSatd(const Sample *a, intptr_t sa, const Sample *b, intptr_t sb, int *out)
{
    Stage &s = stage();
    Serve &v = serve(s);
    const PredMemo &m = v.memo;
    if (!m.valid || sb != m.sd) return false;
    const long off = b - static_cast<const Sample *>(m.dst);
    if (off < 0) return false;
    const int ty = int(off / m.sd), tx = int(off - (long)ty * m.sd);
    if (tx >= m.w || ty >= m.h || (tx % N) || (ty % N) || (m.w % N) || (m.h % N)) return false;
    const Pic *ref = m.ref;
    // the prediction tile must still be what was copied (the caller owns that buffer)
    const Sample *plane = reinterpret_cast<const Sample *>(ref->h_phase) + (long)(4 * m.yf + m.xf) * ref->pe;
    for (int r = 0; r < N; ++r)
        if (memcmp(b + r * sb, plane + (long)(m.y + ty + r + ref->pad) * ref->stride + m.x + tx + ref->pad, sizeof(Sample) * N)) return false;
    // the PU's source block: a is its tile (tx, ty)
    SrcKey key = keyOf(a - (long)ty * sa - tx, sa);
    if (!key.pic)
    {
        // havoc_hadamard_satd only promises N x N readable samples at `a`.  An unregistered source is taken for tile (tx, ty) of a
        // PU-sized block only when this thread has already been handed that very block as the w x h operand of a SAD call (the
        // ideal second predictor of searchMotionBi: its SAD grid precedes its sub-sample stage, Search.hpp:1585-1650)
        bool known = false;
        for (const auto &g : v.surf)
            known |= g.valid && g.src.ptr == key.ptr && g.src.stride == sa && g.w == m.w && g.h == m.h;
        if (!known) return false;
    }
    const int tilesX = m.w / N, ntiles = tilesX * (m.h / N), tile = (ty / N) * tilesX + tx / N;
    const int qx = 4 * m.x + m.xf, qy = 4 * m.y + m.yf;
    SatdSet *f = nullptr;
    for (auto &g : v.sets)
        if (g.valid && g.refId == ref->id && g.w == m.w && g.h == m.h && g.n == N && g.src.same(key) && abs(qx - g.cqx) <= kSubPel && abs(qy - g.cqy) <= kSubPel)
        {
            if (key.ptr && !sameBlock(static_cast<const Sample *>(key.ptr), sa, g.blockCopy, m.w, m.h)) { g.valid = false; continue; }
            f = &g;
            break;
        }
    if (!f)
    {
        if (size_t(ntiles) > 256) return false;
        f = &v.sets[0];
        for (auto &g : v.sets)
            if (!g.valid) { f = &g; break; }
            else if (g.stamp < f->stamp) f = &g;
        f->valid = false;
        const void *dA;
        intptr_t dStrideA;
        long aOrigin;
        if (key.pic)
        {
            dA = key.pic->d_plane;
            dStrideA = key.pic->stride;
            aOrigin = (long)(key.y + key.pic->pad) * key.pic->stride + key.x + key.pic->pad;
        }
        else
        {
            const Sample *blk = static_cast<const Sample *>(key.ptr);
            for (int y = 0; y < m.h; ++y) memcpy(f->blockCopy + sizeof(Sample) * size_t(y) * m.w, blk + y * sa, sizeof(Sample) * m.w);
            dA = v.dev(f->blockCopy);
            dStrideA = m.w;
            aOrigin = 0;
        }
        havoc_mi355x_pair_job *jobs = reinterpret_cast<havoc_mi355x_pair_job *>(v.jobsH);
        int nj = 0;
        for (int c = 0; c < kSubPelCands; ++c)
        {
            const int cq_x = qx + c % kSubPelSide - kSubPel, cq_y = qy + c / kSubPelSide - kSubPel;
            const int X = cq_x >> 2, Y = cq_y >> 2, xf = cq_x & 3, yf = cq_y & 3;
            const bool ok = X >= ref->vx0 && Y >= ref->vy0 && X + m.w <= ref->vx1 && Y + m.h <= ref->vy1;
            for (int t = 0; t < ntiles; ++t)
            {
                f->res[c * ntiles + t] = -1;
                if (!ok) continue;
                const int px = (t % tilesX) * N, py = (t / tilesX) * N;
                jobs[nj] = {int32_t(aOrigin + (long)py * dStrideA + px),
                            int32_t((long)(4 * yf + xf) * ref->pe + (long)(Y + py + ref->pad) * ref->stride + X + px + ref->pad), N, N};
                f->res[c * ntiles + t] = -2 - nj;   // job nj's result: moved into place after the launch
                ++nj;
            }
        }
        CK(havoc_mi355x_satd(s.ctx, sizeof(Sample), N, N, dA, dStrideA, ref->d_phase, ref->stride,
                             reinterpret_cast<const havoc_mi355x_pair_job *>(v.jobsD), nj, reinterpret_cast<int32_t *>(v.dev(v.denseH))));
        waitFor(s.ctx);
        bump(2);
        bump(4);
        for (int i = 0; i < kSubPelCands * ntiles; ++i)
            if (f->res[i] <= -2) f->res[i] = v.denseH[-2 - f->res[i]];
        f->valid = true;
        f->src = key;
        f->refId = ref->id;
        f->w = m.w; f->h = m.h; f->n = N; f->cqx = qx; f->cqy = qy;
    }
    f->stamp = ++v.clock;
    const int c = (qy - f->cqy + kSubPel) * kSubPelSide + (qx - f->cqx + kSubPel);
    const int32_t val = f->res[c * ntiles + tile];
    if (val < 0) return false;
    *out = val;
    bump(0);
    return true;
}

// The 35-mode stage of a partition whose predictions are in f->pred, against the source block at (x, y) of registered picture q: every mode's tile SATDs, every
// mode's residual + forward transform (luma: DST for 4x4 -- Reconstruct.cpp:263) and every mode's reconstruction from a block of ZERO levels with its SSD -- three
// launches on the thread's stream, not waited for here.  False: the block does not lie in the padded plane.
inline void launchIntraMeasureAt(Stage &s, Serve &v, IntraSet *f, const void *dSrc, intptr_t strideSrc, long so, bool withSatd)
{
    const int n = 1 << f->log2;
    constexpr size_t kTuAt = 16384;
    havoc_mi355x_tu_fused_job *tj = reinterpret_cast<havoc_mi355x_tu_fused_job *>(v.jobsH + kTuAt);
    for (int k = 0; k < f->nslots; ++k) tj[k] = {k * n * n, int32_t(so), k * n * n, k * n * n};
    // ONE launch (havoc_mi355x_intra_measure: the tile SATDs, the forward transforms and the zero-level reconstructions of every mode)
    CK(havoc_mi355x_intra_measure(s.ctx, f->S, f->bd, f->log2, reinterpret_cast<int16_t *>(v.dev(f->coef)), reinterpret_cast<int16_t *>(v.dev(f->coefDct)),
                                  reinterpret_cast<int32_t *>(v.dev(f->satd)), v.dev(f->rec0), reinterpret_cast<uint32_t *>(v.dev(f->ssd0)), dSrc, strideSrc, v.dev(f->pred), n,
                                  reinterpret_cast<const havoc_mi355x_tu_fused_job *>(v.jobsD + kTuAt), f->nslots, withSatd ? 1 : 0));
    bump(2);
    f->measured = true;
    f->srcOff = so;
    f->srcDev = static_cast<const char *>(dSrc);
    f->srcStride = strideSrc;
    if (v.chain.set == f) v.chain.valid = false;
}

inline bool launchIntraMeasure(Stage &s, Serve &v, IntraSet *f, const Pic *q, int x, int y)
{
    const int n = 1 << f->log2;
    if (x < -q->pad || y < -q->pad || x + n > q->w + q->pad || y + n > q->h + q->pad) return false;
    launchIntraMeasureAt(s, v, f, q->d_plane, q->stride, (long)(y + q->pad) * q->stride + x + q->pad, true);
    f->ownSource = false;
    f->srcHost = q->origin + ((long)y * q->stride + x) * q->S;
    f->srcPicId = q->id;
    LastPartition &lp = v.lastPart[f->log2 - 2];
    lp.sets = lp.valid && lp.picId == q->id && lp.x == x && lp.y == y ? lp.sets + 1 : 1;
    lp.valid = true;
    lp.picId = q->id;
    lp.x = x;
    lp.y = y;
    lp.nbPtr = f->nbPtr;
    return true;
}

// A candidate of an unmeasured set arrives at `transform`: the source block is residual + prediction (false if a sum is not a sample: then it is not that); every mode's
// forward transform and zero-level reconstruction against it, one wait -- the unit's other candidates from these neighbours are then answered like luma's
template <int BITDEPTH, int LOG2>
bool measureFromResidual(Stage &s, Serve &v, const int16_t *res, intptr_t stride)
{
    constexpr int n = 1 << LOG2;
    const IntraMemo &m = v.imemo;
    if (!m.valid) return false;
    IntraSet *f = m.set;
    if (!f->valid || f->log2 != LOG2 || f->bd != BITDEPTH) return false;
    const int maxv = (1 << BITDEPTH) - 1;
    for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x)
        {
            const int p = f->S == 1 ? reinterpret_cast<const uint8_t *>(f->pred)[m.slot * n * n + y * n + x] : reinterpret_cast<const uint16_t *>(f->pred)[m.slot * n * n + y * n + x];
            const int o = p + res[y * stride + x];
            if (o < 0 || o > maxv) return false;
            if (f->S == 1) reinterpret_cast<uint8_t *>(f->srcCopy)[y * n + x] = uint8_t(o);
            else reinterpret_cast<uint16_t *>(f->srcCopy)[y * n + x] = uint16_t(o);
        }
    launchIntraMeasureAt(s, v, f, v.dev(f->srcCopy), n, 0, false);
    waitFor(s.ctx);
    f->ownSource = true;
    f->guessed = false;
    f->srcHost = f->srcCopy;
    f->srcPicId = 0;
    {
        bool known = false;
        for (const void *p_ : v.noGuess) known |= p_ == f->nbPtr;
        if (!known)
        {
            for (int k = 7; k > 0; --k) v.noGuess[k] = v.noGuess[k - 1];
            v.noGuess[0] = f->nbPtr;
        }
    }
    return true;
}

// the partition of size n that follows (x, y) in coding order: z-order inside the 64 x 64 CTU, then the CTU to the right (a thread encodes a CTU row)
inline void nextPartition(int n, int *x, int *y)
{
    const int cx = *x & ~63, cy = *y & ~63, per = 64 / n;
    int ix = (*x - cx) / n, iy = (*y - cy) / n, m = 0;
    for (int b = 0; b < 4; ++b) m |= ((ix >> b) & 1) << (2 * b) | ((iy >> b) & 1) << (2 * b + 1);
    ++m;
    if (m >= per * per) { *x = cx + 64; *y = cy; return; }
    ix = iy = 0;
    for (int b = 0; b < 4; ++b) { ix |= ((m >> (2 * b)) & 1) << b; iy |= ((m >> (2 * b + 1)) & 1) << b; }
    *x = cx + ix * n;
    *y = cy + iy * n;
}

// ---- intra prediction: every mode from the array the call names, one launch -- Reconstruct.cpp:244-246, 672-674
template <typename Sample, int BD, int LOG2, bool EDGE>
bool serveIntra(Sample *dst, intptr_t sd, const Sample *neighbours, int mode)
{
    constexpr int n = 1 << LOG2, len = 4 * n + 1;
    if (mode < 0 || mode > 34 || (EDGE && mode != 1 && mode != 10 && mode != 26)) return false;
    Stage &s = stage();
    Serve &v = serve(s);
    v.imemo.valid = false;
    const Sample *arr = neighbours - (2 * n + 1);
    IntraSet *f = nullptr;
    for (auto &g : v.intra)
        if (g.valid && g.log2 == LOG2 && g.bd == BD && g.S == int(sizeof(Sample)) && !memcmp(g.nb, arr, sizeof(Sample) * len)) { f = &g; break; }
    if (!f)
    {
        f = &v.intra[0];
        for (auto &g : v.intra)
            if (!g.valid) { f = &g; break; }
            else if (g.stamp < f->stamp) f = &g;
        f->valid = false;
        if (v.chain.set == f) v.chain.valid = false;
        memcpy(f->nb, arr, sizeof(Sample) * len);
        constexpr size_t kIntraAt = 24576;      // (its own place in the job arena: the 35-mode stage queued behind it below fills the arena's head before this launch has run)
        havoc_mi355x_intra_job *jobs = reinterpret_cast<havoc_mi355x_intra_job *>(v.jobsH + kIntraAt);
        const int nslots = LOG2 < 5 ? kIntraSlots : 35;      // the edge filters exist below 32x32 only (havoc/pred_intra.h:41-48)
        static const int edgeMode[3] = {1, 10, 26};
        for (int k = 0; k < nslots; ++k) jobs[k] = {k * n * n, 2 * n + 1, LOG2, k < 35 ? k : edgeMode[k - 35], k < 35 ? 0 : 1, {0, 0, 0}};
        CK(havoc_mi355x_intra(s.ctx, sizeof(Sample), BD, LOG2, v.dev(f->pred), n, v.dev(f->nb), reinterpret_cast<const havoc_mi355x_intra_job *>(v.jobsD + kIntraAt), nslots));
        bump(2);
        f->valid = true;
        f->log2 = LOG2; f->bd = BD; f->S = sizeof(Sample); f->nslots = nslots;
        f->measured = false;
        f->guessed = false;
        f->ownSource = false;
        f->nbPtr = neighbours;
        bool guessable = true;
        for (const void *p_ : v.noGuess) guessable &= p_ != neighbours;
        // round 6: the 35-mode stage in the SAME wait, against the block this thread's partitions of this size have been walking towards (the source picture is on
        // the device: a wrong guess reads nothing of the caller's and costs three small launches; the first SATD call says whether it was right)
        const LastPartition &lp = v.lastPart[LOG2 - 2];
        if (lp.valid && guessable)
        {
            int gx = lp.x, gy = lp.y;
            // the partition's OTHER reference-sample array (its first one was measured there a moment ago), or the next partition
            if (!(lp.sets == 1 && lp.nbPtr != neighbours)) nextPartition(n, &gx, &gy);
            Binding *b = g_live.load(std::memory_order_acquire);
            const PicList *l = b ? b->pics.load(std::memory_order_acquire) : nullptr;
            const Pic *q = nullptr;
            if (l)
                for (const auto &c : *l)
                    if (c->id == lp.picId) { q = c.get(); break; }
            if (q && q->S == int(sizeof(Sample)) && q->bd == BD && gx < q->w && gy < q->h && launchIntraMeasure(s, v, f, q, gx, gy))
            {
                f->guessed = true;
                bump(8);
            }
        }
        waitFor(s.ctx);
    }
    f->stamp = ++v.clock;
    const int slot = EDGE ? (mode == 1 ? 35 : mode == 10 ? 36 : 37) : mode;
    if (slot >= f->nslots) return false;
    const Sample *from = reinterpret_cast<const Sample *>(f->pred) + slot * n * n;
    for (int y = 0; y < n; ++y) memcpy(dst + y * sd, from + y * n, sizeof(Sample) * n);
    v.imemo = IntraMemo{true, dst, sd, f, slot, LOG2 == 2 ? 1 : 0};
    bump(0);
    return true;
}

// The source block an IntraSet was measured against lives in a REGISTERED picture: every use of what was made from it (SATDs, coefficients, the device plane a
// reconstruction reads) first checks that the picture is still the one it was -- same never-reused id, same device plane.  A picture unregistered and another
// registered at the same host address (a reused buffer) with equal neighbour samples (flat areas; the top-left block, whose neighbours are all 1 << (bd - 1))
// would otherwise be answered from the old picture's samples, and its device plane is freed memory.
inline bool sourceStillRegistered(const IntraSet *f)
{
    if (f->ownSource) return true;      // the source block is the set's own copy
    const Pic *q = f->srcHost ? findPic(f->srcHost) : nullptr;
    return q && q->id == f->srcPicId && q->d_plane == f->srcDev;
}

// ---- havoc_hadamard_satd of (source tile, tile of the intra prediction this thread just made): every mode's tiles in one launch, and every mode's forward
// transform with it -- Reconstruct.cpp:684-701, 258-273
template <typename Sample, int N>
bool serveIntraSatd(const Sample *a, intptr_t sa, const Sample *b, intptr_t sb, int *out)
{
    Stage &s = stage();
    Serve &v = serve(s);
    const IntraMemo &m = v.imemo;
    if (!m.valid || sb != m.sd || m.set->S != int(sizeof(Sample))) return false;
    IntraSet *f = m.set;
    const int n = 1 << f->log2;
    if (N != (f->log2 == 2 ? 4 : 8)) return false;
    const long off = b - static_cast<const Sample *>(m.dst);
    if (off < 0) return false;
    const int ty = int(off / m.sd), tx = int(off - (long)ty * m.sd);
    if (tx >= n || ty >= n || (tx % N) || (ty % N)) return false;
    const Sample *pred = reinterpret_cast<const Sample *>(f->pred) + m.slot * n * n;
    for (int r = 0; r < N; ++r)      // the prediction must still be what was served (the caller owns that buffer)
        if (memcmp(b + r * sb, pred + (ty + r) * n + tx, sizeof(Sample) * N)) return false;
    const Sample *block = a - (long)ty * sa - tx;
    if (!f->measured || f->srcHost != reinterpret_cast<const char *>(block) || f->srcStride != sa || !sourceStillRegistered(f))
    {
        f->measured = false;
        f->guessed = false;
        const Pic *q = findPic(block);
        if (!q || q->S != int(sizeof(Sample)) || q->stride != sa) return false;
        int x, y;
        locate(q, block, &x, &y);
        if (!launchIntraMeasure(s, v, f, q, x, y)) return false;
        waitFor(s.ctx);
        for (auto &p_ : v.noGuess)
            if (p_ == f->nbPtr) p_ = nullptr;      // a SATD call named its partition: luma after all
    }
    else if (f->guessed)
    {
        f->guessed = false;      // a right guess: this partition's 35-mode stage cost no wait of its own
        bump(9);
    }
    const int tilesX = n / N;
    *out = f->satd[m.slot * tilesX * tilesX + (ty / N) * tilesX + tx / N];
    bump(0);
    return true;
}

// ---- the forward transform of an intra candidate: made with the 35-mode stage; served when the residual handed in IS source - prediction
template <int BITDEPTH, int LOG2, int TR>
bool serveForward(int16_t *coeffs, const int16_t *res, intptr_t stride)
{
    constexpr int n = 1 << LOG2;
    Stage &s = stage();
    Serve &v = serve(s);
    IntraMemo &m = v.imemo;
    if (!m.valid) return false;
    const IntraSet *f = m.set;
    if (!f->measured || f->log2 != LOG2 || f->bd != BITDEPTH || (LOG2 != 2 && TR != 0)) return false;
    if (!sourceStillRegistered(f)) return false;      // f->srcHost is read below: only while its picture is the one that was measured
    for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x)
        {
            const int p = f->S == 1 ? reinterpret_cast<const uint8_t *>(f->pred)[m.slot * n * n + y * n + x] : reinterpret_cast<const uint16_t *>(f->pred)[m.slot * n * n + y * n + x];
            const int o = f->S == 1 ? reinterpret_cast<const uint8_t *>(f->srcHost)[y * f->srcStride + x] : reinterpret_cast<const uint16_t *>(f->srcHost)[y * f->srcStride + x];
            if (res[y * stride + x] != int16_t(o - p)) return false;
        }
    memcpy(coeffs, (LOG2 == 2 && TR == 0 ? f->coefDct : f->coef) + m.slot * n * n, sizeof(int16_t) * n * n);
    m.tr = TR;
    bump(0);
    return true;
}

// ---- de-quantise (always a launch: its input comes from the host's RDOQ) -- and, for an intra candidate whose prediction and source block are on the device,
// inverse transform + add + SSD in the same wait (Reconstruct.cpp:314-353)
inline void chainAhead(Stage &s, Serve &v, const char *dLevels, int scale, int shift, int n2)
{
    v.chain.valid = false;
    const IntraMemo &m = v.imemo;
    if (!m.valid || !m.set->measured || !sourceStillRegistered(m.set)) return;      // f->srcDev below: freed with its picture
    IntraSet *f = m.set;
    const int n = 1 << f->log2;
    if (n2 != n * n) return;
    havoc_mi355x_tu_fused_job *tj = reinterpret_cast<havoc_mi355x_tu_fused_job *>(v.jobsH);
    tj[0] = {0, int32_t(f->srcOff), m.slot * n * n, 0};
    const int tr = m.tr;
    CK(havoc_mi355x_tu_reconstruct(s.ctx, f->S, f->bd, tr, f->log2, scale, shift, v.dev(v.chain.rec), n, v.dev(f->pred), n, f->srcDev, f->srcStride,
                                   reinterpret_cast<const int16_t *>(dLevels), reinterpret_cast<const havoc_mi355x_tu_fused_job *>(v.jobsD), 1,
                                   reinterpret_cast<uint32_t *>(v.dev(v.chain.ssd))));
    bump(2);
    v.chain.valid = true;
    v.chain.set = f;
    v.chain.slot = m.slot;
    v.chain.tr = tr;
    v.chain.recDst = nullptr;
    v.chain.recAt = v.chain.rec;
    v.chain.ssdAt = v.chain.ssd;
}

template <typename Sample, int LOG2, int TR>
bool serveInverseAdd(Sample *dst, intptr_t sd, const Sample *pred, intptr_t sp, const int16_t *coeffs, int bitDepth)
{
    constexpr int n = 1 << LOG2;
    Stage &s = stage();
    Serve &v = serve(s);
    ChainMemo &c = v.chain;
    if (!c.valid) return false;
    const IntraSet *f = c.set;
    if (!f->valid || f->log2 != LOG2 || f->S != int(sizeof(Sample)) || f->bd != bitDepth || c.tr != TR) return false;
    if (memcmp(coeffs, c.deq, sizeof(int16_t) * n * n)) return false;
    const Sample *want = reinterpret_cast<const Sample *>(f->pred) + c.slot * n * n;
    for (int y = 0; y < n; ++y)
        if (memcmp(pred + y * sp, want + y * n, sizeof(Sample) * n)) return false;
    const Sample *rec = reinterpret_cast<const Sample *>(c.recAt);
    for (int y = 0; y < n; ++y) memcpy(dst + y * sd, rec + y * n, sizeof(Sample) * n);      // (pred may alias dst: compared above, before this)
    c.recDst = dst;
    c.recSd = sd;
    bump(0);
    return true;
}

template <typename Sample>
bool serveSsd(const Sample *pa, intptr_t sa, const Sample *pb, intptr_t sb, int w, int h, uint32_t *out)
{
    Stage &s = stage();
    Serve &v = serve(s);
    const ChainMemo &c = v.chain;
    if (!c.valid || c.recDst != pb || c.recSd != sb) return false;
    const IntraSet *f = c.set;
    const int n = 1 << f->log2;
    if (w != n || h != n || f->S != int(sizeof(Sample))) return false;
    if (f->ownSource)
    {
        if (!sameBlock(pa, sa, f->srcCopy, n, n)) return false;      // the block the call names holds the samples the value was measured against
    }
    else if (reinterpret_cast<const char *>(pa) != f->srcHost || sa != f->srcStride || !sourceStillRegistered(f))
        return false;
    const Sample *rec = reinterpret_cast<const Sample *>(c.recAt);
    for (int y = 0; y < n; ++y)
        if (memcmp(pb + y * sb, rec + y * n, sizeof(Sample) * n)) return false;
    *out = *c.ssdAt;
    bump(0);
    return true;
}

template <typename Sample>
LastPred &rememberPrediction(const Sample *dst, intptr_t sd, int w, int h)
{
    Serve &v = serve(stage());
    int at = -1;
    for (int k = 0; k < kLastPreds; ++k)
        if (v.last[k].valid && v.last[k].dst == dst && v.last[k].sd == sd) at = k;      // the same place again (the next candidate of a search)
    if (at < 0) at = v.lastAt = (v.lastAt + 1) % kLastPreds;
    LastPred &l = v.last[at];
    l.valid = true;
    l.measured = false;
    l.dst = dst; l.sd = sd; l.w = w; l.h = h; l.S = sizeof(Sample);
    l.copy.resize(sizeof(Sample) * size_t(w) * h);
    for (int y = 0; y < h; ++y) memcpy(&l.copy[sizeof(Sample) * size_t(y) * w], dst + y * sd, sizeof(Sample) * w);
    return l;
}

// every tile of the PU in the first tile's launch; false = not this pattern (the caller takes the one-job path)
template <typename Sample, int N>
bool serveTileSatd(const Sample *a, intptr_t sa, const Sample *b, intptr_t sb, int *out)
{
    Stage &s = stage();
    Serve &v = serve(s);
    LastPred *hit = nullptr;
    int tx = 0, ty = 0;
    for (LastPred &c : v.last)
    {
        if (!c.valid || c.S != int(sizeof(Sample)) || sb != c.sd) continue;
        const long off = b - static_cast<const Sample *>(c.dst);
        if (off < 0) continue;
        const int cy = int(off / c.sd), cx = int(off - (long)cy * c.sd);
        if (cx >= c.w || cy >= c.h || (cx % N) || (cy % N) || (c.w % N) || (c.h % N)) continue;
        hit = &c;
        tx = cx; ty = cy;
        break;
    }
    if (!hit) return false;
    LastPred &l = *hit;
    const Sample *pred = reinterpret_cast<const Sample *>(l.copy.data());
    for (int r = 0; r < N; ++r)
        if (memcmp(b + r * sb, pred + (ty + r) * l.w + tx, sizeof(Sample) * N)) return false;
    const Sample *block = a - (long)ty * sa - tx;
    const int tilesX = l.w / N, ntiles = tilesX * (l.h / N);
    if (!l.measured || l.n != N || l.srcBlock != block || l.srcStride != sa)
    {
        // the whole source block must be the caller's to read: a registered picture says so
        const Pic *q = findPic(block);
        if (!q || q->S != int(sizeof(Sample)) || q->stride != sa || ntiles > 256) return false;
        int x, y;
        locate(q, block, &x, &y);
        if (x < -q->pad || y < -q->pad || x + l.w > q->w + q->pad || y + l.h > q->h + q->pad) return false;
        {   // where this destination plane lies against this source plane (PlaneMap)
            const char *recOrigin = static_cast<const char *>(l.dst) - ((long)y * l.sd + x) * (long)sizeof(Sample);
            int at = 0;
            bool known = false;
            for (int k = 0; k < kPlaneMaps && !known; ++k)
                if (v.maps[k].valid && v.maps[k].recOrigin == recOrigin && v.maps[k].sd == l.sd) { at = k; known = true; }
            if (!known)
                for (int k = 1; k < kPlaneMaps; ++k)
                    if (!v.maps[k].valid || (v.maps[at].valid && v.maps[k].stamp < v.maps[at].stamp)) at = k;
            PlaneMap &m = v.maps[at];
            m.valid = true;
            m.recOrigin = recOrigin; m.sd = l.sd;
            m.srcOrigin = q->lo + ((long)q->pad * q->stride + q->pad) * (long)q->S; m.ss = q->stride;
            m.S = int(sizeof(Sample)); m.rows = q->h;
            m.stamp = ++v.clock;
        }
        if (ntiles < 2) return false;      // (a single tile: the one-job call is the same launch)
        const size_t j = s.reserve(sizeof(havoc_mi355x_pair_job) * ntiles), o = s.reserve(4 * size_t(ntiles));
        const size_t pb = s.pack(pred, l.w, l.w, l.h, l.w);
        havoc_mi355x_pair_job *jobs = s.job<havoc_mi355x_pair_job>(j);
        const long so = (long)(y + q->pad) * q->stride + x + q->pad;
        for (int t = 0; t < ntiles; ++t)
        {
            const int px = (t % tilesX) * N, py = (t / tilesX) * N;
            jobs[t] = {int32_t(so + (long)py * q->stride + px), int32_t(py * l.w + px), N, N};
        }
        s.upload();
        CK(havoc_mi355x_satd(s.ctx, sizeof(Sample), N, N, q->d_plane, q->stride, s.d + pb, l.w, s.djob<havoc_mi355x_pair_job>(j), ntiles, (int32_t *)(s.d + o)));
        s.download(o, 4 * size_t(ntiles));
        l.tiles.assign(reinterpret_cast<int32_t *>(&s.h[o]), reinterpret_cast<int32_t *>(&s.h[o]) + ntiles);
        l.measured = true;
        l.n = N;
        l.srcBlock = block;
        l.srcStride = sa;
        bump(2);
        bump(4);
    }
    *out = l.tiles[(ty / N) * tilesX + tx / N];
    bump(0);
    return true;
}

// Round 6: a prediction that takes the launch path (bi-prediction, chroma, references not registered yet) measures its tiles in the same wait -- against the block the
// last measurement of a prediction of this size was made against (SourceGuess).  planTiles reserves the job table and the results BEFORE the prediction is launched (the
// staging buffer must not move under a launch); launchTiles queues the SATD kernel behind the prediction kernel; keepTiles files the results once the call has waited.
struct TilePlan
{
    bool on = false;
    const Pic *q = nullptr;
    int n = 0, ntiles = 0;
    size_t jobs = 0, out = 0;
    const void *block = nullptr;      // the source block the tiles are measured against
};
template <typename Sample>
inline TilePlan planTiles(Stage &s, Serve &v, const Sample *dst, intptr_t sd, int w, int h)
{
    TilePlan t;
    for (PlaneMap &m : v.maps)
    {
        if (!m.valid || m.S != int(sizeof(Sample)) || m.sd != sd) continue;
        const long off = (reinterpret_cast<const char *>(dst) - m.recOrigin) / (long)sizeof(Sample);
        if (off < 0) continue;
        const long y = off / sd, x = off - y * sd;
        if (y + h > m.rows) continue;
        const Sample *block = reinterpret_cast<const Sample *>(m.srcOrigin) + y * m.ss + x;
        const Pic *q = findPic(block);
        if (!q || q->S != int(sizeof(Sample)) || q->stride != m.ss || q->lo + ((long)q->pad * q->stride + q->pad) * (long)q->S != m.srcOrigin) { m.valid = false; continue; }
        if (x + w > q->w + q->pad || y + h > q->h + q->pad) continue;
        t.n = ((w | h) & 3) ? 2 : ((w | h) & 7) ? 4 : 8;      // Measure.h:97-135
        t.ntiles = (w / t.n) * (h / t.n);
        if (t.ntiles < 1 || t.ntiles > 256) return t;
        t.q = q;
        t.block = block;
        t.jobs = s.reserve(sizeof(havoc_mi355x_pair_job) * t.ntiles);
        t.out = s.reserve(4 * size_t(t.ntiles));
        const long so = (long)(y + q->pad) * q->stride + x + q->pad;
        havoc_mi355x_pair_job *jobs = s.job<havoc_mi355x_pair_job>(t.jobs);
        const int tilesX = w / t.n;
        for (int k = 0; k < t.ntiles; ++k)
        {
            const int px = (k % tilesX) * t.n, py = (k / tilesX) * t.n;
            jobs[k] = {int32_t(so + (long)py * q->stride + px), int32_t(py * w + px), t.n, t.n};
        }
        m.stamp = ++v.clock;
        t.on = true;
        return t;
    }
    return t;
}
template <typename Sample>
inline void launchTiles(Stage &s, const TilePlan &t, size_t pred, int w)      // `pred`: where the prediction kernel just launched writes its w-pitch block
{
    if (!t.on) return;
    CK(havoc_mi355x_satd(s.ctx, sizeof(Sample), t.n, t.n, t.q->d_plane, t.q->stride, s.d + pred, w, s.djob<havoc_mi355x_pair_job>(t.jobs), t.ntiles, (int32_t *)(s.d + t.out)));
    bump(2);
    bump(4);
}
inline void keepTiles(Stage &s, LastPred &l, const TilePlan &t)
{
    if (!t.on) return;
    l.tiles.assign(reinterpret_cast<int32_t *>(&s.h[t.out]), reinterpret_cast<int32_t *>(&s.h[t.out]) + t.ntiles);
    l.measured = true;
    l.n = t.n;
    l.srcBlock = t.block;
    l.srcStride = t.q->stride;
}

// ---- distortion metrics ---------------------------------------------------------------------------------------

// a call that is not part of a unit's transform chain: whatever was read ahead for the unit before it is void
inline void endTransformRun()
{
    static thread_local Serve *mine = nullptr;      // (serve() initialises per call: this is on every prediction / SAD / SATD call's path)
    if (!mine) mine = &serve(stage());
    mine->fwd.groupOpen = false;
}

template <typename Sample>
int sad(const Sample *src, intptr_t ss, const Sample *ref, intptr_t rs, uint32_t rect)
{
    Site site_(kSad);
    endTransformRun();
    const int w = rect >> 8, h = rect & 0xff;
    int served;
    if (serveSad<Sample>(src, ss, &ref, 1, rs, w, h, &served)) return served;
    Stage &s = stage();
    oneJob(kSad);
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
    Site site_(kSad4);
    endTransformRun();
    const int w = rect >> 8, h = rect & 0xff;
    if (serveSad<Sample>(src, ss, ref, 4, rs, w, h, out)) return;
    Stage &s = stage();
    oneJob(kSad4);
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
    Site site_(kSsd);
    uint32_t served;
    if (serveSsd<Sample>(pa, sa, pb, sb, w, h, &served)) return served;
    Stage &s = stage();
    {
        const InterAhead &ia = serve(s).inter;
        if (ia.valid && w == ia.n && h == ia.n && ia.S == int(sizeof(Sample)))
        {
            auto equal = [&](const Sample *q, intptr_t st, const std::vector<char> &want) {
                for (int y = 0; y < h; ++y)
                    if (memcmp(q + y * st, &want[sizeof(Sample) * size_t(y) * w], sizeof(Sample) * w)) return false;
                return true;
            };
            if (equal(pa, sa, ia.src))
            {
                if (equal(pb, sb, ia.rec)) { bump(0); return ia.ssdRec; }
                if (equal(pb, sb, ia.pred)) { bump(0); return ia.ssdPred; }
            }
        }
    }
    SsdPair &pr = serve(s).pair;
    const size_t bytes = sizeof(Sample) * size_t(w) * h;
    if (pr.valid && pr.pa == pa && pr.sa == sa && w == pr.n && h == pr.n && pr.S == int(sizeof(Sample)))
    {   // the second SSD of an inter block: the same source block against the prediction its reconstruction was made from
        bool same = true;
        for (int y = 0; y < h && same; ++y)
            same = !memcmp(pb + y * sb, &pr.pred[sizeof(Sample) * size_t(y) * w], sizeof(Sample) * w) && !memcmp(pa + y * sa, &pr.src[sizeof(Sample) * size_t(y) * w], sizeof(Sample) * w);
        pr.valid = false;
        if (same)
        {
            bump(0);
            return pr.value;
        }
    }
    oneJob(kSsd);
    const bool two = pr.havePred && w == pr.n && h == pr.n && pr.S == int(sizeof(Sample));
    const size_t j = s.reserve(2 * sizeof(havoc_mi355x_pair_job)), o = s.reserve(8);
    const size_t a = s.pack(pa, sa, w, h, w), b = s.pack(pb, sb, w, h, w);
    havoc_mi355x_pair_job *jobs = s.job<havoc_mi355x_pair_job>(j);
    jobs[0] = {0, 0, w, h};
    if (two)
    {
        const size_t c = s.pack(reinterpret_cast<const Sample *>(pr.pred.data()), w, w, h, w);
        jobs = s.job<havoc_mi355x_pair_job>(j);      // (the staging buffer may have moved)
        jobs[1] = {0, int32_t((c - b) / sizeof(Sample)), w, h};
    }
    s.upload();
    CK(havoc_mi355x_ssd(s.ctx, sizeof(Sample), s.d + a, w, s.d + b, w, s.djob<havoc_mi355x_pair_job>(j), two ? 2 : 1, (uint32_t *)(s.d + o)));
    s.download(o, 8);
    pr.havePred = false;
    if (two)
    {
        pr.valid = true;
        pr.pa = pa; pr.sa = sa;
        pr.value = reinterpret_cast<uint32_t *>(&s.h[o])[1];
        pr.src.resize(bytes);
        for (int y = 0; y < h; ++y) memcpy(&pr.src[sizeof(Sample) * size_t(y) * w], pa + y * sa, sizeof(Sample) * w);
    }
    return *reinterpret_cast<uint32_t *>(&s.h[o]);
}

template <typename Sample, int N>
int satd(const Sample *pa, intptr_t sa, const Sample *pb, intptr_t sb)
{
    Site site_(kSatd);
    endTransformRun();
    int served;
    if (serveSatd<Sample, N>(pa, sa, pb, sb, &served)) return served;
    if (serveIntraSatd<Sample, N>(pa, sa, pb, sb, &served)) return served;
    if (serveTileSatd<Sample, N>(pa, sa, pb, sb, &served)) return served;
    Stage &s = stage();
    oneJob(kSatd);
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
    oneJob(kSsdLinear);
    const size_t o = s.reserve(4);
    const size_t pa = s.pack(a, 0, size, 1, size), pb = s.pack(b, 0, size, 1, size);
    s.upload();
    CK(havoc_mi355x_ssd_linear(s.ctx, (const uint8_t *)(s.d + pa), (const uint8_t *)(s.d + pb), size, (int32_t *)(s.d + o)));
    s.download(o, 4);
    return *reinterpret_cast<int32_t *>(&s.h[o]);
}

// ---- inter prediction -----------------------------------------------------------------------------------------

// Stages the (w+taps-1) x (h+taps-1) window the kernel addresses (+3 columns its vector loads may touch), but READS from
// the caller only what the reference reads for this phase (havoc/pred_inter.cpp:113-202): no rows above / below the block
// when yFrac == 0, no columns left / right of it when xFrac == 0 -- a caller's buffer may be exactly that tight.  The
// rest of the window is zero (the kernel multiplies it by the zero taps of the {.., 64, ..} filter).  Returns the byte
// offset and sets *origin to the sample offset of the block's integer position.
template <typename Sample>
size_t packWindow(Stage &s, const Sample *ref, intptr_t sr, int w, int h, int taps, int xFrac, int yFrac, int *pitch, int *origin)
{
    const int above = taps / 2 - 1, ww = w + taps - 1, wh = h + taps - 1;
    *pitch = ww + 3;
    *origin = above * *pitch + above;
    const size_t o = s.reserve(sizeof(Sample) * size_t(*pitch) * wh + 16);
    memset(&s.h[o], 0, sizeof(Sample) * size_t(*pitch) * wh + 16);
    const int r0 = yFrac ? 0 : above, r1 = yFrac ? wh : above + h;     // rows of the window the reference touches
    const int c0 = xFrac ? 0 : above, c1 = xFrac ? ww : above + w;
    for (int r = r0; r < r1; ++r)
        memcpy(&s.h[o + sizeof(Sample) * (size_t(r) * *pitch + c0)], ref + (r - above) * sr + (c0 - above), sizeof(Sample) * (c1 - c0));
    return o;
}

template <typename Sample, int TAPS>
void predUni(Sample *dst, intptr_t sd, const Sample *ref, intptr_t sr, int w, int h, int xFrac, int yFrac, int bitDepth)
{
    Site site_(kPredUni);
    endTransformRun();
    if (TAPS == 8 && servePredUni<Sample>(dst, sd, ref, sr, w, h, xFrac, yFrac, bitDepth)) return;
    Stage &s = stage();
    oneJob(kPredUni);
    const size_t j = s.reserve(sizeof(havoc_mi355x_pred_uni_job));
    int pitch, origin;
    const size_t win = packWindow(s, ref, sr, w, h, TAPS, xFrac, yFrac, &pitch, &origin);
    const size_t out = s.reserve(sizeof(Sample) * size_t(w) * h);
    Serve &v = serve(s);
    const TilePlan tiles = planTiles<Sample>(s, v, dst, sd, w, h);
    *s.job<havoc_mi355x_pred_uni_job>(j) = {0, origin, w, h, xFrac, yFrac, {0, 0}};
    s.upload();
    CK(havoc_mi355x_pred_uni(s.ctx, sizeof(Sample), TAPS, bitDepth, w, h, s.d + out, w, s.d + win, pitch, s.djob<havoc_mi355x_pred_uni_job>(j), 1));
    launchTiles<Sample>(s, tiles, out, w);
    s.unpack(dst, sd, w, h, w, out);
    keepTiles(s, rememberPrediction(dst, sd, w, h), tiles);
}

template <typename Sample, int TAPS>
void predBi(Sample *dst, intptr_t sd, const Sample *ref0, const Sample *ref1, intptr_t sr, int w, int h, int xFrac0, int yFrac0, int xFrac1, int yFrac1,
            int bitDepth)
{
    Site site_(kPredBi);
    endTransformRun();
    Stage &s = stage();
    oneJob(kPredBi);
    const size_t j = s.reserve(sizeof(havoc_mi355x_pred_bi_job));
    int pitch, origin;
    const size_t w0 = packWindow(s, ref0, sr, w, h, TAPS, xFrac0, yFrac0, &pitch, &origin);
    const size_t w1 = packWindow(s, ref1, sr, w, h, TAPS, xFrac1, yFrac1, &pitch, &origin);
    const size_t out = s.reserve(sizeof(Sample) * size_t(w) * h);
    Serve &v = serve(s);
    const TilePlan tiles = planTiles<Sample>(s, v, dst, sd, w, h);
    havoc_mi355x_pred_bi_job job = {0, origin, int32_t((w1 - w0) / sizeof(Sample)) + origin, w, h, xFrac0, yFrac0, xFrac1, yFrac1, {0, 0, 0}};
    *s.job<havoc_mi355x_pred_bi_job>(j) = job;
    s.upload();
    CK(havoc_mi355x_pred_bi(s.ctx, sizeof(Sample), TAPS, bitDepth, w, h, s.d + out, w, s.d + w0, pitch, s.djob<havoc_mi355x_pred_bi_job>(j), 1));
    launchTiles<Sample>(s, tiles, out, w);
    s.unpack(dst, sd, w, h, w, out);
    keepTiles(s, rememberPrediction(dst, sd, w, h), tiles);
}

template <typename Sample>
void subtractBi(Sample *dst, intptr_t sd, const Sample *pred, intptr_t sp, const Sample *src, intptr_t ss, int w, int h, int bitDepth)
{
    Site site_(kSubtractBi);
    Stage &s = stage();
    oneJob(kSubtractBi);
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
    Site site_(kIntra);
    constexpr int n = 1 << LOG2;
    endTransformRun();
    if (serveIntra<Sample, BITDEPTH, LOG2, EDGE>(dst, sd, neighbours, mode)) return;
    Stage &s = stage();
    oneJob(kIntra);
    const size_t j = s.reserve(sizeof(havoc_mi355x_intra_job));
    const size_t nb = s.pack(neighbours - 2 * n - 1, 0, 4 * n + 1, 1, 4 * n + 1);
    const size_t out = s.reserve(sizeof(Sample) * n * n);
    *s.job<havoc_mi355x_intra_job>(j) = {0, 2 * n + 1, LOG2, mode, EDGE ? 1 : 0, {0, 0, 0}};
    s.upload();
    CK(havoc_mi355x_intra(s.ctx, sizeof(Sample), BITDEPTH, LOG2, s.d + out, n, s.d + nb, s.djob<havoc_mi355x_intra_job>(j), 1));
    s.unpack(dst, sd, n, n, n, out);
}

// ---- transforms and quantisation ------------------------------------------------------------------------------

// the de-quantiser's table for (scale, shift): tab[(uint16_t)level] = havoc_quantize_inverse(level), every entry computed by the device kernel
const int16_t *dequantTable(int scale, int shift)
{
    Binding *b = g_live.load(std::memory_order_acquire);
    if (!b) return nullptr;
    static thread_local const Binding::DeqTab *last = nullptr;
    static thread_local uint64_t lastOwner = 0;
    if (lastOwner == b->generation && last && last->scale == scale && last->shift == shift) return last->tab;
    auto find = [&]() -> const Binding::DeqTab * {
        for (const Binding::DeqTab *t = b->deq.load(std::memory_order_acquire); t; t = t->next)
            if (t->scale == scale && t->shift == shift) return t;
        return nullptr;
    };
    const Binding::DeqTab *t = find();
    if (!t)
    {
        std::lock_guard<std::mutex> lock(b->mu);
        t = find();
        if (!t)
        {
            constexpr int kLevels = 65536, kJobs = 64;
            if (!b->deqRampH)
            {
                void *h_ = nullptr, *d_ = nullptr;
                CK(havoc_mi355x_host_alloc(b->ctx, sizeof(int16_t) * kLevels, &h_, &d_));
                b->deqRampH = static_cast<int16_t *>(h_);
                b->deqRampD = static_cast<int16_t *>(d_);
                for (int k = 0; k < kLevels; ++k) b->deqRampH[k] = int16_t(uint16_t(k));
            }
            void *h_ = nullptr, *d_ = nullptr;
            CK(havoc_mi355x_host_alloc(b->ctx, sizeof(int16_t) * kLevels + sizeof(havoc_mi355x_quant_job) * kJobs, &h_, &d_));
            havoc_mi355x_quant_job *jobs = reinterpret_cast<havoc_mi355x_quant_job *>(static_cast<char *>(h_) + sizeof(int16_t) * kLevels);
            for (int k = 0; k < kJobs; ++k) jobs[k] = {k * (kLevels / kJobs), k * (kLevels / kJobs), kLevels / kJobs, scale, shift, 0, {0, 0}};
            CK(havoc_mi355x_quantize_inverse(b->ctx, static_cast<int16_t *>(d_), b->deqRampD,
                                             reinterpret_cast<const havoc_mi355x_quant_job *>(static_cast<char *>(d_) + sizeof(int16_t) * kLevels), kJobs));
            CK(havoc_mi355x_sync(b->ctx));
            b->stat[2].fetch_add(1, std::memory_order_relaxed);
            Binding::DeqTab *n = new Binding::DeqTab{scale, shift, static_cast<int16_t *>(h_), b->deq.load(std::memory_order_relaxed)};
            b->deq.store(n, std::memory_order_release);
            t = n;
        }
    }
    last = t;
    lastOwner = b->generation;
    return t->tab;
}

// launches havoc_mi355x_transform for `count` residual blocks of one (size, type) class packed n x n in pinned memory; jobs at v.fwd.jobsH + at
inline void launchForwardClass(Stage &s, Serve &v, int bd, int tr, int log2, int16_t *const *res, int16_t *const *coef, int count, int at)
{
    havoc_mi355x_tu_job *jobs = reinterpret_cast<havoc_mi355x_tu_job *>(v.fwd.jobsH) + at;
    const int n = 1 << log2;
    const int16_t *base = reinterpret_cast<const int16_t *>(v.baseH);      // offsets are int16 indices into the thread's pinned arena
    for (int k = 0; k < count; ++k) jobs[k] = {int32_t(coef[k] - base), int32_t(res[k] - base), 0, 0};
    CK(havoc_mi355x_transform(s.ctx, bd, tr, log2, reinterpret_cast<int16_t *>(v.baseD), reinterpret_cast<const int16_t *>(v.baseD), n,
                              reinterpret_cast<const havoc_mi355x_tu_job *>(v.dev(jobs)), count));
    bump(2);
}

template <int BITDEPTH, int LOG2, int TR>
void forwardTransform(int16_t *coeffs, const int16_t *src, intptr_t stride)
{
    Site site_(kTransform);
    constexpr int n = 1 << LOG2;
    if (serveForward<BITDEPTH, LOG2, TR>(coeffs, src, stride)) return;
    Stage &s = stage();
    Serve &v = serve(s);
    if (v.imemo.valid && (LOG2 == 2 || TR == 0) && (!v.imemo.set->measured || v.imemo.set->ownSource || v.imemo.set->guessed) && measureFromResidual<BITDEPTH, LOG2>(s, v, src, stride) &&
        serveForward<BITDEPTH, LOG2, TR>(coeffs, src, stride))
        return;
    v.imemo.valid = false;      // not the residual of the intra prediction last served: the de-quantiser call that follows is not that candidate's
    {
        InterAhead &ia = v.inter;
        ia.haveRes = true;
        ia.valid = false;
        ia.resN = n;
        ia.res.resize(n * n);
        for (int y = 0; y < n; ++y) memcpy(&ia.res[y * n], src + y * stride, sizeof(int16_t) * n);
    }
    FwdState &fs = v.fwd;
    FwdKey key;
    key.ptr = src; key.stride = stride; key.log2 = LOG2; key.tr = TR; key.bd = BITDEPTH;
    auto record = [&]() {      // this call followed the run's first block: the first block's next occurrence reads it ahead
        if (fs.cur < 0) return;
        FwdHead &h = fs.heads[fs.cur];
        for (int k = 0; k < h.nf; ++k)
            if (h.f[k].same(key)) return;
        if (h.nf < kFwdFollowers && !h.key.same(key)) h.f[h.nf++] = key;
    };
    if (fs.groupOpen)
    {
        for (auto &e : fs.spec)
            if (e.valid && e.key.same(key))
            {
                e.valid = false;
                bool equal = true;
                for (int y = 0; y < n && equal; ++y) equal = !memcmp(src + y * stride, e.res + y * n, sizeof(int16_t) * n);
                if (!equal) break;
                memcpy(coeffs, e.coef, sizeof(int16_t) * n * n);
                record();
                bump(0);
                return;
            }
        record();
    }
    oneJob(kTransform);
    // the blocks to transform in this wait: the one asked for, and -- when it opens a run -- those that followed it the last time
    int16_t *res[kFwdFollowers + 1], *coef[kFwdFollowers + 1];
    FwdKey keys[kFwdFollowers + 1];
    int count = 1;
    keys[0] = key;
    res[0] = fs.headRes;
    coef[0] = fs.headCoef;
    for (int y = 0; y < n; ++y) memcpy(fs.headRes + y * n, src + y * stride, sizeof(int16_t) * n);
    if (!fs.groupOpen)
    {
        for (auto &e : fs.spec) e.valid = false;
        int at = -1, lru = 0;
        for (int k = 0; k < kFwdHeads; ++k)
            if (fs.heads[k].valid && fs.heads[k].key.same(key)) at = k;
            else if (!fs.heads[k].valid || (fs.heads[lru].valid && fs.heads[k].stamp < fs.heads[lru].stamp)) lru = k;
        if (at < 0)
        {
            at = lru;
            fs.heads[at] = FwdHead();
            fs.heads[at].valid = true;
            fs.heads[at].key = key;
        }
        FwdHead &h = fs.heads[at];
        h.stamp = ++v.clock;
        for (int k = 0; k < h.nf; ++k)
        {
            FwdSpec &e = fs.spec[k];
            const int m = 1 << h.f[k].log2;
            for (int y = 0; y < m; ++y) memcpy(e.res + y * m, h.f[k].ptr + y * h.f[k].stride, sizeof(int16_t) * m);
            e.key = h.f[k];
            e.valid = true;
            keys[count] = h.f[k];
            res[count] = e.res;
            coef[count] = e.coef;
            ++count;
        }
        h.nf = 0;      // recorded again by the calls that follow
        fs.cur = at;
        fs.groupOpen = true;
    }
    // one launch per (size, type) class among them
    bool done[kFwdFollowers + 1] = {};
    int at = 0;
    for (int k = 0; k < count; ++k)
    {
        if (done[k]) continue;
        int16_t *r[kFwdFollowers + 1], *c[kFwdFollowers + 1];
        int m = 0;
        for (int q = k; q < count; ++q)
            if (!done[q] && keys[q].log2 == keys[k].log2 && keys[q].tr == keys[k].tr && keys[q].bd == keys[k].bd)
            {
                done[q] = true;
                r[m] = res[q];
                c[m] = coef[q];
                ++m;
            }
        launchForwardClass(s, v, keys[k].bd, keys[k].tr, keys[k].log2, r, c, m, at);
        at += m;
    }
    waitFor(s.ctx);
    memcpy(coeffs, fs.headCoef, sizeof(int16_t) * n * n);
}

template <int LOG2, int TR>
void inverseTransform(int16_t dst[], int16_t const coeffs[], int bitDepth)
{
    constexpr int n = 1 << LOG2;
    Stage &s = stage();
    oneJob(kInverse);
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
    Site site_(kInverseAdd);
    constexpr int n = 1 << LOG2;
    if (serveInverseAdd<Sample, LOG2, TR>(dst, sd, pred, sp, coeffs, bitDepth)) return;
    Stage &s = stage();
    oneJob(kInverseAdd);
    {   // (before dst is written: pred may alias it) the prediction, for the SSD(source, prediction) that follows SSD(source, reconstruction)
        SsdPair &pr = serve(s).pair;
        pr.havePred = true;
        pr.valid = false;
        pr.n = n; pr.S = sizeof(Sample);
        pr.pred.resize(sizeof(Sample) * n * n);
        for (int y = 0; y < n; ++y) memcpy(&pr.pred[sizeof(Sample) * size_t(y) * n], pred + y * sp, sizeof(Sample) * n);
    }
    InterAhead &ia = serve(s).inter;
    ia.valid = false;
    bool ahead = ia.haveRes && ia.resN == n;
    if (ahead)
    {   // the source block this residual was taken against: residual + prediction (no clipping happened if every sum is a sample)
        ia.src.resize(sizeof(Sample) * n * n);
        Sample *sp_ = reinterpret_cast<Sample *>(ia.src.data());
        const int maxv = (1 << bitDepth) - 1;
        for (int y = 0; y < n && ahead; ++y)
            for (int x = 0; x < n; ++x)
            {
                const int v = int(pred[y * sp + x]) + ia.res[y * n + x];
                if (v < 0 || v > maxv) { ahead = false; break; }
                sp_[y * n + x] = Sample(v);
            }
    }
    ia.haveRes = false;
    const size_t j = s.reserve(sizeof(havoc_mi355x_tu_job)), j2 = s.reserve(2 * sizeof(havoc_mi355x_pair_job)), o2 = s.reserve(8);
    const size_t c = s.pack(coeffs, 0, n * n, 1, n * n);
    const size_t p = s.pack(pred, sp, n, n, n);       // staged before dst is written: pred may alias dst
    const size_t out = s.reserve(sizeof(Sample) * n * n);
    const size_t so = ahead ? s.pack(reinterpret_cast<const Sample *>(ia.src.data()), n, n, n, n) : 0;
    *s.job<havoc_mi355x_tu_job>(j) = {0, 0, 0, 0};
    if (ahead)
    {
        havoc_mi355x_pair_job *pj = s.job<havoc_mi355x_pair_job>(j2);
        pj[0] = {0, int32_t((long)(out - p) / (long)sizeof(Sample)), n, n};      // (source', reconstruction): b offsets are relative to the prediction's staging place
        pj[1] = {0, 0, n, n};                                                     // (source', prediction)
    }
    s.upload();
    CK(havoc_mi355x_inverse_transform_add(s.ctx, sizeof(Sample), bitDepth, TR, LOG2, s.d + out, n, s.d + p, n, (const int16_t *)(s.d + c),
                                          s.djob<havoc_mi355x_tu_job>(j), 1));
    if (ahead)
    {
        CK(havoc_mi355x_ssd(s.ctx, sizeof(Sample), s.d + so, n, s.d + p, n, s.djob<havoc_mi355x_pair_job>(j2), 2, (uint32_t *)(s.d + o2)));
        bump(2);
    }
    if (ahead)
    {
        ia.pred.resize(sizeof(Sample) * n * n);
        for (int y = 0; y < n; ++y) memcpy(&ia.pred[sizeof(Sample) * size_t(y) * n], pred + y * sp, sizeof(Sample) * n);
    }
    s.unpack(dst, sd, n, n, n, out);
    if (ahead)
    {
        s.download(o2, 8);
        ia.ssdRec = reinterpret_cast<uint32_t *>(&s.h[o2])[0];
        ia.ssdPred = reinterpret_cast<uint32_t *>(&s.h[o2])[1];
        ia.rec.assign(&s.h[out], &s.h[out] + sizeof(Sample) * n * n);
        ia.n = n;
        ia.S = sizeof(Sample);
        ia.valid = true;
    }
}

// Round 6.  The de-quantiser is answered from the device-made table of its (scale, shift) pair -- but for an intra candidate with levels (prediction and source on the
// device): its call stays a launch, with the candidate's inverse transform + add + SSD in the same wait (Reconstruct.cpp:314-353).  An intra candidate WITHOUT a level is
// reconstructed already: the 35-mode stage made every mode's zero-level reconstruction and SSD.
void quantizeInverse(int16_t *dst, const int16_t *src, int scale, int shift, int n)
{
    Site site_(kDequant);
    Stage &s = stage();
    Serve &v = serve(s);
    const IntraMemo &m = v.imemo;
    const bool intraCandidate = m.valid && m.set->measured && n == (1 << (2 * m.set->log2)) && sourceStillRegistered(m.set);
    bool anyLevel = false;
    for (int i = 0; i < n && !anyLevel; ++i) anyLevel = src[i] != 0;
    if (!intraCandidate || !anyLevel)
    {
        const int16_t *tab = dequantTable(scale, shift);
        for (int i = 0; i < n; ++i) dst[i] = tab[uint16_t(src[i])];
        v.chain.valid = false;
        if (intraCandidate)
        {
            IntraSet *f = m.set;
            const int nn = 1 << f->log2;
            v.chain.valid = true;
            v.chain.set = f;
            v.chain.slot = m.slot;
            v.chain.tr = m.tr;
            v.chain.recDst = nullptr;
            v.chain.recAt = f->rec0 + size_t(m.slot) * nn * nn * f->S;
            v.chain.ssdAt = f->ssd0 + m.slot;
            memcpy(v.chain.deq, dst, 2 * n);
        }
        bump(0);
        return;
    }
    // an intra candidate with levels: the values from the table like every other call; what is launched and waited for is the candidate's reconstruction and SSD from
    // its LEVELS (havoc_mi355x_tu_reconstruct de-quantises them itself), which the inverse_transform_add and ssd calls that follow are answered from
    const int16_t *tab = dequantTable(scale, shift);
    for (int i = 0; i < n; ++i) dst[i] = tab[uint16_t(src[i])];
    const size_t in = s.pack(src, 0, n, 1, n);
    s.upload();
    chainAhead(s, v, s.d + in, scale, shift, n);
    s.wait();
    if (v.chain.valid) memcpy(v.chain.deq, dst, 2 * n);
    bump(0);
}

int quantize(int16_t *dst, const int16_t *src, int scale, int shift, int offset, int n)
{
    Stage &s = stage();
    oneJob(kQuant);
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
    oneJob(kQuantRec);
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

static void releasePic(Binding *b, Pic *q)
{
    if (q->d_plane) (void)havoc_mi355x_free(b->ctx, q->d_plane);
    if (q->d_phase) (void)havoc_mi355x_free(b->ctx, q->d_phase);
    if (q->h_phase) (void)havoc_mi355x_host_free(b->ctx, q->h_phase);
    q->d_plane = q->d_phase = q->h_phase = nullptr;
}

// A replaced snapshot may still be under the eyes of a table call on another thread (findPic walks it without a lock, for a few hundred
// nanoseconds), so it is not freed at once -- but not kept for the life of the binding either (an encode registers pictures per frame):
// it goes when kRetired newer snapshots have been retired after it.
static void retire(Binding *b, const PicList *old)
{
    constexpr size_t kRetired = 64;
    b->retired.push_back(old);
    while (b->retired.size() > kRetired)
    {
        delete b->retired.front();
        b->retired.erase(b->retired.begin());
    }
}

havoc_code havoc_new_code(havoc_instruction_set mask, int size)
{
    (void)mask;
    (void)size;
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g_binding)
    {
        Binding *b = new Binding();
        static uint64_t generations = 0;      // (under g_mu)
        b->generation = ++generations;
        if (const char *e = getenv("HAVOC_MI355X_DEVICE")) b->device = atoi(e);
        const int rc = havoc_mi355x_create(&b->ctx, b->device, HAVOC_MI355X_NEW_STREAM);
        if (rc) die("havoc_new_code: havoc_mi355x_create", rc);
        g_binding = b;
        g_live.store(b, std::memory_order_release);
    }
    ++g_binding->refs;
    havoc_code code;
    code.implementation = g_binding;
    return code;
}

// The binding lives as long as any havoc_code does.  Table entries populated from a deleted code must not be called any
// more (as with the reference, whose JIT buffer goes away); entries of OTHER live codes keep working.
void havoc_delete_code(havoc_code code)
{
    std::lock_guard<std::mutex> lock(g_mu);
    Binding *b = static_cast<Binding *>(code.implementation);
    if (!b || b != g_binding || b->refs <= 0) return;
    if (--b->refs > 0) return;
    if (getenv("HAVOC_CLASSIC_REPORT"))
        fprintf(stderr, "libhavoc_classic: table calls served %lld, one-job launches %lld, launches %lld, surfaces %lld, tile-SATD batches %lld, pictures %lld\n",
                (long long)b->stat[0], (long long)b->stat[1], (long long)b->stat[2], (long long)b->stat[3], (long long)b->stat[4], (long long)b->stat[5]);
    if (getenv("HAVOC_CLASSIC_REPORT"))
        fprintf(stderr, "libhavoc_classic: waits %lld (%.3f s in them, all threads); 35-mode stages at a guessed position %lld, right %lld\n", (long long)b->stat[10], double(b->stat[11]) * 1e-9, (long long)b->stat[8], (long long)b->stat[9]);
    if (getenv("HAVOC_CLASSIC_REPORT"))
    {
        fprintf(stderr, "libhavoc_classic: one-job launches by entry point:");
        for (int k = 0; k < 15; ++k)
            if (b->oneJob[k].load()) fprintf(stderr, " %s %lld", kOneJobNames[k], (long long)b->oneJob[k].load());
        fprintf(stderr, "\n");
        fprintf(stderr, "libhavoc_classic: waits / launches by entry point:");
        for (int k = 0; k < 16; ++k)
            if (b->waitsBy[k].load() || b->launchesBy[k].load()) fprintf(stderr, " %s %lld/%lld", k < 15 ? kOneJobNames[k] : "other", (long long)b->waitsBy[k].load(), (long long)b->launchesBy[k].load());
        fprintf(stderr, "\n");
    }
    g_live.store(nullptr, std::memory_order_release);
    // device memory of the pictures still registered; per-thread contexts are left to process exit (see Stage)
    if (const PicList *l = b->pics.load())
        for (const auto &q : *l) releasePic(b, q.get());
    for (const PicList *l : b->retired) delete l;
    delete b->pics.load();
    for (Binding::DeqTab *t = b->deq.load(); t;)
    {
        Binding::DeqTab *next = t->next;
        (void)havoc_mi355x_host_free(b->ctx, t->tab);
        delete t;
        t = next;
    }
    if (b->deqRampH) (void)havoc_mi355x_host_free(b->ctx, b->deqRampH);
    havoc_mi355x_destroy(b->ctx);
    delete b;
    g_binding = nullptr;
}

int havoc_classic_register_picture(havoc_code code, const void *origin, intptr_t stride, int width, int height, int pad, int S, int bit_depth, int role)
{
    Binding *b = static_cast<Binding *>(code.implementation);
    if (!b || !origin || (S != 1 && S != 2) || width <= 0 || height <= 0 || pad < 0 || stride < width + 2 * pad) return HAVOC_MI355X_EINVAL;
    if (bit_depth < 8 || bit_depth > (S == 1 ? 8 : 10)) return HAVOC_MI355X_EINVAL;
    std::lock_guard<std::mutex> lock(b->mu);
    auto q = std::make_shared<Pic>();
    q->id = g_nextPicId.fetch_add(1);
    q->origin = static_cast<const char *>(origin);
    q->stride = stride;
    q->w = width; q->h = height; q->pad = pad; q->S = S; q->bd = bit_depth; q->role = role;
    q->lo = q->origin + q->first() * S;
    const size_t elems = size_t(stride) * (height + 2 * pad);
    // the last row of a plane need not own its alignment tail: copy up to the last sample of the padded plane
    const size_t used = elems - size_t(stride - (width + 2 * pad));
    q->hi = q->lo + used * S;
    q->d_plane = q->d_phase = q->h_phase = nullptr;
    q->pe = 0;
    q->vx0 = q->vy0 = q->vx1 = q->vy1 = 0;
    havoc_mi355x_ctx *ctx = b->ctx;
    int rc;
    void *dp = nullptr;
    // whatever a failed registration has allocated goes back before it returns
    struct Undo { Binding *b; Pic *q; bool armed = true; ~Undo() { if (armed) releasePic(b, q); } } undo{b, q.get()};
    if ((rc = havoc_mi355x_malloc(ctx, &dp, elems * S + 256))) return rc;
    q->d_plane = static_cast<char *>(dp);
    if ((rc = havoc_mi355x_h2d(ctx, q->d_plane, q->lo, used * S))) return rc;
    b->stat[6] += int64_t(used * S);
    if (role == HAVOC_PICTURE_REFERENCE && pad >= 16)
    {
        // the 16 fractional-sample planes, valid where the 8-tap window and the kernels' vector loads stay inside the plane
        q->pe = long((elems + 63) & ~size_t(63));
        void *ph = nullptr, *hh = nullptr, *hd = nullptr;
        if ((rc = havoc_mi355x_malloc(ctx, &ph, size_t(q->pe) * 16 * S + 256))) return rc;
        q->d_phase = static_cast<char *>(ph);
        if ((rc = havoc_mi355x_h2d(ctx, q->d_phase, q->lo, used * S))) return rc;          // slot 0 = the picture itself
        const int x0 = 12, y0 = 4, wdt = width + 2 * pad - 24, hgt = height + 2 * pad - 8;
        if ((rc = havoc_mi355x_interp_planes(ctx, S, bit_depth, q->d_phase, q->pe, q->d_plane, stride, x0, y0, wdt, hgt))) return rc;
        if ((rc = havoc_mi355x_host_alloc(ctx, size_t(q->pe) * 16 * S, &hh, &hd))) return rc;
        q->h_phase = static_cast<char *>(hh);
        if ((rc = havoc_mi355x_d2h(ctx, q->h_phase, q->d_phase, size_t(q->pe) * 16 * S))) return rc;
        b->stat[7] += int64_t(size_t(q->pe) * 16 * S);
        q->vx0 = x0 - pad; q->vy0 = y0 - pad; q->vx1 = x0 - pad + wdt; q->vy1 = y0 - pad + hgt;
    }
    else if ((rc = havoc_mi355x_sync(ctx)))
        return rc;
    undo.armed = false;
    const PicList *old = b->pics.load();
    PicList *next = new PicList();
    next->serial = g_nextSerial.fetch_add(1);
    // a plane that overlaps this one in host memory cannot still be a picture (its buffer was freed and reused, or the same picture is registered
    // again with new contents): it goes, so that no pointer can name two pictures
    std::vector<std::shared_ptr<Pic>> evicted;
    if (old)
    {
        for (const auto &o : *old)
        {
            if (o->lo < q->hi && q->lo < o->hi) evicted.push_back(o);
            else next->push_back(o);
        }
    }
    next->push_back(q);
    b->pics.store(next, std::memory_order_release);
    if (old) retire(b, old);
    for (const auto &o : evicted) releasePic(b, o.get());
    b->stat[5] += 1;
    return 0;
}

int havoc_classic_unregister_picture(havoc_code code, const void *origin)
{
    Binding *b = static_cast<Binding *>(code.implementation);
    if (!b || !origin) return HAVOC_MI355X_EINVAL;
    std::lock_guard<std::mutex> lock(b->mu);
    const PicList *old = b->pics.load();
    if (!old) return HAVOC_MI355X_EINVAL;
    PicList *next = new PicList();
    next->serial = g_nextSerial.fetch_add(1);
    std::shared_ptr<Pic> gone;
    for (const auto &q : *old)
        if (q->origin == origin && !gone) gone = q;
        else next->push_back(q);
    if (!gone)
    {
        delete next;
        return HAVOC_MI355X_EINVAL;
    }
    b->pics.store(next, std::memory_order_release);
    retire(b, old);   // the snapshot object itself (a reader on another thread may be walking it) is kept for a while
    // Contract (as for the reference's own picture buffers): unregister a plane only when no table call can still name it.
    // Its device and pinned memory go now; per-thread caches match pictures by id, and ids are never reused.
    releasePic(b, gone.get());
    return 0;
}

void havoc_classic_stats(havoc_code code, int64_t out[8])
{
    Binding *b = static_cast<Binding *>(code.implementation);
    for (int k = 0; k < 8; ++k) out[k] = b ? b->stat[k].load() : 0;
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
