// Sample-adaptive offset primitives (the SAO third):
//
//   k_sao_stats  : the statistics the encoder's SAO decision is made from, turing/EncSao.h:151-283 (edge_offset_stats_class0..3:
//                  per category the number of samples and the sum of original - reconstruction) and :111-148
//                  (band_offset_luma_stats: the same per band of 8 << (bitDepth - 8) values, and the four-band window holding most
//                  samples), over a block WITHOUT its outermost ring of samples.  The horizontal class counts the first interior
//                  sample of every row twice, the second time in category 0 (EncSao.h:166-176): reproduced.
//   k_sao_filter : turing/sao.cpp:33-92 -- sao_filter_band (offset by band) and sao_filter_edge (offset by the sign pattern
//                  against the two neighbours of the edge class); source and destination are different pictures, so every
//                  sample is independent.
// The rate-distortion decision between them (EncSao.h:286-1125, floating point) and the CTU availability rules
// (LoopFilter.h:886-1008) are encoder control and stay on the host, like the deblocking filter's boundary strengths.
//
// One workgroup per job (a CTU of one colour component).  Statistics: a lane takes samples of the interior at stride 256, reads
// the 3x3 neighbourhood through the cache; the 4 x 5 edge categories accumulate in registers (select on the category, wave
// reduction at the end), the 32 bands in per-wavefront LDS counters; the sums fit 32 bits (62 x 62 samples x 16-bit differences).
#include "common.h"

namespace havoc_gpu {

namespace {

struct SaoStatsJob { int32_t src_off, rec_off, w, h; };
struct SaoJob { int32_t dst_off, src_off, w, h, type, eo_class; int16_t offsets[32]; int32_t reserved[2]; };
static_assert(sizeof(SaoStatsJob) == sizeof(havoc_mi355x_sao_stats_job) && sizeof(SaoJob) == sizeof(havoc_mi355x_sao_job) && sizeof(SaoJob) == 96, "sao job layout");

__device__ __forceinline__ int sign3(int v) { return (v > 0) - (v < 0); }

template <int S>
__global__ __launch_bounds__(256) void k_sao_stats(const char *__restrict__ srcPlane, long strideSrc, const char *__restrict__ recPlane, long strideRec,
                                                   const SaoStatsJob *__restrict__ jobs, int shift, long long *__restrict__ out)
{
    typedef typename Sample<S>::T T;
    __shared__ int acc[104];            // the block's totals
    __shared__ int band[4][64];         // per wavefront: sums [0..31], counts [32..63] (fewer lanes fighting over a counter)
    const SaoStatsJob job = jobs[blockIdx.x];
    const T *src = reinterpret_cast<const T *>(srcPlane) + job.src_off, *rec = reinterpret_cast<const T *>(recPlane) + job.rec_off;
    const int tid = threadIdx.x, wave = tid >> 6, iw = job.w - 2, ih = job.h - 2;
    if (tid < 104) acc[tid] = 0;
    band[wave][tid & 63] = 0;
    __syncthreads();
    // edge classes: 4 classes x 5 categories x (sum, count) in registers, selected by comparison -- no memory traffic per sample
    int e[4][5], n[4][5];
#pragma unroll
    for (int cls = 0; cls < 4; ++cls)
#pragma unroll
        for (int k = 0; k < 5; ++k) e[cls][k] = n[cls][k] = 0;
    for (int k = tid; k < iw * ih; k += 256)
    {
        const int y = 1 + k / iw, x = 1 + k - (y - 1) * iw;
        const T *r = rec + y * strideRec + x;
        const int c = r[0], diff = (int)src[y * strideSrc + x] - c;
        const int idx[4] = { 2 + sign3(c - (int)r[-1]) + sign3(c - (int)r[1]), 2 + sign3(c - (int)r[-strideRec]) + sign3(c - (int)r[strideRec]),
                             2 + sign3(c - (int)r[-strideRec - 1]) + sign3(c - (int)r[strideRec + 1]),
                             2 + sign3(c - (int)r[-strideRec + 1]) + sign3(c - (int)r[strideRec - 1]) };
#pragma unroll
        for (int cls = 0; cls < 4; ++cls)
#pragma unroll
            for (int q = 0; q < 5; ++q)      // q = 2 + sign + sign; its category is 1, 2, 0, 3, 4 (applied when the registers are flushed)
            {
                const bool hit = idx[cls] == q;
                e[cls][q] += hit ? diff : 0;
                n[cls][q] += hit;
            }
        const int b = c >> (3 + shift);
        atomicAdd(&band[wave][b], diff);
        atomicAdd(&band[wave][32 + b], 1);
    }
#pragma unroll
    for (int cls = 0; cls < 4; ++cls)
#pragma unroll
        for (int q = 0; q < 5; ++q)
        {
            const int category = (0x43021 >> (4 * q)) & 7;
            const int se = wave_sum(e[cls][q]), sn = wave_sum(n[cls][q]);
            if ((tid & 63) == 0)
            {
                atomicAdd(&acc[10 * cls + category], se);
                atomicAdd(&acc[10 * cls + 5 + category], sn);
            }
        }
    if (iw > 0)
        for (int y = 1 + tid; y <= ih; y += 256)      // the horizontal class's second look at x = 1
        {
            atomicAdd(&acc[0], (int)src[y * strideSrc + 1] - (int)rec[y * strideRec + 1]);
            atomicAdd(&acc[5], 1);
        }
    __syncthreads();
    if (tid < 64) acc[40 + tid] = band[0][tid] + band[1][tid] + band[2][tid] + band[3][tid];
    __syncthreads();
    long long *o = out + 105L * blockIdx.x;
    if (tid < 104) o[tid] = acc[tid];
    if (tid == 0)
    {
        int best = 0, start = 0;
        for (int b = 0; b < 29; ++b)
        {
            const int cum = acc[72 + b] + acc[73 + b] + acc[74 + b] + acc[75 + b];
            if (cum > best)
            {
                best = cum;
                start = b;
            }
        }
        o[104] = max(start + 1, 2);
    }
}

// band_offset_chroma_stats (turing/EncSao.h:62-109): ONE band histogram over the interiors of the Cb and the Cr block of a CTU -- per band
// the number of samples and the sum of original - reconstruction of both planes -- and the four-band window holding most samples.
// job = offsets of the Cb / Cr blocks in the source and reconstruction chroma planes; out[65 * job]: E[32], count[32], band position.
struct SaoChromaJob { int32_t src_u, src_v, rec_u, rec_v, w, h, reserved[2]; };
static_assert(sizeof(SaoChromaJob) == sizeof(havoc_mi355x_sao_chroma_job), "sao chroma job layout");

template <int S>
__global__ __launch_bounds__(256) void k_sao_band_chroma(const char *__restrict__ srcPlane, long strideSrc, const char *__restrict__ recPlane, long strideRec,
                                                         const SaoChromaJob *__restrict__ jobs, int shift, long long *__restrict__ out)
{
    typedef typename Sample<S>::T T;
    __shared__ int band[4][64];         // per wavefront: sums [0..31], counts [32..63]
    __shared__ int total[64];
    const SaoChromaJob job = jobs[blockIdx.x];
    const int tid = threadIdx.x, wave = tid >> 6, iw = job.w - 2, ih = job.h - 2;
    band[wave][tid & 63] = 0;
    __syncthreads();
    for (int plane = 0; plane < 2; ++plane)
    {
        const T *src = reinterpret_cast<const T *>(srcPlane) + (plane ? job.src_v : job.src_u);
        const T *rec = reinterpret_cast<const T *>(recPlane) + (plane ? job.rec_v : job.rec_u);
        for (int k = tid; k < iw * ih; k += 256)
        {
            const int y = 1 + k / iw, x = 1 + k - (y - 1) * iw;
            const int c = rec[y * strideRec + x], b = c >> (3 + shift);
            atomicAdd(&band[wave][b], (int)src[y * strideSrc + x] - c);
            atomicAdd(&band[wave][32 + b], 1);
        }
    }
    __syncthreads();
    if (tid < 64) total[tid] = band[0][tid] + band[1][tid] + band[2][tid] + band[3][tid];
    __syncthreads();
    long long *o = out + 65L * blockIdx.x;
    if (tid < 64) o[tid] = total[tid];
    if (tid == 0)
    {
        int best = 0, start = 0;
        for (int b = 0; b < 29; ++b)
        {
            const int cum = total[32 + b] + total[33 + b] + total[34 + b] + total[35 + b];
            if (cum > best)
            {
                best = cum;
                start = b;
            }
        }
        o[64] = max(start + 1, 2);
    }
}

template <int S>
__global__ __launch_bounds__(256) void k_sao_filter(char *__restrict__ dstPlane, long strideDst, const char *__restrict__ srcPlane, long strideSrc,
                                                    const SaoJob *__restrict__ jobs, int bitDepth)
{
    typedef typename Sample<S>::T T;
    __shared__ SaoJob job;
    if (threadIdx.x < sizeof(SaoJob) / 4) reinterpret_cast<int *>(&job)[threadIdx.x] = reinterpret_cast<const int *>(jobs + blockIdx.x)[threadIdx.x];
    __syncthreads();
    T *dst = reinterpret_cast<T *>(dstPlane) + job.dst_off;
    const T *src = reinterpret_cast<const T *>(srcPlane) + job.src_off;
    const int mx = (1 << bitDepth) - 1, w = job.w, n = job.w * job.h;
    // neighbours of the edge class (sao.cpp:63-73): horizontal, vertical, 135 degrees, 45 degrees
    const int e = job.eo_class & 3, hx = e == 1 ? 0 : (e == 3 ? 1 : -1), vy = e == 0 ? 0 : -1;
    const long n0 = vy * strideSrc + hx;
    for (int k = threadIdx.x; k < n; k += 256)
    {
        const int y = k / w, x = k - y * w;
        const T *p = src + y * strideSrc + x;
        const int c = p[0];
        int v = c;
        if (job.type == 1)
            v = c + job.offsets[c >> (bitDepth - 5)];
        else if (job.type == 2)
        {
            int idx = 2 + sign3(c - (int)p[n0]) + sign3(c - (int)p[-n0]);
            idx = idx > 2 ? idx : (idx == 2 ? 0 : idx + 1);
            v = c + job.offsets[idx];
        }
        dst[y * strideDst + x] = (T)min(max(v, 0), mx);
    }
}

} // namespace

hipError_t launch_sao_stats(hipStream_t st, int S, int bitDepth, const void *src, long strideSrc, const void *rec, long strideRec, const void *jobs, int njobs,
                            int64_t *out)
{
    if (njobs <= 0) return hipSuccess;
    const SaoStatsJob *j = static_cast<const SaoStatsJob *>(jobs);
    if (S == 1) hipLaunchKernelGGL(k_sao_stats<1>, dim3(njobs), dim3(256), 0, st, (const char *)src, strideSrc, (const char *)rec, strideRec, j, bitDepth - 8, (long long *)out);
    else hipLaunchKernelGGL(k_sao_stats<2>, dim3(njobs), dim3(256), 0, st, (const char *)src, strideSrc, (const char *)rec, strideRec, j, bitDepth - 8, (long long *)out);
    return hipGetLastError();
}

hipError_t launch_sao_band_chroma(hipStream_t st, int S, int bitDepth, const void *src, long strideSrc, const void *rec, long strideRec, const void *jobs, int njobs,
                                  int64_t *out)
{
    if (njobs <= 0) return hipSuccess;
    const SaoChromaJob *j = static_cast<const SaoChromaJob *>(jobs);
    if (S == 1) hipLaunchKernelGGL(k_sao_band_chroma<1>, dim3(njobs), dim3(256), 0, st, (const char *)src, strideSrc, (const char *)rec, strideRec, j, bitDepth - 8, (long long *)out);
    else hipLaunchKernelGGL(k_sao_band_chroma<2>, dim3(njobs), dim3(256), 0, st, (const char *)src, strideSrc, (const char *)rec, strideRec, j, bitDepth - 8, (long long *)out);
    return hipGetLastError();
}

hipError_t launch_sao_filter(hipStream_t st, int S, int bitDepth, void *dst, long strideDst, const void *src, long strideSrc, const void *jobs, int njobs)
{
    if (njobs <= 0) return hipSuccess;
    const SaoJob *j = static_cast<const SaoJob *>(jobs);
    if (S == 1) hipLaunchKernelGGL(k_sao_filter<1>, dim3(njobs), dim3(256), 0, st, (char *)dst, strideDst, (const char *)src, strideSrc, j, bitDepth);
    else hipLaunchKernelGGL(k_sao_filter<2>, dim3(njobs), dim3(256), 0, st, (char *)dst, strideDst, (const char *)src, strideSrc, j, bitDepth);
    return hipGetLastError();
}

} // namespace havoc_gpu
