// Micro-benchmark: what an EMPTY kernel costs on this GPU as a function of grid size, workgroup size, static LDS and VGPR
// budget -- i.e. the wave-launch floor under every short kernel of the step.  Build: hipcc --offload-arch=gfx950 -O3 -o
// launch_rate launch_rate.hip ; run on the GPU box (profiles/gpu_call_s.sh).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int LDS, int REGS>
__global__ __launch_bounds__(256) void k_empty(int *out, int never)
{
    __shared__ int s[LDS / 4 > 0 ? LDS / 4 : 1];
    if (never)
    {   // keeps the LDS allocation and a register budget alive without executing anything
        int v[REGS];
        for (int i = 0; i < REGS; ++i) v[i] = out[threadIdx.x + i];
        s[threadIdx.x % (LDS / 4 > 0 ? LDS / 4 : 1)] = v[0];
        __syncthreads();
        int a = 0;
        for (int i = 0; i < REGS; ++i) a += v[i] * s[(threadIdx.x + i) % (LDS / 4 > 0 ? LDS / 4 : 1)];
        out[threadIdx.x] = a;
    }
}

template <int LDS, int REGS>
static void run(const char *name, int wgs, int threads, int *d)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int reps = 200;
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k_empty<LDS, REGS>), dim3(wgs), dim3(threads), 0, 0, d, 0);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_empty<LDS, REGS>), dim3(wgs), dim3(threads), 0, 0, d, 0);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    printf("%-28s wgs %6d x %3d threads (%6d waves): %7.2f us per launch\n", name, wgs, threads, wgs * ((threads + 63) / 64), ms * 1e3 / reps);
}

int main()
{
    int *d;
    hipMalloc(&d, 1 << 20);
    for (int wgs : {1, 256, 1024, 2040, 4080, 8160, 16320, 32640})
        run<0, 1>("no LDS", wgs, 256, d);
    for (int threads : {64, 128, 256, 512, 1024})
        run<0, 1>("8160 waves by block size", 8160 * 64 / threads, threads, d);
    run<13312, 1>("13 KB LDS", 2040, 256, d);
    run<32768, 1>("32 KB LDS", 2040, 256, d);
    run<65536, 1>("64 KB LDS", 2040, 256, d);
    run<0, 64>("64 live registers", 2040, 256, d);
    run<13312, 64>("13 KB LDS + 64 registers", 2040, 256, d);
    hipFree(d);
    return 0;
}
