// Probe: s_memtime tick rate vs wall clock, and dependent-FMA latency in ticks, at 1 wave/CU and at full occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(float *out, unsigned long long *ticks, int iters)
{
    float a = threadIdx.x * 1e-9f, b = 1.0000001f, c = 1e-7f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) a = __builtin_fmaf(a, b, c);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
int main()
{
    float *out; unsigned long long *tk;
    hipMalloc(&out, 4 * 256 * 65536); hipMalloc(&tk, 8 * 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 200000;
    for (int cfg = 0; cfg < 3; cfg++) {
        int blocks = cfg == 0 ? 256 : cfg == 1 ? 1024 : 2048, threads = cfg == 0 ? 64 : 256;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0); k<<<blocks, threads>>>(out, tk, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks); hipMemcpy(h.data(), tk, 8 * blocks, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= blocks;
        printf("blocks %d x %d thr: wall %.3f ms, avg ticks %.0f -> tick rate %.3f GHz (if ticks span ~ wall), ticks per dependent fma %.2f\n",
               blocks, threads, ms, avg, avg / (ms * 1e6), avg / (16.0 * iters));
    }
    return 0;
}
