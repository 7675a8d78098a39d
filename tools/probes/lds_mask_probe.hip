// Probe: does an exec-masked LDS store cost as much as a full one?  (k_run256v2's stash: 4 active lanes per wave.)
//   ./lds_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE> __global__ __launch_bounds__(256) void k(float *out, int reps)
{
    __shared__ __attribute__((aligned(16))) float L[16384];
    const unsigned a16 = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 8192, a8 = (threadIdx.x & 63) * 8 + (threadIdx.x >> 6) * 8192;
    v4f q = {1.f, 2.f, 3.f, (float)threadIdx.x};
    v2f p = {1.f, (float)threadIdx.x};
    const bool act = MODE == 0 || MODE == 3 ? true : (MODE == 1 || MODE == 4 ? (threadIdx.x & 15) == 15 : (threadIdx.x & 63) == 63);
    for (int r = 0; r < reps; r++) {
        if (act) {
            if (MODE < 3)
                asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %1 offset:1024\n\tds_write_b128 %0, %1 offset:2048\n\tds_write_b128 %0, %1 offset:3072\n\t"
                             "ds_write_b128 %0, %1 offset:4096\n\tds_write_b128 %0, %1 offset:5120\n\tds_write_b128 %0, %1 offset:6144\n\tds_write_b128 %0, %1 offset:7168\n\t"
                             "s_waitcnt lgkmcnt(0)" :: "v"(a16), "v"(q) : "memory");
            else
                asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %1 offset:512\n\tds_write_b64 %0, %1 offset:1024\n\tds_write_b64 %0, %1 offset:1536\n\t"
                             "ds_write_b64 %0, %1 offset:2048\n\tds_write_b64 %0, %1 offset:2560\n\tds_write_b64 %0, %1 offset:3072\n\tds_write_b64 %0, %1 offset:3584\n\t"
                             "s_waitcnt lgkmcnt(0)" :: "v"(a8), "v"(p) : "memory");
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = L[threadIdx.x];
}
template <int MODE> static void run(const char *name)
{
    float *out; (void)hipMalloc(&out, 4 * 256 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int reps = 2000; float ms = 0;
    for (int it = 0; it < 2; it++) { (void)hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(256), 0, 0, out, reps); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1); }
    // 4 blocks per CU x 4 waves: LDS cycles per wave-instruction at the CU = wall * 2.4e9 / (reps * 8 * 16)
    printf("%-40s %8.1f us  -> %.2f cycles per wave-instruction per CU (@2.4 GHz)\n", name, ms * 1e3, ms * 1e-3 * 2.4e9 / (reps * 8.0 * 16.0));
    (void)hipFree(out);
}
int main()
{
    run<0>("ds_write_b128, all 64 lanes");
    run<1>("ds_write_b128, lanes 15/31/47/63");
    run<2>("ds_write_b128, lane 63 only");
    run<3>("ds_write_b64, all 64 lanes");
    run<4>("ds_write_b64, lanes 15/31/47/63");
    run<5>("ds_write_b64, lane 63 only");
    return 0;
}
