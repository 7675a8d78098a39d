// Access-pattern probe for the AGC tail: every wave reads 64 streams that lie `stride` bytes apart, `piece` bytes of
// each per block, block after block (the next block continues each stream), the way k_agc_spec does.  Reports the
// bandwidth the pattern reaches with nothing else going on, for several piece sizes and waves per SIMD.
//   ./stride_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int PIECE>                                            // bytes per stream per block: 128, 256, 512, 1024
__global__ __launch_bounds__(64) void k_probe(const float4 *__restrict__ src, float *__restrict__ sink, size_t stride16,
                                              uint32_t nblk, size_t wave_span16)
{
    constexpr int LPS = PIECE / 16;                             // lanes that share one stream in one instruction
    constexpr int SPI = 64 / LPS;                               // streams per instruction
    constexpr int NI = 64 / SPI;                                // instructions per block
    const int lane = threadIdx.x;
    const float4 *base = src + (size_t)blockIdx.x * wave_span16;
    float acc = 0.f;
    for (uint32_t k = 0; k < nblk; k++) {
        float4 v[NI];
#pragma unroll
        for (int m = 0; m < NI; m++) {
            const int j = SPI * m + lane / LPS, pc = lane % LPS;
            v[m] = base[(size_t)j * stride16 + (size_t)k * LPS + pc];
        }
#pragma unroll
        for (int m = 0; m < NI; m++) acc += v[m].x + v[m].y + v[m].z + v[m].w;
    }
    if (acc == 123.456f) sink[blockIdx.x * 64 + lane] = acc;
}

template <int PIECE>
static void run(const float4 *src, float *sink, size_t total_bytes, uint32_t seg_bytes, uint32_t waves)
{
    // each wave owns 64 consecutive segments of seg_bytes and reads all of them once
    const size_t stride16 = seg_bytes / 16, span16 = 64 * stride16;
    const uint32_t nblk = seg_bytes / PIECE;
    const uint32_t nw = (uint32_t)(total_bytes / (64ull * seg_bytes));
    (void)waves;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int it = 0; it < 5; it++) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k_probe<PIECE>, dim3(nw), dim3(64), 0, 0, src, sink, stride16, nblk, span16);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    printf("  piece %4d B  seg %6u B  waves %5u : %.3f ms  %.0f GB/s\n", PIECE, seg_bytes, nw, best, (double)nw * 64 * seg_bytes / best * 1e-6);
}

int main()
{
    const size_t total = 1ull << 30;                            // 1 GiB, like 2^27 CF32 samples
    float4 *src; float *sink;
    hipMalloc(&src, total); hipMalloc(&sink, 1 << 24);
    hipMemset(src, 0, total);
    for (uint32_t seg : {3072u, 6144u, 12288u, 24576u}) {       // L = 384, 768, 1536, 3072 samples
        printf("segment %u B (%u CF32 samples)\n", seg, seg / 8);
        run<128>(src, sink, total, seg, 0);
        run<256>(src, sink, total, seg, 0);
        run<512>(src, sink, total, seg, 0);
        run<1024>(src, sink, total, seg, 0);
    }
    return 0;
}
