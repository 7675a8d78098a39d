// How much of the gain wave's time per sample is stall?  K independent AGC streams per lane run the warm-up quad of k_agc_spec_tm
// (agc_common.h's gain update, packed energy, the squelch "does anything move" test per quad) out of registers, K = 1, 2, 4, with
// 1 or 2 waves per SIMD.  Reports shader cycles (s_memtime at 100 MHz -> converted with the measured clock ratio is not needed:
// the figure of interest is the RATIO between K = 1 and K = 2, 4 per stream-sample).
//   ./chain2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../../composable_sdr_amd/csrc/agc_common.h"

typedef float v2f __attribute__((ext_vector_type(2)));

struct St { float g, y2; uint32_t S; };

__device__ __forceinline__ void quad(const float4 va, const float4 vb, St &q, float alpha, float g_thr)
{
    const v2f xr0 = {va.x, va.z}, xi0 = {va.y, va.w}, xr1 = {vb.x, vb.z}, xi1 = {vb.y, vb.w};
    const v2f al = {alpha, alpha};
    const v2f e0 = al * __builtin_elementwise_fma(xr0, xr0, xi0 * xi0), e1 = al * __builtin_elementwise_fma(xr1, xr1, xi1 * xi1);
    const float e[4] = {e0.x, e0.y, e1.x, e1.y};
    unsigned long long mex[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        csdr::agc_gain_update(e[i], q.g, q.y2, alpha);
        asm("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(mex[i]) : "v"(q.g), "v"(g_thr));
    }
    unsigned long long m3, m1;
    asm("v_cmp_eq_u32_e64 %0, 3, %1" : "=s"(m3) : "v"(q.S));
    asm("v_cmp_eq_u32_e64 %0, 1, %1" : "=s"(m1) : "v"(q.S));
    const unsigned long long all_ex = mex[0] & mex[1] & mex[2] & mex[3], any_ex = mex[0] | mex[1] | mex[2] | mex[3];
    const unsigned long long steady = (m3 & all_ex) | (m1 & ~any_ex);
    const unsigned long long act = __builtin_amdgcn_ballot_w64(true);
    if ((act & ~steady) != 0ull) q.S = (q.g < g_thr) ? 3u : 1u;
}

template <int K>
__global__ __launch_bounds__(64) void k_chain(float *out, uint64_t *cyc, int n, float alpha, float thr, const float4 *xin)
{
    St q[K];
    float4 va[K], vb[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        q[k].g = 1.0f + 0.001f * threadIdx.x + 0.01f * k; q[k].y2 = 1.0f; q[k].S = 3;
        va[k] = xin[(threadIdx.x + 7 * k) & 63]; vb[k] = xin[(threadIdx.x + 7 * k + 3) & 63];
    }
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i += 4) {
#pragma unroll
        for (int k = 0; k < K; k++) quad(va[k], vb[k], q[k], alpha, thr);
#pragma unroll
        for (int k = 0; k < K; k++) { va[k].x += 1e-6f; vb[k].y -= 1e-6f; }       // (the inputs are not loop-invariant)
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < K; k++) acc += q[k].g + q[k].y2 + (float)q[k].S;
    out[blockIdx.x * 64 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int K> static void run(int waves)
{
    float *out; uint64_t *cyc; float4 *xin; const int n = 4096;
    hipMalloc(&out, waves * 64 * 4); hipMalloc(&cyc, waves * 8); hipMalloc(&xin, 64 * 16);
    float4 hx[64];
    for (int i = 0; i < 64; i++) hx[i] = make_float4(0.5f + 0.01f * i, -0.4f + 0.007f * i, 0.3f - 0.004f * i, 0.6f - 0.003f * i);
    hipMemcpy(xin, hx, sizeof(hx), hipMemcpyHostToDevice);
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k_chain<K>, dim3(waves), dim3(64), 0, 0, out, cyc, n, 0.01f, 0.05f, xin);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_chain<K>, dim3(waves), dim3(64), 0, 0, out, cyc, n, 0.01f, 0.05f, xin);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    static uint64_t h[8192]; hipMemcpy(h, cyc, waves * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < waves; i++) s += (double)h[i];
    printf("  K = %d streams per lane, %5d waves : %.2f memtime ticks per step = %.2f per stream-sample; kernel %.1f us = %.1f ns per stream-sample and wave\n",
           K, waves, s / waves / n, s / waves / n / K, ms * 1e3, ms * 1e6 / n / K);
    hipFree(out); hipFree(cyc); hipFree(xin);
}

int main()
{
    for (int waves : {1024, 2048, 4096}) { run<1>(waves); run<2>(waves); run<4>(waves); }
    return 0;
}
