// Dependent-chain probe for the AGC tail: one wave per SIMD runs N gain steps out of registers (no memory), in
// several variants, and reports shader cycles per step (s_memtime).  Shows what a lone wave pays per dependent sample.
//   ./chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int V>
__global__ __launch_bounds__(64) void k_chain(float *out, uint64_t *cyc, int n, float alpha, float thr, float x0, float x1)
{
    float g = 1.0f + 0.001f * threadIdx.x, y2h = 1.0f;
    uint32_t S = 3;
    float accx = 0.f;
    const float xr = x0 + 1e-3f * threadIdx.x, xi = x1;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    if (V >= 4) {
        // shortened chain: e = alpha |x|^2 is off the chain, y2' = (1-a) y2' + e g^2; clamp by v_med3; V = 5: squelch test
        // per four samples, V = 6: per sixteen
        const float e = alpha * fmaf(xr, xr, xi * xi);
        for (int i = 0; i < n; i += 16) {
            bool all_ex = true, none_ex = true;
#pragma unroll
            for (int u = 0; u < 16; u++) {
                y2h = fmaf(1.0f - alpha, y2h, e * (g * g));
                const float upd = __builtin_amdgcn_exp2f((-0.5f * alpha) * __builtin_amdgcn_logf(y2h));
                g = (y2h > 1e-6f) ? g * upd : g;
                g = __builtin_amdgcn_fmed3f(g, 0.0f, 1e6f);
                const bool ex = g < thr;
                all_ex = all_ex && ex; none_ex = none_ex && !ex;
                accx += (S == 3u) ? xr * g : 0.f;
                if (V == 4 || (V == 5 && (u & 3) == 3) || (V == 6 && u == 15)) {
                    const bool steady = (S == 3u && all_ex) || (S == 1u && none_ex);
                    if (__builtin_amdgcn_ballot_w64(!steady) != 0ull) S = ex ? 3u : 1u;
                    all_ex = true; none_ex = true;
                }
            }
        }
    } else
    for (int i = 0; i < n; i++) {
        const float yr = xr * g, yi = xi * g;
        const float y2 = fmaf(yr, yr, yi * yi);
        y2h = fmaf(1.0f - alpha, y2h, alpha * y2);
        float upd;
        if (V == 1) upd = 1.0f - 0.5f * alpha * (y2h - 1.0f);                          // no transcendental
        else upd = __builtin_amdgcn_exp2f((-0.5f * alpha) * __builtin_amdgcn_logf(y2h));
        g = (y2h > 1e-6f) ? g * upd : g;
        if (V == 3) g = (g < 1e6f) ? g : 1e6f; else g = fminf(g, 1e6f);
        const bool ex = g < thr;
        if (V != 2) {                                                                   // V == 2: no squelch test / branch
            const bool steady = (S == 3u && ex) || (S == 1u && !ex);
            if (__builtin_amdgcn_ballot_w64(!steady) != 0ull) S = ex ? 3u : 1u;
        }
        accx += (S == 3u) ? yr : 0.f;
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = g + accx + y2h;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V> static void run(const char *name, int waves)
{
    float *out; uint64_t *cyc; const int n = 4096;
    hipMalloc(&out, waves * 64 * 4); hipMalloc(&cyc, waves * 8);
    hipLaunchKernelGGL(k_chain<V>, dim3(waves), dim3(64), 0, 0, out, cyc, n, 0.01f, 0.05f, 0.7f, 0.7f);
    hipLaunchKernelGGL(k_chain<V>, dim3(waves), dim3(64), 0, 0, out, cyc, n, 0.01f, 0.05f, 0.7f, 0.7f);
    hipDeviceSynchronize();
    uint64_t h[4096]; hipMemcpy(h, cyc, (waves < 4096 ? waves : 4096) * 8, hipMemcpyDeviceToHost);
    double s = 0; int m = waves < 4096 ? waves : 4096; for (int i = 0; i < m; i++) s += (double)h[i];
    printf("  %-34s waves %5d : %.1f cycles per step\n", name, waves, s / m / n);
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int waves : {1024, 2048}) {                      // 1, 2, 3, 4 per SIMD
        run<0>("full gain step + squelch test", waves);
        run<1>("no log2/exp2", waves);
        run<2>("no squelch test / branch", waves);
        run<3>("select instead of fminf", waves);
        run<4>("short chain, test per sample", waves);
        run<5>("short chain, test per 4", waves);
        run<6>("short chain, test per 16", waves);
    }
    return 0;
}
