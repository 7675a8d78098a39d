// Multi-stage resampler (resampler r as, Liquid.chs:56-117: msresamp_crcf_create / _execute): half-band
// decimators + one arbitrary-rate polyphase stage.  Filters and the exact Q32.32 output timing are fixed in
// design.cpp (design_msresamp); every output sample is independent given the integer time, so each stage is one
// data-parallel launch over a history-prefixed buffer.  Bound: HBM (8 B read + 8 r B written per stage sample,
// taps and the 14 KiB filter bank stay in L1/L2).
#include "../../include/csdr.h"
#include "csdr_internal.h"

namespace csdr {

namespace {

__global__ __launch_bounds__(256) void k_hb_decim(const float2 *__restrict__ w, const float *__restrict__ h, float2 *__restrict__ y,
                                                  uint32_t ny, uint32_t base0, uint32_t m)
{
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= ny) return;
    const float2 *p = w + base0 + 2 * (size_t)j;
    float re = 0.f, im = 0.f;
    // same tap order as the restatement: i = 0 .. 4m (the even-offset taps of a half-band filter are zero
    // except the centre; they are kept so that the f32 sums round alike)
    for (uint32_t i = 0; i <= 4 * m; i++) {
        const float2 v = *(p - i);
        const float c = h[i];
        re = fmaf(c, v.x, re); im = fmaf(c, v.y, im);
    }
    y[j] = make_float2(re, im);
}

__global__ __launch_bounds__(256) void k_resamp_arb(const float2 *__restrict__ w, const float *__restrict__ pfb, float2 *__restrict__ y,
                                                    uint32_t ny, uint64_t t_first, uint64_t delta, uint32_t npfb, uint32_t P)
{
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= ny) return;
    const uint64_t t = t_first + (uint64_t)k * delta;
    const uint64_t n = t >> 32;
    const uint32_t frac = (uint32_t)t;
    const uint32_t b = frac >> 24;                               // npfb = 256 phases
    const float mu = (float)(frac & 0xffffffu) * (1.0f / 16777216.0f);
    const float *f0 = pfb + (size_t)b * P;
    const bool wrap = b + 1 >= npfb;
    const float *f1 = wrap ? pfb : f0 + P;
    const float2 *p0 = w + n, *p1 = p0 + (wrap ? 1 : 0);
    float r0 = 0.f, i0 = 0.f, r1 = 0.f, i1 = 0.f;
    for (uint32_t j = 0; j < P; j++) {
        const float2 v0 = *(p0 - j), v1 = *(p1 - j);
        r0 = fmaf(f0[j], v0.x, r0); i0 = fmaf(f0[j], v0.y, i0);
        r1 = fmaf(f1[j], v1.x, r1); i1 = fmaf(f1[j], v1.y, i1);
    }
    const float a = 1.0f - mu;
    y[k] = make_float2(fmaf(mu, r1, a * r0), fmaf(mu, i1, a * i0));
}

__global__ __launch_bounds__(256) void k_keep_tail(float2 *w, uint32_t H, uint32_t n)
{
    // one workgroup: read everything, then write (source and destination may overlap)
    float2 v[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { const uint32_t i = threadIdx.x + 256 * q; if (i < H) v[q] = w[(size_t)n + i]; }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++) { const uint32_t i = threadIdx.x + 256 * q; if (i < H) w[i] = v[q]; }
}

}  // namespace

int launch_hb_decim(const float2 *w, const float *h, float2 *y, uint32_t ny, uint32_t base0, uint32_t m, hipStream_t s)
{
    if (!ny) return 0;
    hipLaunchKernelGGL(k_hb_decim, dim3((ny + 255) / 256), dim3(256), 0, s, w, h, y, ny, base0, m);
    CSDR_HIP(hipGetLastError());
    return 0;
}

int launch_resamp_arb(const float2 *w, const float *pfb, float2 *y, uint32_t ny, uint64_t t_first, uint64_t delta, uint32_t npfb,
                      uint32_t P, hipStream_t s)
{
    if (!ny) return 0;
    hipLaunchKernelGGL(k_resamp_arb, dim3((ny + 255) / 256), dim3(256), 0, s, w, pfb, y, ny, t_first, delta, npfb, P);
    CSDR_HIP(hipGetLastError());
    return 0;
}

int launch_keep_tail(float2 *w, uint32_t H, uint32_t n, hipStream_t s)
{
    if (!n) return 0;
    if (H > 1024) { set_error("resampler: history %u too long", H); return CSDR_ERR_INVALID; }
    hipLaunchKernelGGL(k_keep_tail, dim3(1), dim3(256), 0, s, w, H, n);
    CSDR_HIP(hipGetLastError());
    return 0;
}

}  // namespace csdr
