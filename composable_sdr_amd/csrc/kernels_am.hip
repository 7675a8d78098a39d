// ampmodem DSB / carrier-present demodulator (amDemodulator, Liquid.chs:439-469: ampmodem_create(0.8, DSB, 0)),
// per channel on channel-major rows Z[C][nf] -> F[C][nf]:
//     t = |y| ;  q_hat <- alpha t + (1 - alpha) q_hat ;  x = 2 (t - q_hat)           (alpha = 0.01, q_hat0 = 0)
// -- liquid-dsp 1.3.2's non-coherent peak detector as recalled (unpinned: nothing in the reference confirms the constants).
//
// The smoother is linear, so time is cut into chunks of 2048 samples per workgroup: a workgroup stages its chunk
// and the 2048 samples before it (warm-up: 0.99^2048 = 1.2e-9 of the older state is dropped; the first chunk of a
// call starts from the stored q_hat instead), every thread scans 16 consecutive samples from zero, a decayed scan
// over the 256 thread totals gives each thread its carry, and the outputs leave coalesced.  16 B read + 4 B
// written per sample, no inter-workgroup dependency.
#include "../../include/csdr.h"
#include "csdr_internal.h"
#include <cmath>

namespace csdr {

namespace {

constexpr int AM_T = 256, AM_PER = 16, AM_TOT = AM_T * AM_PER, AM_REAL = AM_TOT / 2;   // 4096 staged, 2048 produced

__global__ __launch_bounds__(AM_T) void k_am(const float2 *__restrict__ Z, float *__restrict__ F, uint32_t C, uint32_t nf,
                                             const float *__restrict__ q_in, float *__restrict__ q_out, float alpha,
                                             float dec1, float dec16)
{
    __shared__ float ts[17 * AM_T];
    __shared__ float wtot[4];
    const int tid = threadIdx.x;
    const uint32_t c = blockIdx.y, chunk = blockIdx.x;
    const int64_t t0 = (int64_t)chunk * AM_REAL - AM_REAL;          // first staged sample (negative for chunk 0)
    const float2 *row = Z + (size_t)c * nf;
    // clamped, unconditional loads first (all sixteen in flight), the range test afterwards: a load under a branch
    // would be waited for on its own
    float2 y[AM_PER];
#pragma unroll
    for (int i = 0; i < AM_PER; i++) {
        const int64_t t = t0 + tid + AM_T * i;
        y[i] = row[t < 0 ? 0 : (t < (int64_t)nf ? t : (int64_t)nf - 1)];
    }
#pragma unroll
    for (int i = 0; i < AM_PER; i++) {
        const int s = tid + AM_T * i;
        const int64_t t = t0 + s;
        const float v = hypotf(y[i].x, y[i].y);
        ts[17 * (s >> 4) + (s & 15)] = (t >= 0 && t < (int64_t)nf) ? v : 0.f;
    }
    __syncthreads();
    // zero-state scan of my 16 samples
    float tv[AM_PER], q[AM_PER];
    float acc = 0.f;
    const float beta = 1.0f - alpha;
#pragma unroll
    for (int k = 0; k < AM_PER; k++) {
        tv[k] = ts[17 * tid + k];
        acc = fmaf(beta, acc, alpha * tv[k]);
        q[k] = acc;
    }
    // the stored state enters as the total of the thread in front of the first produced sample
    if (chunk == 0 && tid == AM_T / 2 - 1) acc = q_in[c];
    // decayed inclusive scan of the thread totals: wave level, then across the four waves
    float sc = acc, d = dec16;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float up = __shfl_up(sc, off);
        if ((tid & 63) >= off) sc = fmaf(d, up, sc);
        d = d * d;
    }
    if ((tid & 63) == 63) wtot[tid >> 6] = sc;
    __syncthreads();
    // d == dec16^64 now: decay across one wave
    float carry_w = 0.f;
    for (int w = 0; w < (tid >> 6); w++) carry_w = fmaf(d, carry_w, wtot[w]);
    // exclusive carry of my thread = inclusive of the previous lane (+ the waves in front, decayed to that lane)
    float prev = __shfl_up(sc, 1);
    if ((tid & 63) == 0) prev = 0.f;
    float dl = 1.0f;                                              // dec16^(lane): decay of the wave carry to the lane before me
    {
        float b = dec16; int e = tid & 63;
#pragma unroll
        for (int bit = 0; bit < 6; bit++) { if (e & 1) dl *= b; b *= b; e >>= 1; }
    }
    const float carry = fmaf(dl, carry_w, prev);                  // q_hat just before my first sample
    // the channel's state after the call is q_hat at the row's last sample: staged index `last`
    const int64_t last = (int64_t)nf - 1 - t0;
    const bool owns_last = last >= AM_REAL && last < AM_TOT && (last >> 4) == tid;
    float m = dec1;
#pragma unroll
    for (int k = 0; k < AM_PER; k++) {
        const float qq = fmaf(m, carry, q[k]);
        ts[17 * tid + k] = 2.0f * (tv[k] - qq);
        if (owns_last && k == (int)(last & 15)) q_out[c] = qq;
        m *= dec1;
    }
    __syncthreads();
#pragma unroll
    for (int i = AM_PER / 2; i < AM_PER; i++) {
        const int s = tid + AM_T * i;
        const int64_t t = t0 + s;
        if (t < (int64_t)nf) F[(size_t)c * nf + t] = ts[17 * (s >> 4) + (s & 15)];
    }
}

}  // namespace

int launch_am(const float2 *Z, float *F, uint32_t C, uint32_t nf, const float *q_in, float *q_out, float alpha, hipStream_t s)
{
    if (!C || !nf) return 0;
    const double b = 1.0 - (double)alpha;
    const dim3 grid((nf + AM_REAL - 1) / AM_REAL, C);
    hipLaunchKernelGGL(k_am, grid, dim3(AM_T), 0, s, Z, F, C, nf, q_in, q_out, alpha, (float)b, (float)std::pow(b, 16.0));
    CSDR_HIP(hipGetLastError());
    return 0;
}

}  // namespace csdr
