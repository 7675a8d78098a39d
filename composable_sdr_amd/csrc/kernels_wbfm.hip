// WBFM audio tail on channel-major F32 rows (wbFMDemodulator, Liquid.chs:653-656 = firDecimator decim . iirDeemph .
// fmDemodulator 0.6):
//   k_biquad   : iirFilter 2 fc 0 10 10 (Liquid.chs:636-638): one direct-form-II second-order section per channel
//                  v0 = x - a1 v1 - a2 v2 ;  y = b0 v0 + b1 v1 + b2 v2
//                The recurrence is linear: s' = A s + B x with s = (v1, v2).  One workgroup per channel walks its
//                row in chunks of 4096 samples; a thread runs 16 consecutive samples from zero state, a Hillis-
//                Steele scan over the 256 thread end states with the precomputed powers A^(16 2^k) yields every
//                thread's true start state, a second pass produces the outputs.  The state crosses chunks (and
//                calls) exactly; only the f32 summation order differs from the sequential loop.
//   k_firdecim : firDecimator m (Liquid.chs:485-501): y[j] = sum_i h[i] x[jM - i] over a per-channel history prefix.
// Arithmetic recalled from liquid-dsp 1.3.2 (iirdes / iirfiltsos / firdecim): unpinned, DESIGN.md 4.8.
#include "../../include/csdr.h"
#include "csdr_internal.h"

namespace csdr {

namespace {

constexpr int BQ_T = 256, BQ_PER = 16, BQ_CHUNK = BQ_T * BQ_PER;

__global__ __launch_bounds__(BQ_T) void k_biquad(const float *__restrict__ X, float *__restrict__ Y, uint32_t nf, BiquadParams p,
                                                 const float2 *__restrict__ st_in, float2 *__restrict__ st_out)
{
    __shared__ float xs[17 * BQ_T];
    __shared__ double2 sc[2][BQ_T];
    __shared__ float2 carry_s;
    const int tid = threadIdx.x;
    const uint32_t c = blockIdx.x;
    const float *row = X + (size_t)c * nf;
    float *orow = Y + (size_t)c * nf;
    if (tid == 0) carry_s = st_in[c];
    __syncthreads();
    const float k1 = p.b1 - p.b0 * p.a1, k2 = p.b2 - p.b0 * p.a2;       // y = b0 x + k1 v1 + k2 v2 (pre-state)
    for (uint32_t base = 0; base < nf; base += BQ_CHUNK) {
#pragma unroll
        for (int i = 0; i < BQ_PER; i++) {
            const uint32_t s = tid + BQ_T * i, t = base + s;
            xs[17 * (s >> 4) + (s & 15)] = t < nf ? row[t] : 0.f;
        }
        __syncthreads();
        const float2 cin = carry_s;
        float x[BQ_PER];
        // pass 1: end state from zero state (thread 0: from the carried state, so that its end state is the true one)
        float v1 = tid == 0 ? cin.x : 0.f, v2 = tid == 0 ? cin.y : 0.f;
#pragma unroll
        for (int k = 0; k < BQ_PER; k++) {
            x[k] = xs[17 * tid + k];
            const float v0 = fmaf(-p.a2, v2, fmaf(-p.a1, v1, x[k]));
            v2 = v1; v1 = v0;
        }
        // inclusive scan of s_i = A^16 s_{i-1} + e_i over the threads, in f64: a narrow low-pass keeps a state
        // thousands of times larger than its output, and A^n has entries ~n for poles near the unit circle
        double2 sv = make_double2((double)v1, (double)v2);
        int cur = 0;
        sc[0][tid] = sv;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int d = 1 << k;
            if (tid >= d) {
                const double2 u = sc[cur][tid - d];
                sv.x = fma(p.pw[k][0], u.x, fma(p.pw[k][1], u.y, sv.x));
                sv.y = fma(p.pw[k][2], u.x, fma(p.pw[k][3], u.y, sv.y));
            }
            sc[cur ^ 1][tid] = sv;
            cur ^= 1;
            __syncthreads();
        }
        // pass 2 from the true start state
        if (tid) { const double2 st = sc[cur][tid - 1]; v1 = (float)st.x; v2 = (float)st.y; }
        else { v1 = cin.x; v2 = cin.y; }
#pragma unroll
        for (int k = 0; k < BQ_PER; k++) {
            const float y = fmaf(p.b0, x[k], fmaf(k1, v1, k2 * v2));
            const float v0 = fmaf(-p.a2, v2, fmaf(-p.a1, v1, x[k]));
            v2 = v1; v1 = v0;
            xs[17 * tid + k] = y;
            // the row may end inside this chunk: the state after its last sample is what the next call needs
            if (base + (uint32_t)(BQ_PER * tid + k) == nf - 1) st_out[c] = make_float2(v1, v2);
        }
        if (tid == BQ_T - 1) carry_s = make_float2(v1, v2);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < BQ_PER; i++) {
            const uint32_t s = tid + BQ_T * i, t = base + s;
            if (t < nf) orow[t] = xs[17 * (s >> 4) + (s & 15)];
        }
        __syncthreads();
    }
}

// out[c][j] = sum_i h[i] x[c][jM - i]; x[c][t < 0] = hist_in[c][H + t], H = h_len - 1
__global__ __launch_bounds__(256) void k_firdecim(const float *__restrict__ X, float *__restrict__ out, uint32_t nf, uint32_t M,
                                                  const float *__restrict__ h, uint32_t h_len, const float *__restrict__ hist_in,
                                                  float *__restrict__ hist_out)
{
    const uint32_t c = blockIdx.y, no = nf / M, H = h_len - 1;
    const float *row = X + (size_t)c * nf, *hin = hist_in + (size_t)c * H;
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j < no) {
        const int64_t t0 = (int64_t)j * M;
        float acc = 0.f;
        for (uint32_t i = 0; i < h_len; i++) {
            const int64_t t = t0 - i;
            const float v = t >= 0 ? row[t] : hin[(int64_t)H + t];
            acc = fmaf(h[i], v, acc);
        }
        out[(size_t)c * no + j] = acc;
    }
    // the first workgroup of every row also moves the history forward: the last H samples of (hist | row)
    if (blockIdx.x == 0) {
        for (uint32_t i = threadIdx.x; i < H; i += 256) {
            const int64_t t = (int64_t)nf - H + i;
            hist_out[(size_t)c * H + i] = t >= 0 ? row[t] : hin[(int64_t)H + t];
        }
    }
}

}  // namespace

int launch_biquad(const float *X, float *Y, uint32_t C, uint32_t nf, const BiquadParams &p, const float2 *st_in, float2 *st_out,
                  hipStream_t s)
{
    if (!C || !nf) return 0;
    hipLaunchKernelGGL(k_biquad, dim3(C), dim3(BQ_T), 0, s, X, Y, nf, p, st_in, st_out);
    CSDR_HIP(hipGetLastError());
    return 0;
}

int launch_firdecim(const float *X, float *out, uint32_t C, uint32_t nf, uint32_t M, const float *h, uint32_t h_len,
                    const float *hist_in, float *hist_out, hipStream_t s)
{
    if (!C || !nf) return 0;
    const uint32_t no = nf / M;
    hipLaunchKernelGGL(k_firdecim, dim3((no + 255) / 256 ? (no + 255) / 256 : 1, C), dim3(256), 0, s, X, out, nf, M, h, h_len, hist_in, hist_out);
    CSDR_HIP(hipGetLastError());
    return 0;
}

}  // namespace csdr
