// Generic (any channel count M, any chunk length) HIP kernels for gfx950.
// One kernel per stage of the reference's chain; used for M that the fused kernels
// do not cover, for the standalone Pipes (dcBlocker, mixDown/mixUp, AGC, freqdem)
// and for the sequential AGC tail.  Wave = 64 lanes throughout.
#include "csdr_internal.h"
#include "fm_common.h"
#include "agc_common.h"
#include "fft16_generic.h"

namespace csdr {

// ---------------------------------------------------------------------------
// DC blocker (iirfilt_crcf dc_blocker; Liquid.chs:575-589) as a 3-kernel chained
// scan of the linear recurrence  v[n] = x[n] + beta*v[n-1],  y[n] = v[n] - v[n-1].
//   k_dc_aggregate : per block of DC_BLOCK samples, A_b = sum beta^(B-1-i) x[i]
//   k_dc_carry     : one workgroup scans the block carries c_{b+1} = beta^B c_b + A_b
//   k_dc_apply     : re-runs the recurrence inside each block from its carry, in the
//                    reference's operation order (v0 = x - a1*v1 ; y = v0 - v1),
//                    then optionally applies the NCO mix and stores.
// ---------------------------------------------------------------------------

__device__ __forceinline__ float2 cmadd(float2 a, float s, float2 b)  // a*s + b
{
    return make_float2(fmaf(a.x, s, b.x), fmaf(a.y, s, b.y));
}

// inclusive Hillis-Steele scan over the 256 per-thread aggregates held in LDS with
// the decaying weight beta^(DC_PER_THREAD*d) per hop of distance d.
__device__ __forceinline__ float2 block_scan_decay(float2 mine, float2 *sh, const DcParams &dc)
{
    const int q = threadIdx.x;
    sh[q] = mine;
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 8; s++) {
        const int d = 1 << s;
        float2 other = make_float2(0.f, 0.f);
        if (q >= d) other = sh[q - d];
        __syncthreads();
        mine = cmadd(other, dc.beta_pow_thr[s], mine);
        sh[q] = mine;
        __syncthreads();
    }
    return mine;
}

__global__ __launch_bounds__(DC_THREADS) void k_dc_aggregate(const float2 *__restrict__ x, uint32_t n,
                                                             DcParams dc, float2 *__restrict__ agg)
{
    __shared__ float2 sh[DC_THREADS];
    const uint32_t base = blockIdx.x * DC_BLOCK + threadIdx.x * DC_PER_THREAD;
    float2 a = make_float2(0.f, 0.f);
#pragma unroll
    for (int i = 0; i < DC_PER_THREAD; i++) {
        float2 xi = (base + i < n) ? x[base + i] : make_float2(0.f, 0.f);
        a = cmadd(a, dc.beta, xi);
    }
    float2 s = block_scan_decay(a, sh, dc);
    if (threadIdx.x == DC_THREADS - 1) agg[blockIdx.x] = s;
}

// carries[b] = v just before block b (b = 0 .. nb-1).  Single workgroup.
__global__ __launch_bounds__(256) void k_dc_carry(const float2 *__restrict__ agg, uint32_t nb,
                                                  DcParams dc, const float2 *__restrict__ state,
                                                  float2 *__restrict__ carries)
{
    __shared__ double shr[256], shi[256];
    const int q = threadIdx.x;
    const uint32_t K = (nb + 255) / 256;          // blocks per thread
    const uint32_t b0 = q * K;
    // local aggregate of my K blocks
    double ar = 0.0, ai = 0.0;
    for (uint32_t i = 0; i < K; i++) {
        uint32_t b = b0 + i;
        float2 A = (b < nb) ? agg[b] : make_float2(0.f, 0.f);
        ar = ar * dc.beta_blk + A.x;
        ai = ai * dc.beta_blk + A.y;
    }
    // inclusive scan across threads, hop weight beta_blk^(K*d)
    shr[q] = ar; shi[q] = ai;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        double orr = 0.0, oi = 0.0;
        if (q >= d) { orr = shr[q - d]; oi = shi[q - d]; }
        __syncthreads();
        double w = pow(dc.beta_blk, (double)K * (double)d);
        ar += w * orr; ai += w * oi;
        shr[q] = ar; shi[q] = ai;
        __syncthreads();
    }
    // exclusive prefix of my range + the stream state decayed to my start
    double er = (q > 0) ? shr[q - 1] : 0.0, ei = (q > 0) ? shi[q - 1] : 0.0;
    const float2 st = state[0];
    double w0 = pow(dc.beta_blk, (double)b0);
    double cr = er + w0 * st.x, ci = ei + w0 * st.y;
    for (uint32_t i = 0; i < K; i++) {
        uint32_t b = b0 + i;
        if (b >= nb) break;
        carries[b] = make_float2((float)cr, (float)ci);
        float2 A = agg[b];
        cr = cr * dc.beta_blk + A.x;
        ci = ci * dc.beta_blk + A.y;
    }
}

__device__ __forceinline__ float2 nco_rotate(float2 v, uint32_t idx, const NcoParams &nco,
                                             const float2 *__restrict__ tab)
{
    float c, s;
    if (nco.tab_len) {
        float2 cs = tab[(nco.tab_pos + idx) % nco.tab_len];
        c = cs.x; s = cs.y;
    } else {
        uint32_t theta = nco.theta0 + idx * nco.d_theta;
        float ph = (float)(6.283185307179586 * (double)(float)theta / 4294967296.0);
        sincosf(ph, &s, &c);
    }
    if (!nco.up) s = -s;                            // multiply by conj(v)
    return make_float2(v.x * c - v.y * s, v.x * s + v.y * c);
}

__global__ __launch_bounds__(DC_THREADS) void k_dc_apply(const float2 *__restrict__ x, float2 *__restrict__ y,
                                                         uint32_t n, int do_dc, DcParams dc,
                                                         const float2 *__restrict__ carries,
                                                         float2 *__restrict__ state_out, int do_mix,
                                                         NcoParams nco, const float2 *__restrict__ tab)
{
    __shared__ float2 sh[DC_THREADS];
    const int q = threadIdx.x;
    const uint32_t base = blockIdx.x * DC_BLOCK + q * DC_PER_THREAD;
    float2 xs[DC_PER_THREAD];
    const bool full = base + DC_PER_THREAD <= n;            // 64-byte aligned run of 8 samples
    if (full) {
        const float4 *x4 = reinterpret_cast<const float4 *>(x + base);
#pragma unroll
        for (int i = 0; i < DC_PER_THREAD / 2; i++) {
            const float4 t = x4[i];
            xs[2 * i] = make_float2(t.x, t.y); xs[2 * i + 1] = make_float2(t.z, t.w);
        }
    } else {
#pragma unroll
        for (int i = 0; i < DC_PER_THREAD; i++)
            xs[i] = (base + i < n) ? x[base + i] : make_float2(0.f, 0.f);
    }

    float2 v1 = make_float2(0.f, 0.f);
    if (do_dc) {
        float2 a = make_float2(0.f, 0.f);
#pragma unroll
        for (int i = 0; i < DC_PER_THREAD; i++) a = cmadd(a, dc.beta, xs[i]);
        float2 incl = block_scan_decay(a, sh, dc);
        (void)incl;
        float2 excl = (q > 0) ? sh[q - 1] : make_float2(0.f, 0.f);
        // carry decayed to my first sample: beta^(DC_PER_THREAD*q)
        const float w = exp2f((float)(DC_PER_THREAD * q) * dc.log2_beta);
        const float2 c = carries[blockIdx.x];
        v1 = cmadd(c, w, excl);
    }
    float2 os[DC_PER_THREAD];
#pragma unroll
    for (int i = 0; i < DC_PER_THREAD; i++) {
        float2 o = xs[i];
        if (do_dc) {
            // iirfilt_crcf_execute_norm: v0 = x - a1*v1 ; y = v0 - v1
            float2 v0 = make_float2(__fsub_rn(xs[i].x, __fmul_rn(dc.a1, v1.x)),
                                    __fsub_rn(xs[i].y, __fmul_rn(dc.a1, v1.y)));
            o = make_float2(v0.x - v1.x, v0.y - v1.y);
            v1 = v0;
            if (base + i == n - 1) state_out[0] = v0;
        }
        if (do_mix) o = nco_rotate(o, base + i, nco, tab);
        os[i] = o;
    }
    if (full) {
        float4 *y4 = reinterpret_cast<float4 *>(y + base);
#pragma unroll
        for (int i = 0; i < DC_PER_THREAD / 2; i++) y4[i] = make_float4(os[2 * i].x, os[2 * i].y, os[2 * i + 1].x, os[2 * i + 1].y);
    } else {
#pragma unroll
        for (int i = 0; i < DC_PER_THREAD; i++) if (base + i < n) y[base + i] = os[i];
    }
}

int launch_dc_mix(const float2 *x, float2 *y, uint32_t n, bool do_dc, const DcParams &dc,
                  float2 *state, float2 *scratch, bool do_mix, const NcoParams &nco,
                  const float2 *nco_tab, hipStream_t s)
{
    if (n == 0) return 0;
    const uint32_t nb = (n + DC_BLOCK - 1) / DC_BLOCK;
    float2 *agg = scratch, *carries = scratch + nb;
    if (do_dc) {
        hipLaunchKernelGGL(k_dc_aggregate, dim3(nb), dim3(DC_THREADS), 0, s, x, n, dc, agg);
        hipLaunchKernelGGL(k_dc_carry, dim3(1), dim3(256), 0, s, agg, nb, dc, state, carries);
    }
    hipLaunchKernelGGL(k_dc_apply, dim3(nb), dim3(DC_THREADS), 0, s, x, y, n, (int)do_dc, dc, carries,
                       state, (int)do_mix, nco, nco_tab);
    CSDR_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// Polyphase branch filters (firpfbch_crcf analyzer_push + dotprod; Liquid.chs:843).
// ---------------------------------------------------------------------------
// One thread = one polyphase branch j x FIR_F consecutive frames: the p+FIR_F-1 window samples
// are loaded once (coalesced along j) and slide through registers.
constexpr int FIR_F = 8;
__global__ __launch_bounds__(256) void k_pfb_fir(const float2 *__restrict__ u, const float *__restrict__ taps,
                                                 float2 *__restrict__ X, uint32_t M, uint32_t p, uint32_t nf)
{
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t j = (uint32_t)(gid % M);
    const int64_t t0 = (int64_t)(gid / M) * FIR_F;
    if (t0 >= (int64_t)nf) return;
    if (p == 14) {
        float h[14];
#pragma unroll
        for (int n = 0; n < 14; n++) h[n] = taps[(M - 1 - j) + (uint32_t)n * M];
        float2 w[13 + FIR_F];
#pragma unroll
        for (int i = 0; i < 13 + FIR_F; i++) {
            const int64_t t = t0 - 13 + i;
            w[i] = (t < (int64_t)nf) ? u[t * (int64_t)M + j] : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int f = 0; f < FIR_F; f++) {
            float2 acc = make_float2(0.f, 0.f);
#pragma unroll
            for (int n = 13; n >= 0; n--) {                      // oldest sample first
                acc.x = fmaf(h[n], w[13 + f - n].x, acc.x);
                acc.y = fmaf(h[n], w[13 + f - n].y, acc.y);
            }
            if (t0 + f < (int64_t)nf) X[(t0 + f) * (int64_t)M + j] = acc;
        }
    } else {
        for (int f = 0; f < FIR_F && t0 + f < (int64_t)nf; f++) {
            float2 acc = make_float2(0.f, 0.f);
            for (int nn = (int)p - 1; nn >= 0; nn--) {
                const float h = taps[(M - 1 - j) + (uint32_t)nn * M];
                const float2 v = u[(t0 + f - nn) * (int64_t)M + j];
                acc.x = fmaf(h, v.x, acc.x);
                acc.y = fmaf(h, v.y, acc.y);
            }
            X[(t0 + f) * (int64_t)M + j] = acc;
        }
    }
}

int launch_pfb_fir(const float2 *u, const float *taps, float2 *X, uint32_t M, uint32_t p, uint32_t nf,
                   hipStream_t s)
{
    if (!nf) return 0;
    const uint64_t total = (uint64_t)M * ((nf + FIR_F - 1) / FIR_F);
    hipLaunchKernelGGL(k_pfb_fir, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, u, taps, X, M, p, nf);
    CSDR_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// Forward DFT per frame.  Power-of-two M: in-LDS radix-2 (one workgroup per frame);
// other M: direct O(M^2) sum (small channel counts such as README Example 3's 20).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fft_pow2(const float2 *__restrict__ X, float2 *__restrict__ Y,
                                                  const float2 *__restrict__ tw, uint32_t M, uint32_t lg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *a = reinterpret_cast<float2 *>(smem_raw);
    const float2 *src = X + (uint64_t)blockIdx.x * M;
    for (uint32_t i = threadIdx.x; i < M; i += blockDim.x) {
        uint32_t r = __brev(i) >> (32 - lg);
        a[r] = src[i];
    }
    __syncthreads();
    for (uint32_t len = 2; len <= M; len <<= 1) {
        const uint32_t half = len >> 1, step = M / len;
        for (uint32_t b = threadIdx.x; b < M / 2; b += blockDim.x) {
            const uint32_t k = b % half, s0 = (b / half) * len;
            const float2 w = tw[k * step];
            const float2 lo = a[s0 + k], hi = a[s0 + k + half];
            const float2 tt = make_float2(hi.x * w.x - hi.y * w.y, hi.x * w.y + hi.y * w.x);
            a[s0 + k] = make_float2(lo.x + tt.x, lo.y + tt.y);
            a[s0 + k + half] = make_float2(lo.x - tt.x, lo.y - tt.y);
        }
        __syncthreads();
    }
    float2 *dst = Y + (uint64_t)blockIdx.x * M;
    for (uint32_t i = threadIdx.x; i < M; i += blockDim.x) dst[i] = a[i];
}

__global__ __launch_bounds__(256) void k_dft_direct(const float2 *__restrict__ X, float2 *__restrict__ Y,
                                                    const float2 *__restrict__ tw, uint32_t M, uint64_t total)
{
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const uint32_t k = (uint32_t)(gid % M);
    const float2 *src = X + (gid / M) * M;
    float sr = 0.f, si = 0.f;
    uint32_t idx = 0;
    for (uint32_t j = 0; j < M; j++) {
        const float2 w = tw[idx], v = src[j];
        sr += v.x * w.x - v.y * w.y;
        si += v.x * w.y + v.y * w.x;
        idx += k; if (idx >= M) idx -= M;
    }
    Y[gid] = make_float2(sr, si);
}


// ---------------------------------------------------------------------------
// Forward DFT for N = 1024 (16*16*4) and N = 4096 (16*16*16): three register passes
// (radix 16, 16, R3) with two LDS exchanges per 4096-point tile (4 frames or 1 frame).
// ---------------------------------------------------------------------------


// MIX: instead of writing Y[nf][N], fold every frame over its N channels in k_mix_frames' order (thread t adds
// channels t, t+256, ... ascending, then the same tree over the 256 threads) and write one sample per frame:
// --mix over all channels without the 16 B/sample round trip of Y through HBM (bit-identical to the two-kernel path).
template <int R3, bool MIX>   // N = 256 * R3
__global__ __launch_bounds__(256) void k_fft_r16(const float2 *__restrict__ X, float2 *__restrict__ Y,
                                                 const float2 *__restrict__ tw, uint32_t nf)
{
    constexpr int N = 256 * R3, F = 4096 / N;             // frames per tile
    constexpr int AS = 17 * R3;                           // padded stride between k1 rows of the pass-1 image
    __shared__ float2 bufA[F * 16 * AS];
    __shared__ float2 bufB[4096];
    const int t = threadIdx.x;
    const uint64_t f0 = (uint64_t)blockIdx.x * F;
    v2fg v[16];
    // ---- pass 1: radix 16 over n1 (stride N/16) for (frame, m) ----
    {
        const int fr = t / (N / 16), m = t % (N / 16);
        const bool ok = f0 + fr < nf;
        const float2 *src = X + (f0 + fr) * N + m;
#pragma unroll
        for (int n1 = 0; n1 < 16; n1++) { const float2 x = ok ? src[(N / 16) * n1] : make_float2(0.f, 0.f); v[n1] = (v2fg){x.x, x.y}; }
        g_fft16(v);
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int k1 = GXIDX(i);
            if (k1) { const float2 w = tw[m * k1]; v[i] = g_cmul(v[i], (v2fg){w.x, w.y}); }
            bufA[fr * 16 * AS + k1 * AS + m] = make_float2(v[i].x, v[i].y);
        }
    }
    __syncthreads();
    // ---- pass 2: radix 16 over n2 for (frame, k1, n3) ----
    {
        const int n3 = t % R3, k1 = (t / R3) % 16, fr = t / (16 * R3);
#pragma unroll
        for (int n2 = 0; n2 < 16; n2++) { const float2 x = bufA[fr * 16 * AS + k1 * AS + R3 * n2 + n3]; v[n2] = (v2fg){x.x, x.y}; }
        g_fft16(v);
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int k2 = GXIDX(i);
            if (k2) {
                // unconditional load + select: a load under the lane-varying `n3 != 0` would be waited for on its
                // own, fifteen round trips in a row (the n3 = 0 lanes keep their value untouched, as before)
                const float2 w = tw[16 * n3 * k2];
                const v2fg r = g_cmul(v[i], (v2fg){w.x, w.y});
                v[i] = n3 ? r : v[i];
            }
            bufB[fr * N + (k1 + 16 * k2) * R3 + n3] = make_float2(v[i].x, v[i].y);
        }
    }
    __syncthreads();
    // ---- pass 3: radix R3 over n3 for (frame, k1 + 16 k2), straight to global ----
    float2 part[F];                                            // MIX: my channels' partial sum per frame
    if (R3 == 16) {
        const bool ok = f0 < nf;
#pragma unroll
        for (int n3 = 0; n3 < 16; n3++) { const float2 x = bufB[t * 16 + n3]; v[n3] = (v2fg){x.x, x.y}; }
        g_fft16(v);
        if (MIX) {
            float2 acc = make_float2(0.f, 0.f);
#pragma unroll
            for (int k3 = 0; k3 < 16; k3++) {                   // ascending channel order: v[i] holds k3 = GXIDX(i)
                const int i = 4 * (k3 & 3) + (k3 >> 2);
                acc.x += v[i].x; acc.y += v[i].y;
            }
            part[0] = acc;
        } else if (ok) {
#pragma unroll
            for (int i = 0; i < 16; i++) Y[f0 * N + t + 256 * GXIDX(i)] = make_float2(v[i].x, v[i].y);
        }
    } else {
#pragma unroll
        for (int fr = 0; fr < F; fr++) {
            v2fg a[4];
#pragma unroll
            for (int n3 = 0; n3 < 4; n3++) { const float2 x = bufB[fr * N + t * 4 + n3]; a[n3] = (v2fg){x.x, x.y}; }
            g_bfly4(a[0], a[1], a[2], a[3]);
            if (MIX) {
                float2 acc = make_float2(0.f, 0.f);
#pragma unroll
                for (int k3 = 0; k3 < 4; k3++) { acc.x += a[k3].x; acc.y += a[k3].y; }
                part[fr] = acc;
            } else if (f0 + fr < nf) {
#pragma unroll
                for (int k3 = 0; k3 < 4; k3++) Y[(f0 + fr) * N + t + 256 * k3] = make_float2(a[k3].x, a[k3].y);
            }
        }
    }
    if (MIX) {
        __syncthreads();                                        // bufB consumed
        float2 *red = bufB;                                     // [F][256]
#pragma unroll
        for (int fr = 0; fr < F; fr++) red[fr * 256 + t] = part[fr];
        __syncthreads();
        // the same tree as before (t += t + d for d = 128 ... 1, so the sums round identically), but only the two levels
        // that cross waves go through LDS and a barrier; the rest runs inside wave 0 on shuffles
        if (t < 128) {
#pragma unroll
            for (int fr = 0; fr < F; fr++) { red[fr * 256 + t].x += red[fr * 256 + t + 128].x; red[fr * 256 + t].y += red[fr * 256 + t + 128].y; }
        }
        __syncthreads();
        if (t < 64) {
#pragma unroll
            for (int fr = 0; fr < F; fr++) {
                float2 x = red[fr * 256 + t];
                const float2 u = red[fr * 256 + t + 64];
                x.x += u.x; x.y += u.y;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) { x.x += __shfl_down(x.x, d); x.y += __shfl_down(x.y, d); }
                if (t == 0 && f0 + fr < nf) Y[f0 + fr] = x;     // Y is the mixed output [nf] here
            }
        }
    }
}

int launch_dft(const float2 *X, float2 *Y, const float2 *tw, uint32_t M, uint32_t nf, hipStream_t s)
{
    if (!nf) return 0;
    if (M == 1) {
        CSDR_HIP(hipMemcpyAsync(Y, X, sizeof(float2) * nf, hipMemcpyDeviceToDevice, s));
        return 0;
    }
    if (M == 1024) {
        hipLaunchKernelGGL((k_fft_r16<4, false>), dim3((nf + 3) / 4), dim3(256), 0, s, X, Y, tw, nf);
    } else if (M == 4096) {
        hipLaunchKernelGGL((k_fft_r16<16, false>), dim3(nf), dim3(256), 0, s, X, Y, tw, nf);
    } else if ((M & (M - 1)) == 0 && M <= 8192) {
        uint32_t lg = 0; while ((1u << lg) < M) lg++;
        unsigned th = M / 2 < 64 ? 64 : (M / 2 > 256 ? 256 : M / 2);
        hipLaunchKernelGGL(k_fft_pow2, dim3(nf), dim3(th), M * sizeof(float2), s, X, Y, tw, M, lg);
    } else {
        const uint64_t total = (uint64_t)M * nf;
        hipLaunchKernelGGL(k_dft_direct, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, X, Y, tw, M, total);
    }
    CSDR_HIP(hipGetLastError());
    return 0;
}

// Pruned DFT for an interleaved channel shard (channels g, g + G, ...): Y[g + G m] = sum_{j1 < M/G} z[j1] W_{M/G}^{j1 m} with
// z[j1] = W_M^{j1 g} sum_{j2 < G} X[j1 + (M/G) j2] W_G^{j2 g}.  This kernel makes z (one thread per (frame, j1)); the
// (M/G)-point DFT follows as an ordinary launch_dft.  ph: G phasors W_G^{j2 g}, then M/G phasors W_M^{j1 g}.
__global__ __launch_bounds__(256) void k_fold(const float2 *__restrict__ X, float2 *__restrict__ Z, const float2 *__restrict__ ph,
                                             uint32_t M, uint32_t G, uint64_t total)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const uint32_t Mg = M / G;
    const uint64_t t = i / Mg;
    const uint32_t j1 = (uint32_t)(i - t * Mg);
    const float2 *x = X + t * M + j1;
    float2 acc = make_float2(0.f, 0.f);
    for (uint32_t j2 = 0; j2 < G; j2++) {
        const float2 v = x[(size_t)j2 * Mg], w = ph[j2];
        acc.x = fmaf(v.x, w.x, fmaf(-v.y, w.y, acc.x));
        acc.y = fmaf(v.x, w.y, fmaf(v.y, w.x, acc.y));
    }
    const float2 w = ph[G + j1];
    Z[i] = make_float2(acc.x * w.x - acc.y * w.y, acc.x * w.y + acc.y * w.x);
}

int launch_fold(const float2 *X, float2 *Z, const float2 *ph, uint32_t M, uint32_t G, uint32_t nf, hipStream_t s)
{
    const uint64_t total = (uint64_t)(M / G) * nf;
    if (!total) return 0;
    hipLaunchKernelGGL(k_fold, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, X, Z, ph, M, G, total);
    CSDR_HIP(hipGetLastError());
    return 0;
}

// DFT of every frame folded over all M channels: out[t] = sum_k Y[t][k] (DeNo --mix, Trans.hs:119-122), M = 1024 / 4096
bool dft_mix_supported(uint32_t M) { return M == 1024 || M == 4096; }
int launch_dft_mix(const float2 *X, float2 *out, const float2 *tw, uint32_t M, uint32_t nf, hipStream_t s)
{
    if (!nf) return 0;
    if (M == 1024) hipLaunchKernelGGL((k_fft_r16<4, true>), dim3((nf + 3) / 4), dim3(256), 0, s, X, out, tw, nf);
    else if (M == 4096) hipLaunchKernelGGL((k_fft_r16<16, true>), dim3(nf), dim3(256), 0, s, X, out, tw, nf);
    else { set_error("dft_mix: unsupported M=%u", M); return -1; }
    CSDR_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// [nf][M] -> channel-major [C][nf]  (the transpose of Liquid.chs:840-844)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_transpose(const float2 *__restrict__ Y, float2 *__restrict__ Z,
                                                   uint32_t M, uint32_t nf, uint32_t c0, uint32_t C)
{
    __shared__ float2 tile[32][33];
    const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const uint32_t cb = blockIdx.x * 32, tb = blockIdx.y * 32;
    for (uint32_t r = ty; r < 32; r += 8) {
        uint32_t t = tb + r, c = cb + tx;
        if (t < nf && c < C) tile[r][tx] = Y[(uint64_t)t * M + c0 + c];
    }
    __syncthreads();
    for (uint32_t r = ty; r < 32; r += 8) {
        uint32_t c = cb + r, t = tb + tx;
        if (t < nf && c < C) Z[(uint64_t)c * nf + t] = tile[tx][r];
    }
}

int launch_transpose(const float2 *Y, float2 *Z, uint32_t M, uint32_t nf, uint32_t c0, uint32_t C, hipStream_t s)
{
    if (!nf || !C) return 0;
    hipLaunchKernelGGL(k_transpose, dim3((C + 31) / 32, (nf + 31) / 32), dim3(256), 0, s, Y, Z, M, nf, c0, C);
    CSDR_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// agc_crcf + squelch, one lane per channel walking its samples in order
// (agcExecuteBlock, Liquid.chs:695-705).  Exactly sequential per channel.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_agc_init(AgcState *st, uint32_t C)
{
    uint32_t c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    // agcCreate (Liquid.chs:707-717): signal level 1e-3 -> g = 1000, y2' = 1, squelch ENABLED
    st[c].g = 1.0f / 1e-3f; st[c].y2 = 1.0f; st[c].mode = 1; st[c].timer = 1000u;
}

int launch_agc_init(AgcState *st, uint32_t C, hipStream_t s)
{
    hipLaunchKernelGGL(k_agc_init, dim3((C + 63) / 64), dim3(64), 0, s, st, C);
    CSDR_HIP(hipGetLastError());
    return 0;
}

__device__ __forceinline__ float2 agc_step(float2 x, AgcState &q, const AgcParams &p)
{
    // agc_crcf_execute: y = x*g ; y2' <- (1-alpha) y2' + alpha |y|^2 ; g <- g * exp(-alpha/2 * ln y2')
    float2 y = make_float2(x.x * q.g, x.y * q.g);
    agc_gain_update(agc_energy(x, p.alpha), q.g, q.y2, p.alpha);     // agc_common.h: the shared, chain-shortened float path
    const bool ex = q.g < p.g_thr;                    // rssi > threshold
    // squelch state machine (agc_crcf_squelch_update_mode), modes 1..6
    int m = q.mode;
    const bool lo_to = (m == 5) && (q.timer == 1u);
    q.timer = (m == 4) ? p.timeout : ((m == 5) ? q.timer - 1u : q.timer);
    const int nxt_ex = (m == 1) ? 2 : ((m == 6) ? 1 : 3);                 // threshold exceeded
    const int nxt_no = (m == 1) ? 1 : ((m == 4) ? 5 : ((m == 5) ? 5 : ((m == 6) ? 1 : 4)));
    m = ex ? nxt_ex : nxt_no;
    m = lo_to ? 6 : m;
    q.mode = m;
    if (m != 3) y = make_float2(0.f, 0.f);            // reference mute rule (Liquid.chs:703-704)
    return y;
}

// One lane per channel; each lane streams its row in blocks of 16 samples (128 B = one cache
// line per lane and block), the next block's loads in flight while the current one runs through
// the recurrence, so the loop is bound by the AGC's dependent chain and not by memory latency.
__global__ __launch_bounds__(64) void k_agc(float2 *__restrict__ Z, uint32_t C, uint32_t nf,
                                            AgcState *__restrict__ st, AgcParams p)
{
    const uint32_t c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    AgcState q = st[c];
    float2 *row = Z + (uint64_t)c * nf;
    const bool vec = ((uint64_t)c * nf) % 2 == 0;            // 16-byte aligned row start
    uint32_t t = 0;
    if (vec && nf >= 16) {
        float4 cur[8], nxt[8];
        const float4 *r4 = reinterpret_cast<const float4 *>(row);
#pragma unroll
        for (int i = 0; i < 8; i++) cur[i] = r4[i];
        for (; t + 16 <= nf; t += 16) {
            const bool more = t + 32 <= nf;
            if (more) {
#pragma unroll
                for (int i = 0; i < 8; i++) nxt[i] = r4[(t + 16) / 2 + i];
            }
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const float2 a = agc_step(make_float2(cur[i].x, cur[i].y), q, p);
                const float2 b = agc_step(make_float2(cur[i].z, cur[i].w), q, p);
                reinterpret_cast<float4 *>(row)[t / 2 + i] = make_float4(a.x, a.y, b.x, b.y);
            }
            if (more) {
#pragma unroll
                for (int i = 0; i < 8; i++) cur[i] = nxt[i];
            }
        }
    }
    for (; t < nf; t++) row[t] = agc_step(row[t], q, p);
    st[c] = q;
}

int launch_agc(float2 *Z, uint32_t C, uint32_t nf, AgcState *st, const AgcParams &p, hipStream_t s)
{
    if (!nf || !C) return 0;
    hipLaunchKernelGGL(k_agc, dim3((C + 63) / 64), dim3(64), 0, s, Z, C, nf, st, p);
    CSDR_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// freqdem (Liquid.chs:303-334): m = arg(conj(r') r) / (2 pi kf)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fm(const float2 *__restrict__ Z, float *__restrict__ F, uint32_t nf,
                                            uint64_t total, float ref, const float2 *__restrict__ rp_in,
                                            float2 *__restrict__ rp_out)
{
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const uint32_t t = (uint32_t)(gid % nf);
    const uint32_t c = (uint32_t)(gid / nf);
    const float2 r = Z[gid];
    const float2 rp = t ? Z[gid - 1] : rp_in[c];
    F[gid] = fm_sample_rn(rp, r, ref);
    if (t == nf - 1) rp_out[c] = r;
}

int launch_fm(const float2 *Z, float *F, uint32_t C, uint32_t nf, float ref, const float2 *rp_in,
              float2 *rp_out, hipStream_t s)
{
    const uint64_t total = (uint64_t)C * nf;
    if (!total) return 0;
    hipLaunchKernelGGL(k_fm, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, Z, F, nf, total, ref, rp_in, rp_out);
    CSDR_HIP(hipGetLastError());
    return 0;
}


// ---------------------------------------------------------------------------
// Frame-major tails (generic path, M > 1 without AGC): freqdem only needs the same channel of
// the previous frame, which is the previous ROW of Y[nf][M] -- no transpose needed before it.
//   k_transpose_fm : F[c][t] = arg(conj(Y[t-1][c0+c]) Y[t][c0+c]) * ref, 32x32 tiles through LDS
//   k_mix_frames   : out[t] = sum_c (FM ? m[t][c] : Y[t][c0+c]), one workgroup per frame
// ---------------------------------------------------------------------------
__device__ __forceinline__ float fm_sample(float2 rp, float2 r, float ref)
{
    return fm_sample_rn(rp, r, ref);
}

__global__ __launch_bounds__(256) void k_transpose_fm(const float2 *__restrict__ Y, float *__restrict__ F, uint32_t M,
                                                      uint32_t nf, uint32_t c0, uint32_t C, float ref,
                                                      const float2 *__restrict__ rp_in, float2 *__restrict__ rp_out)
{
    __shared__ float tile[32][33];
    const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const uint32_t cb = blockIdx.x * 32, tb = blockIdx.y * 32;
    for (uint32_t r = ty; r < 32; r += 8) {
        const uint32_t t = tb + r, c = cb + tx;
        if (t < nf && c < C) {
            const float2 cur = Y[(uint64_t)t * M + c0 + c];
            const float2 prv = t ? Y[(uint64_t)(t - 1) * M + c0 + c] : rp_in[c];
            tile[r][tx] = fm_sample(prv, cur, ref);
            if (t == nf - 1) rp_out[c] = cur;
        }
    }
    __syncthreads();
    for (uint32_t r = ty; r < 32; r += 8) {
        const uint32_t c = cb + r, t = tb + tx;
        if (t < nf && c < C) F[(uint64_t)c * nf + t] = tile[tx][r];
    }
}

template <bool FM>
__global__ __launch_bounds__(256) void k_mix_frames(const float2 *__restrict__ Y, void *__restrict__ out, uint32_t M,
                                                    uint32_t nf, uint32_t c0, uint32_t C, float ref,
                                                    const float2 *__restrict__ rp_in, float2 *__restrict__ rp_out)
{
    __shared__ float2 red[256];
    const uint32_t t = blockIdx.x;
    const float2 *row = Y + (uint64_t)t * M + c0;
    const float2 *prow = t ? Y + (uint64_t)(t - 1) * M + c0 : nullptr;
    // thread i folds channels i, i+256, ... in ascending order; then an ordered tree over threads
    float2 acc = make_float2(0.f, 0.f);
    for (uint32_t c = threadIdx.x; c < C; c += 256) {
        const float2 cur = row[c];
        if (FM) {
            const float2 prv = t ? prow[c] : rp_in[c];
            acc.x += fm_sample(prv, cur, ref);
            if (t == nf - 1) rp_out[c] = cur;
        } else { acc.x += cur.x; acc.y += cur.y; }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) { red[threadIdx.x].x += red[threadIdx.x + d].x; red[threadIdx.x].y += red[threadIdx.x + d].y; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (FM) ((float *)out)[t] = red[0].x; else ((float2 *)out)[t] = red[0];
    }
}

int launch_transpose_fm(const float2 *Y, float *F, uint32_t M, uint32_t nf, uint32_t c0, uint32_t C, float ref,
                        const float2 *rp_in, float2 *rp_out, hipStream_t s)
{
    if (!nf || !C) return 0;
    hipLaunchKernelGGL(k_transpose_fm, dim3((C + 31) / 32, (nf + 31) / 32), dim3(256), 0, s, Y, F, M, nf, c0, C, ref, rp_in, rp_out);
    CSDR_HIP(hipGetLastError());
    return 0;
}

int launch_mix_frames(const float2 *Y, void *out, bool fm, uint32_t M, uint32_t nf, uint32_t c0, uint32_t C, float ref,
                      const float2 *rp_in, float2 *rp_out, hipStream_t s)
{
    if (!nf || !C) return 0;
    if (fm) hipLaunchKernelGGL(k_mix_frames<true>, dim3(nf), dim3(256), 0, s, Y, out, M, nf, c0, C, ref, rp_in, rp_out);
    else hipLaunchKernelGGL(k_mix_frames<false>, dim3(nf), dim3(256), 0, s, Y, out, M, nf, c0, C, ref, rp_in, rp_out);
    CSDR_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------
// mix (Trans.hs:119-122): strict left fold over the channel list
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mix(const float *__restrict__ in, float *__restrict__ out, uint32_t C, uint32_t E)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= E) return;
    float acc = in[i];
    for (uint32_t c = 1; c < C; c++) acc = __fadd_rn(acc, in[(uint64_t)c * E + i]);
    out[i] = acc;
}

// CSDR_FLAG_DFT_BACKWARD: out row k = in row (C - k) mod C; rows of W 4-byte words (V = 4: whole 16-byte vectors)
template <int V> __global__ void k_rows_reversed(const float *__restrict__ in, float *__restrict__ out, uint32_t C, size_t W)
{
    const uint32_t k = blockIdx.y, src = k ? C - k : 0u;
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (i >= W) return;
    if (V == 4) *reinterpret_cast<float4 *>(out + (size_t)k * W + i) = *reinterpret_cast<const float4 *>(in + (size_t)src * W + i);
    else out[(size_t)k * W + i] = in[(size_t)src * W + i];
}

int launch_rows_reversed(const void *in, void *out, uint32_t C, size_t row_bytes, hipStream_t s)
{
    const size_t W = row_bytes / 4;
    if (!W || !C) return 0;
    if (W % 4 == 0 && ((size_t)in | (size_t)out) % 16 == 0)
        hipLaunchKernelGGL(k_rows_reversed<4>, dim3((unsigned)((W / 4 + 255) / 256), C), dim3(256), 0, s, (const float *)in, (float *)out, C, W);
    else
        hipLaunchKernelGGL(k_rows_reversed<1>, dim3((unsigned)((W + 255) / 256), C), dim3(256), 0, s, (const float *)in, (float *)out, C, W);
    CSDR_HIP(hipGetLastError());
    return 0;
}

int launch_mix(const float *in, float *out, uint32_t C, uint32_t E, hipStream_t s)
{
    if (!E) return 0;
    hipLaunchKernelGGL(k_mix, dim3((E + 255) / 256), dim3(256), 0, s, in, out, C, E);
    CSDR_HIP(hipGetLastError());
    return 0;
}

}  // namespace csdr
