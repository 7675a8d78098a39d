// Fused polyphase FIR + 1024-point DFT + per-channel tail for M = 1024 (BASELINE configs[3] shape), behind the
// single-pass DC blocker + NCO mix kernel (k_dc_tile):
//     u (DC-blocked, pre-mixed stream, 13 frames of history in front) -> firpfbch analyzer (Liquid.chs:843)
//       -> channel-major CF32 [C][nf]  or  freqdem F32 [C][nf]   (Liquid.chs:840-862, 303-334)
// The any-M path runs these as three kernels (k_pfb_fir, k_fft_r16, k_transpose(_fm)) with X and Y going
// through HBM (8+8, 8+8, 8+4 bytes per sample); here they are one: 8 B read + 4 (8) B written.
//
// One workgroup = 1024 threads = the 1024 polyphase branches; it walks a RUN of consecutive 8-frame tiles:
//   * thread j keeps the 13-frame FIR window of branch j in registers and loads its 8 new samples straight into
//     them (a wave load instruction covers 512 contiguous bytes); 14 taps per branch stay in registers for the run;
//   * X (8 frames x 1024) goes to LDS, the DFT is 16 x 16 x 4: radix-16 over n1 (stride 64), radix-16 over n2,
//     radix-4 over n3, two LDS exchanges (same index scheme as k_fft_r16<4>); passes 1-2 use 512 threads
//     (one radix-16 butterfly each), pass 3 all 1024;
//   * Y returns to LDS frame-major; thread k reads the 8 frames of channel k (conflict-free both ways), applies
//     freqdem against the r' it keeps in a register, and stores its 32 (64) contiguous bytes of the channel row.
// Runs are independent: the window of a run's first frame is read from u (the history is materialised there);
// the first freqdem sample of a run >= 1 needs the last frame of the run before it: parked in yfirst / ylast and
// finished by k_pfb1024_fixup.  134 KiB of LDS, <= 128 VGPRs: one workgroup (16 waves) per CU.
#include "../../include/csdr.h"
#include "csdr_internal.h"
#include "fm_common.h"
#include "fft16_generic.h"
#include "fused_common.h"
#include "fused.h"
#include <cmath>

namespace csdr {

namespace {

#ifndef CSDR_ABLATE1024
#define CSDR_ABLATE1024 0        // timing experiments only: 1 skip DFT pass 1, 2 skip pass 2, 4 skip the FM tail arithmetic
#endif
constexpr int PM = 1024, PT = 8, PP = 14;       // channels, frames per tile, taps per branch
constexpr int PAS = 17 * 4;                     // padded stride between k1 rows of the pass-1 image (as k_fft_r16<4>)

struct Pfb1024Args {
    const float2 *u;            // first NEW sample; 13 * 1024 samples of history in front
    const float *taps;          // h[(1023 - j) + 1024 n]
    const float4 *taps_t;       // DC = true: the same taps gathered per branch, [1024][16] floats (14 used): four 16-byte loads from one address
    const float2 *tw;           // e^{-j 2 pi i / 1024}
    void *out;                  // [C][nf] F32 (FM) or CF32
    const float2 *rp_in; float2 *rp_out;        // [C] freqdem r'
    float2 *yfirst, *ylast;     // [nruns][1024]
    uint32_t nf, nb, nruns, c0, C;
    float ref;
    // DC = true (k_run1024): u is the RAW input of this call (no history prefix); DC blocker + NCO pre-mix happen here
    const float2 *wpre;         // [2][1024] conj(nco phasor) of branch j for even / odd global frames
    const float2 *vend_in; float2 *vend_out;      // DC blocker state v1
    const float2 *uhist_in; float2 *uhist_out;    // [13][1024] pre-mixed, DC-blocked window before / after the call
    uint32_t parity0;
    float alpha, l2beta;
    float bp[4];                // beta^1, ^2, ^4, ^8       (in-run DPP scan)
    float dp[4];                // beta^16, ^32, ^64, ^128   (scan over the 64 runs of a frame, row part)
    float dm;                   // beta^1024                 (frame to frame)
    PhaseK pk;                  // ref-scaled atan2 polynomial of the run kernels (k_run1024's freqdem)
};

template <int NV> __device__ __forceinline__ void shift_window_n(float2 (&old)[13], const float2 (&nw)[PT])
{
#pragma unroll
    for (int i = 0; i < 13; i++) old[i] = (i + NV < 13) ? old[i + NV] : nw[i + NV - 13];
}
__device__ __forceinline__ void shift_window(float2 (&old)[13], const float2 (&nw)[PT], int nv)
{
    switch (nv) {
    case 1: shift_window_n<1>(old, nw); break;
    case 2: shift_window_n<2>(old, nw); break;
    case 3: shift_window_n<3>(old, nw); break;
    case 4: shift_window_n<4>(old, nw); break;
    case 5: shift_window_n<5>(old, nw); break;
    case 6: shift_window_n<6>(old, nw); break;
    case 7: shift_window_n<7>(old, nw); break;
    default: break;
    }
}

template <bool FM, bool DC>
__global__ __launch_bounds__(1024) void k_pfb1024(Pfb1024Args A)
{
    __shared__ float2 bufA[PT * 16 * PAS];      // 69 632 B
    __shared__ float2 bufB[PT * PM];            // 65 536 B
    __shared__ float2 tw1[16 * 64];             // pass-1 twiddles W1024^(m k1) at [k1][m]
    __shared__ float2 tw2[16 * 4];              // pass-2 twiddles W1024^(16 n3 k2) at [k2][n3]
    __shared__ float2 Tt[PT];                   // DC: frame totals
    __shared__ float2 red[16];
    const int tid = threadIdx.x, j = tid;
    { const int k1 = tid >> 6, m = tid & 63; tw1[tid] = A.tw[(m * k1) & 1023]; }
    if (tid < 64) { const int k2 = tid >> 2, n3 = tid & 3; tw2[tid] = A.tw[(16 * n3 * k2) & 1023]; }
    const uint32_t w = blockIdx.x;
    const uint32_t first = (uint32_t)((uint64_t)w * A.nb / A.nruns), last = (uint32_t)((uint64_t)(w + 1) * A.nb / A.nruns);
    if (first >= last) return;

    const bool owned = (uint32_t)tid >= A.c0 && (uint32_t)tid < A.c0 + A.C;
    float2 prev = (FM && w == 0 && owned) ? A.rp_in[tid - A.c0] : make_float2(0.f, 0.f);
    float2 old[13];
    float2 c = make_float2(0.f, 0.f);           // DC: blocker state at the start of the run (then kept in carry_s)
    uint32_t tile_begin = first;                // DC, run >= 1: two halo tiles refill the window (no output)
    __shared__ float2 carry_s[2];               // DC: blocker state before the tile, ping-pong by tile parity (saves a barrier)
    if (!DC) {
        const int64_t fa = (int64_t)first * PT;
#pragma unroll
        for (int i = 0; i < 13; i++) old[i] = A.u[(fa - 13 + i) * PM + j];          // reaches into the history for run 0
    } else if (w == 0) {
#pragma unroll
        for (int i = 0; i < 13; i++) old[i] = A.uhist_in[i * PM + j];
        c = A.vend_in[0];
    } else {
#pragma unroll
        for (int i = 0; i < 13; i++) old[i] = make_float2(0.f, 0.f);
        // read-only warm-up: DC state before tile first-2 from the three tiles (24 576 samples) in front of it
        // (beta^24576 = 4.6e-6 of the older state is dropped, like k_run256)
        tile_begin = first - 2;
        float wg[PT];
#pragma unroll
        for (int f = 0; f < PT; f++) wg[f] = exp2f(A.l2beta * (float)(PT * PM - 1 - (PM * f + tid)));
        const float btile = exp2f(A.l2beta * (float)(PT * PM));
        float2 acc = make_float2(0.f, 0.f);
#pragma unroll 1
        for (uint32_t t = tile_begin - 3; t < tile_begin; t++) {
            float2 p = make_float2(0.f, 0.f);
#pragma unroll
            for (int f = 0; f < PT; f++) p = cfma(A.u[((size_t)t * PT + f) * PM + j], wg[f], p);
            acc = cfma(acc, btile, p);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { acc.x += __shfl_xor(acc.x, d); acc.y += __shfl_xor(acc.y, d); }
        if ((tid & 63) == 0) red[tid >> 6] = acc;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; i++) { c.x += red[i].x; c.y += red[i].y; }
    }

    float2 nw[PT];
    {
        const uint32_t t0 = tile_begin * PT;
#pragma unroll
        for (int f = 0; f < PT; f++) nw[f] = t0 + f < A.nf ? A.u[(size_t)(t0 + f) * PM + j] : make_float2(0.f, 0.f);
    }
    if (DC && tid == 0) carry_s[0] = c;
    __syncthreads();                                                // twiddle tables, carry_s
    // (decay factors inside the loop use the bare v_exp_f32: results below 2^-126 flush to zero, which is what a
    // factor that small amounts to anyway, and the library exp2f's denormal handling costs ~6 instructions a call)
    for (uint32_t b = tile_begin; b < last; b++) {
        // keep the per-phase address arithmetic inside the iteration (hoisted out of the tile loop it pins dozens of VGPRs)
        int tid_i = tid;
        asm volatile("" : "+v"(tid_i));
        const int j_i = tid_i;
        const uint32_t t0 = b * PT;
        const int nvalid = (int)min((uint32_t)PT, A.nf - t0);
        // the 14 taps (and the two pre-mix phasors) are re-read per tile (L2-resident; held across the tile loop they
        // would spill): issued first, so that their latency hides behind the DC stage
        float h[PP];
        if (DC) {
            const float4 *tp = A.taps_t + 4 * j_i;
            const float4 ha = tp[0], hb = tp[1], hc = tp[2], hd = tp[3];
            h[0] = ha.x; h[1] = ha.y; h[2] = ha.z; h[3] = ha.w; h[4] = hb.x; h[5] = hb.y; h[6] = hb.z; h[7] = hb.w;
            h[8] = hc.x; h[9] = hc.y; h[10] = hc.z; h[11] = hc.w; h[12] = hd.x; h[13] = hd.y;
        } else {
#pragma unroll
            for (int n = 0; n < PP; n++) h[n] = A.taps[(PM - 1 - j_i) + n * PM];
        }
        float2 Wa = make_float2(1.f, 0.f), Wb = Wa;
        if (DC) { Wa = A.wpre[(A.parity0 & 1) * PM + j_i]; Wb = A.wpre[((A.parity0 & 1) ^ 1) * PM + j_i]; }
        if (DC) {
            // ---- DC blocker (zero-state part) on the column-layout registers: in-run scans by DPP, run totals to LDS ----
            float2 *TR = bufA, *E = bufA + 512;
            // the DPP operands must be VGPRs; materialise them inside the iteration (hoisted they get spilled)
            float b1 = A.bp[0], b2 = A.bp[1], b4 = A.bp[2], b8 = A.bp[3], na = -A.alpha;
            asm volatile("" : "+v"(b1), "+v"(b2), "+v"(b4), "+v"(b8), "+v"(na));
#pragma unroll
            for (int f = 0; f < PT; f += 2) {
                float s0, s1, s2, s3;
                row_scan4(nw[f].x, nw[f].y, nw[f + 1].x, nw[f + 1].y, s0, s1, s2, s3, b1, b2, b4, b8, na);
                if ((tid_i & 15) == 15) {
                    TR[64 * f + (tid_i >> 4)] = make_float2(s0, s1);
                    TR[64 * (f + 1) + (tid_i >> 4)] = make_float2(s2, s3);
                }
            }
            __syncthreads();
            // ---- wave f scans the 64 run totals of frame f: four DPP row steps, then the three row-to-row carries ----
            float2 e0 = make_float2(0.f, 0.f);
            if (tid_i < 512) {
                const int l = tid_i & 63;
                const float wrow = __builtin_amdgcn_exp2f(A.l2beta * 16.0f * (float)((tid_i & 15) + 1));   // beta^(16 ((r & 15) + 1))
                float2 sv = TR[tid_i], t;
                t = dpp2<0x111>(sv); sv = cfma(t, A.dp[0], sv);
                t = dpp2<0x112>(sv); sv = cfma(t, A.dp[1], sv);
                t = dpp2<0x114>(sv); sv = cfma(t, A.dp[2], sv);
                t = dpp2<0x118>(sv); sv = cfma(t, A.dp[3], sv);
#pragma unroll
                for (int rho = 1; rho < 4; rho++) {
                    const float2 v = make_float2(__shfl(sv.x, 16 * rho - 1), __shfl(sv.y, 16 * rho - 1));
                    if ((l >> 4) == rho) sv = cfma(v, wrow, sv);
                }
                e0 = make_float2(__shfl_up(sv.x, 1), __shfl_up(sv.y, 1));
                if (l == 0) e0 = make_float2(0.f, 0.f);
                if (l == 63) Tt[tid_i >> 6] = sv;
            }
            __syncthreads();
            // ---- frame carries: state before frame f = vb_f + beta^(1024 f) c; after the tile: ve + beta^8192 c ----
            {
                const int fme = (tid_i >> 6) & 7;
                float2 vb = make_float2(0.f, 0.f), ve = make_float2(0.f, 0.f);
                float dn = 1.0f;                                    // beta^(1024 nvalid): a ragged last tile stops the state there
#pragma unroll
                for (int f = 0; f < PT; f++) {
                    if (f == fme) vb = ve;
                    if (f < nvalid) { ve = cfma(ve, A.dm, Tt[f]); dn *= A.dm; }
                }
                const int par = (int)((b - tile_begin) & 1u);
                const float2 cc = carry_s[par];
                if (tid_i < 512) {
                    const float br = __builtin_amdgcn_exp2f(A.l2beta * 16.0f * (float)(tid_i & 63));           // beta^(16 r), r = run inside the frame
                    const float bf = __builtin_amdgcn_exp2f(A.l2beta * 1024.0f * (float)fme);                 // beta^(1024 f)
                    E[tid_i] = cfma(cfma(cc, bf, vb), br, e0);
                }
                if (tid_i == 0) carry_s[par ^ 1] = cfma(cc, dn, ve);
            }
            __syncthreads();
            // ---- finish the DC blocker, apply the NCO pre-mix ----
            {
                const float kj = -A.alpha * __builtin_amdgcn_exp2f(A.l2beta * (float)(tid_i & 15));
#pragma unroll
                for (int f = 0; f < PT; f++) {
                    const float2 y = cfma(E[64 * f + (j_i >> 4)], kj, nw[f]);
                    nw[f] = cmul(y, (f & 1) ? Wb : Wa);
                }
            }
        }
        if (DC && b < first) {
            // halo tile: only the window moves
#pragma unroll
            for (int i = 0; i < 13; i++) old[i] = (i + PT < 13) ? old[i + PT] : nw[i + PT - 13];
            {
                const uint32_t t1 = t0 + PT;
#pragma unroll
                for (int f = 0; f < PT; f++) nw[f] = t1 + f < A.nf ? A.u[(size_t)(t1 + f) * PM + j_i] : make_float2(0.f, 0.f);
            }
            __syncthreads();                                        // E consumed before the next tile's run totals
            continue;
        }
        // ---- polyphase FIR, oldest tap first; four frames at a time (independent accumulators) ----
#pragma unroll
        for (int f0 = 0; f0 < PT; f0 += 4) {
            v2fg acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
            for (int n = PP - 1; n >= 0; n--) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int i = f0 + q - n;
                    const float2 s2 = (i >= 0) ? nw[i] : old[13 + i];
                    acc[q] = __builtin_elementwise_fma((v2fg){s2.x, s2.y}, (v2fg){h[n], h[n]}, acc[q]);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; q++) bufB[(f0 + q) * PM + j_i] = make_float2(acc[q].x, acc[q].y);
        }
        // next tile's window: the last 13 of (old | nw); a ragged last tile only moves by its valid frames
        if (nvalid == PT) {
#pragma unroll
            for (int i = 0; i < 13; i++) old[i] = (i + PT < 13) ? old[i + PT] : nw[i + PT - 13];
        } else shift_window(old, nw, nvalid);
        __syncthreads();                                            // X complete (in bufB)

        v2fg v[16];
        // ---- pass 1: radix 16 over n1 (stride 64) for (frame, m): threads 0..511 ----
        if (tid_i < 512 && !(CSDR_ABLATE1024 & 1)) {
            const int fr = tid_i >> 6, m = tid_i & 63;
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++) { const float2 x = bufB[fr * PM + 64 * n1 + m]; v[n1] = (v2fg){x.x, x.y}; }
            g_fft16(v);
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int k1 = GXIDX(i);
                if (k1) { const float2 tq = tw1[k1 * 64 + m]; v[i] = g_cmul(v[i], (v2fg){tq.x, tq.y}); }
                bufA[fr * 16 * PAS + k1 * PAS + m] = make_float2(v[i].x, v[i].y);
            }
        }
        __syncthreads();
        // ---- pass 2: radix 16 over n2 for (frame, k1, n3) ----
        if (tid_i < 512 && !(CSDR_ABLATE1024 & 2)) {
            const int n3 = tid_i & 3, k1 = (tid_i >> 2) & 15, fr = tid_i >> 6;
#pragma unroll
            for (int n2 = 0; n2 < 16; n2++) { const float2 x = bufA[fr * 16 * PAS + k1 * PAS + 4 * n2 + n3]; v[n2] = (v2fg){x.x, x.y}; }
            g_fft16(v);
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int k2 = GXIDX(i);
                if (k2) {                                           // read + select: a read under the lane-varying n3 test is waited for on its own
                    const float2 tq = tw2[k2 * 4 + n3];
                    const v2fg r = g_cmul(v[i], (v2fg){tq.x, tq.y});
                    v[i] = n3 ? r : v[i];
                }
                bufB[fr * PM + (k1 + 16 * k2) * 4 + n3] = make_float2(v[i].x, v[i].y);
            }
        }
        __syncthreads();
        // the next tile's samples fly in during pass 3 and the tail (issued here, after the two radix-16 passes, so that
        // they do not sit in registers next to the butterflies)
        if (b + 1 < last) {
            const uint32_t t1 = t0 + PT;
            if (t1 + PT <= A.nf) {
#pragma unroll
                for (int f = 0; f < PT; f++) nw[f] = A.u[(size_t)(t1 + f) * PM + j_i];
            } else {
#pragma unroll
                for (int f = 0; f < PT; f++) nw[f] = t1 + f < A.nf ? A.u[(size_t)(t1 + f) * PM + j_i] : make_float2(0.f, 0.f);
            }
        }
        // ---- pass 3: radix 4 over n3 for (frame, k1 + 16 k2): thread (fh, tq) does frames fh and fh + 4 ----
        {
            const int tq = tid_i & 255, fh = tid_i >> 8;
            v2fg a[2][4];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const int fr = fh + 4 * e;
                const float4 z0 = *reinterpret_cast<const float4 *>(&bufB[fr * PM + tq * 4]);
                const float4 z1 = *reinterpret_cast<const float4 *>(&bufB[fr * PM + tq * 4 + 2]);
                a[e][0] = (v2fg){z0.x, z0.y}; a[e][1] = (v2fg){z0.z, z0.w}; a[e][2] = (v2fg){z1.x, z1.y}; a[e][3] = (v2fg){z1.z, z1.w};
                g_bfly4(a[e][0], a[e][1], a[e][2], a[e][3]);
            }
            // Y frame-major into bufA's space: [8][1024] (bufA's last readers finished before the barrier above)
            float2 *Yl = bufA;
#pragma unroll
            for (int e = 0; e < 2; e++)
#pragma unroll
                for (int k3 = 0; k3 < 4; k3++) Yl[(fh + 4 * e) * PM + tq + 256 * k3] = make_float2(a[e][k3].x, a[e][k3].y);
        }
        __syncthreads();                                            // Y complete
        // ---- tail: thread k = channel k, frames t0 .. t0 + 7 ----
        {
            const float2 *Yl = bufA;
            const bool vec = nvalid == PT && (A.nf & 3) == 0;        // whole tile, 16-byte aligned rows
            if (FM) {
                if (b == first && w > 0) A.yfirst[(size_t)w * PM + tid_i] = Yl[tid_i];
                float *Mt = reinterpret_cast<float *>(bufB);         // [1024][8 + 1] floats: demodulated samples, row-major
#pragma unroll 1
                for (int g = 0; g < PT; g += 4) {
                    float2 y[4];
#pragma unroll
                    for (int f = 0; f < 4; f++) y[f] = Yl[(g + f) * PM + tid_i];
                    const int nv = nvalid - g;
                    float m[4];
#pragma unroll
                    for (int f = 0; f < 4; f++) {
                        const float2 rp = f ? y[f - 1] : prev, r = y[f];
                        // k_pfb1024 keeps the any-M route's routine (bit-identical outputs); k_run1024 the run kernels' lean one
                        m[f] = DC ? scaled_atan2f(fmaf(rp.x, r.y, -(rp.y * r.x)), fmaf(rp.x, r.x, rp.y * r.y), A.pk) : fm_sample_rn(rp, r, A.ref);
                    }
#pragma unroll
                    for (int f = 0; f < 4; f++) if (f < nv) prev = y[f];
                    if (vec) {
#pragma unroll
                        for (int f = 0; f < 4; f++) Mt[tid_i * 9 + g + f] = m[f];
                    } else if (owned && nv > 0) {
                        float *o = (float *)A.out + (size_t)(tid_i - A.c0) * A.nf + t0 + g;
#pragma unroll
                        for (int f = 0; f < 4; f++) if (f < nv) o[f] = m[f];
                    }
                }
                if (vec) {
                    // rows leave as 32-byte segments: two lanes per row, a wave instruction = 32 rows x 32 B
                    __syncthreads();
#pragma unroll
                    for (int it = 0; it < 2; it++) {
                        const int item = tid_i + 1024 * it, rowk = item >> 1, piece = item & 1;
                        if ((uint32_t)rowk >= A.c0 && (uint32_t)rowk < A.c0 + A.C) {
                            const float *src = Mt + rowk * 9 + 4 * piece;
                            *reinterpret_cast<float4 *>((float *)A.out + (size_t)(rowk - A.c0) * A.nf + t0 + 4 * piece) =
                                make_float4(src[0], src[1], src[2], src[3]);
                        }
                    }
                }
            } else if (vec && (A.nf & 1) == 0) {
                // CF32 rows: four lanes per 64-byte row segment (frames 2p, 2p+1 each), a wave instruction = 16 rows x 64 B
#pragma unroll
                for (int it = 0; it < 4; it++) {
                    const int item = tid_i + 1024 * it, rowk = item >> 2, piece = item & 3;
                    if ((uint32_t)rowk >= A.c0 && (uint32_t)rowk < A.c0 + A.C) {
                        const float2 a0 = Yl[(2 * piece) * PM + rowk], a1 = Yl[(2 * piece + 1) * PM + rowk];
                        *reinterpret_cast<float4 *>((float2 *)A.out + (size_t)(rowk - A.c0) * A.nf + t0 + 2 * piece) = make_float4(a0.x, a0.y, a1.x, a1.y);
                    }
                }
            } else if (owned) {
                float2 *o = (float2 *)A.out + (size_t)(tid_i - A.c0) * A.nf + t0;
#pragma unroll 1
                for (int f = 0; f < PT; f++) if (f < nvalid) o[f] = Yl[f * PM + tid_i];
            }
        }
        // Y (bufA) and Mt (bufB) consumed before the next tile overwrites them.  The vectorised FM tail has already
        // passed a barrier after its last read of bufA, and with the DC stage in front bufB is not written again before
        // three more barriers.
        if (!(DC && FM && nvalid == PT && (A.nf & 3) == 0)) __syncthreads();
    }
    if (FM) {
        A.ylast[(size_t)w * PM + tid] = prev;
        if (w + 1 == A.nruns && owned) A.rp_out[tid - A.c0] = prev;
    }
    if (DC && w + 1 == A.nruns) {
        if (tid == 0) A.vend_out[0] = carry_s[(last - tile_begin) & 1u];
#pragma unroll
        for (int i = 0; i < 13; i++) A.uhist_out[i * PM + tid] = old[i];
    }
}

// first freqdem sample of every run w >= 1
__global__ __launch_bounds__(1024) void k_pfb1024_fixup(Pfb1024Args A)
{
    const uint32_t w = blockIdx.x + 1, k = threadIdx.x;
    if (k < A.c0 || k >= A.c0 + A.C) return;
    const uint32_t first = (uint32_t)((uint64_t)w * A.nb / A.nruns), last = (uint32_t)((uint64_t)(w + 1) * A.nb / A.nruns);
    if (first >= last) return;
    // the run before w that is not empty
    uint32_t wp = w - 1;
    while (wp > 0 && (uint32_t)((uint64_t)wp * A.nb / A.nruns) >= (uint32_t)((uint64_t)(wp + 1) * A.nb / A.nruns)) wp--;
    ((float *)A.out)[(size_t)(k - A.c0) * A.nf + (size_t)first * PT] = fm_sample_rn(A.ylast[(size_t)wp * PM + k], A.yfirst[(size_t)w * PM + k], A.ref);
}

}  // namespace

bool pfb1024_supported(uint32_t M, uint32_t p) { return M == 1024 && p == 14; }

// scratch: 2 * nruns * 1024 float2 (nruns <= number of CUs)
int launch_pfb1024(const float2 *u_new, const float *taps, const float2 *tw, void *out, bool fm, uint32_t nf, uint32_t c0, uint32_t C,
                   float ref, const float2 *rp_in, float2 *rp_out, float2 *scratch, uint32_t max_runs, hipStream_t s)
{
    if (!nf || !C) return 0;
    Pfb1024Args A{};
    A.u = u_new; A.taps = taps; A.tw = tw; A.out = out; A.rp_in = rp_in; A.rp_out = rp_out;
    A.nf = nf; A.nb = (nf + PT - 1) / PT; A.c0 = c0; A.C = C; A.ref = ref;
    uint32_t nruns = max_runs ? max_runs : 256;
    if (nruns > A.nb / 4) nruns = A.nb / 4;                    // at least 4 tiles per run
    if (nruns < 1) nruns = 1;
    A.nruns = nruns; A.yfirst = scratch; A.ylast = scratch + (size_t)nruns * PM;
    if (fm) hipLaunchKernelGGL((k_pfb1024<true, false>), dim3(nruns), dim3(1024), 0, s, A);
    else hipLaunchKernelGGL((k_pfb1024<false, false>), dim3(nruns), dim3(1024), 0, s, A);
    if (fm && nruns > 1) hipLaunchKernelGGL(k_pfb1024_fixup, dim3(nruns - 1), dim3(1024), 0, s, A);
    CSDR_HIP(hipGetLastError());
    return 0;
}


// ---------------------------------------------------------------------------------------------
// k_run1024: the same kernel with the DC blocker and the NCO pre-mix fused in (DC = true) -- the whole chain of
// assembleFold for -c 1024 in one launch per chunk, same call interface as the M = 256 / 64 run kernels (fused.h)
// ---------------------------------------------------------------------------------------------
// interleaved shard: row m = j + (16/G) k2 + (256/G) k3 of the shard's plane <- row G j + 16 k2 + 256 k3 of the whole primed band (w floats per sample)
__global__ __launch_bounds__(256) void k_shard_gather1024(const float *__restrict__ src, float *__restrict__ dst, uint32_t G, uint32_t nf, uint32_t w)
{
    const uint32_t m = blockIdx.x, n1 = 16u / G, j = m % n1, k2 = (m / n1) & 15u, k3 = m / (16u * n1);
    const size_t so = (size_t)(G * j + 16u * k2 + 256u * k3) * nf * w, dn = (size_t)m * nf * w;
    for (size_t i = threadIdx.x; i < (size_t)nf * w; i += 256) dst[dn + i] = src[so + i];
}

struct BigPlan {
    FusedConfig cfg;
    uint32_t cus = 256;
    uint64_t frames_done = 0;
    float *d_taps = nullptr;
    float4 *d_taps_t = nullptr, *d_taps_q = nullptr;
    bool v3_ok = false, v3_last = false;      // k_run1024v3 (FM, whole band) selected; used by the last call
    bool v2_ok = false, v2_last = false;      // k_run1024v2 usable (whole band, not disabled); used by the last call
    bool s1_ok = false, s1_last = false;      // k_shard1024<.., G> usable (interleaved shard, G = 4, 8); used by the last call
    int s1_mfix = -1;                         // the shard's row among the four channels around DC (k_shard1024_dcfix), -1: none
    float2 *d_tw = nullptr, *d_wpre = nullptr;
    float2 *d_uhist[2] = {nullptr, nullptr}, *d_vend[2] = {nullptr, nullptr}, *d_rp[2] = {nullptr, nullptr};
    float2 *d_scratch = nullptr;     // yfirst | ylast
    void *d_full = nullptr;          // interleaved shard, calls k_run1024v2 does not take: whole-band result [1024][max_nf] (allocated on first use)
    void *d_premix = nullptr;
    float2 *d_cpre = nullptr, *d_side = nullptr, *d_rt = nullptr;       // k_run1024v3 without warm-up windows (Run1024v2Host::cpre / side / rt)
    int cur = 0;
};

bool big_supported(uint32_t M, uint32_t p) { return M == (uint32_t)PM && p == (uint32_t)PP; }

void big_destroy(BigPlan *p)
{
    if (!p) return;
    void *ptrs[] = {p->d_taps, p->d_taps_t, p->d_taps_q, p->d_tw, p->d_wpre, p->d_uhist[0], p->d_uhist[1], p->d_vend[0], p->d_vend[1], p->d_rp[0], p->d_rp[1],
                    p->d_scratch, p->d_premix, p->d_full, p->d_cpre, p->d_side, p->d_rt};
    for (void *q : ptrs) if (q) (void)hipFree(q);
    delete p;
}

int big_create(const FusedConfig &cfg, BigPlan **out)
{
    BigPlan *p = new BigPlan();
    p->cfg = cfg;
    auto fail = [&](int r) { big_destroy(p); return r; };
    {
        int dev = 0, cus = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        p->cus = (uint32_t)cus;
    }
#define ALLOC(ptr, bytes) do { hipError_t e = hipMalloc((void **)&(ptr), (bytes) ? (bytes) : 1); if (e != hipSuccess) return fail(hip_fail(e, "hipMalloc", __FILE__, __LINE__)); } while (0)
    ALLOC(p->d_taps, sizeof(float) * PM * PP);
    ALLOC(p->d_taps_t, sizeof(float) * PM * 16);
    ALLOC(p->d_taps_q, sizeof(float) * PM * 18);
    ALLOC(p->d_tw, sizeof(float2) * PM);
    ALLOC(p->d_wpre, sizeof(float2) * 2 * PM);
    for (int i = 0; i < 2; i++) {
        ALLOC(p->d_uhist[i], sizeof(float2) * 13 * PM);
        ALLOC(p->d_vend[i], sizeof(float2));
        ALLOC(p->d_rp[i], sizeof(float2) * (cfg.G > 1 ? (uint32_t)PM : cfg.C));     // interleaved shard: indexed by the primed channel k'
    }
    ALLOC(p->d_scratch, sizeof(float2) * 2 * (size_t)p->cus * PM);
    if (cfg.mix) ALLOC(p->d_premix, (size_t)cfg.C * cfg.max_nf * (cfg.fm ? 4 : 8));
#undef ALLOC
    CSDR_HIP(hipMemcpy(p->d_taps, cfg.taps, sizeof(float) * PM * PP, hipMemcpyHostToDevice));
    {
        std::vector<float> tt((size_t)PM * 16, 0.f);
        for (int j = 0; j < PM; j++) for (int n = 0; n < PP; n++) tt[(size_t)j * 16 + n] = cfg.taps[(PM - 1 - j) + n * PM];
        CSDR_HIP(hipMemcpy(p->d_taps_t, tt.data(), sizeof(float) * tt.size(), hipMemcpyHostToDevice));
    }
    std::vector<float2> tw(PM), wpre(2 * PM);
    for (int i = 0; i < PM; i++) {
        const double a = -2.0 * 3.14159265358979323846 * (double)i / (double)PM;
        tw[i] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    for (uint32_t i = 0; i < 2 * (uint32_t)PM; i++) {           // the NCO phase sequence has period 2M for a power-of-two M
        float c, sn;
        nco_phasor(i * cfg.d_theta, &c, &sn);
        wpre[i] = make_float2(c, -sn);
        if (cfg.G > 1) {
            // interleaved shard g = c0 of G: a shift of the whole spectrum by g channels, W1024^(r g) on branch r, rides on the branch's
            // pre-mix phasor for free (the FIR is linear): every kernel of the plan then sees the PRIMED spectrum Y'[k'] = Y[k' + g],
            // of which the shard owns the rows k' = 0 mod G
            const int n = (int)(((i & (uint32_t)(PM - 1)) * cfg.c0) & (uint32_t)(PM - 1));
            const double a = -2.0 * 3.14159265358979323846 * (double)n / (double)PM;
            const double wr = std::cos(a), wi = std::sin(a), xr = wpre[i].x, xi = wpre[i].y;
            wpre[i] = make_float2((float)(xr * wr - xi * wi), (float)(xr * wi + xi * wr));
        }
    }
    CSDR_HIP(hipMemcpy(p->d_tw, tw.data(), sizeof(float2) * tw.size(), hipMemcpyHostToDevice));
    CSDR_HIP(hipMemcpy(p->d_wpre, wpre.data(), sizeof(float2) * wpre.size(), hipMemcpyHostToDevice));
    {
        // k_run1024v2's table: [q][piece][j] float4 = floats 4 piece .. 4 piece + 3 of the row (14 taps, even-frame phasor) of
        // branch 256 q + j; behind it [q][j] float2: the odd-frame phasor (the f32 phases of the two are not exact negatives)
        std::vector<float> tq((size_t)PM * 16 + (size_t)PM * 2, 0.f);
        for (int r = 0; r < PM; r++) {
            const int q = r >> 8, j = r & 255;
            float row[16];
            for (int n = 0; n < PP; n++) row[n] = cfg.taps[(PM - 1 - r) + n * PM];
            row[14] = wpre[r].x; row[15] = wpre[r].y;
            for (int e = 0; e < 16; e++) tq[((size_t)(q * 4 + (e >> 2)) * 256 + j) * 4 + (e & 3)] = row[e];
            tq[(size_t)PM * 16 + 2 * (size_t)r] = wpre[PM + r].x; tq[(size_t)PM * 16 + 2 * (size_t)r + 1] = wpre[PM + r].y;
        }
        CSDR_HIP(hipMemcpy(p->d_taps_q, tq.data(), sizeof(float) * tq.size(), hipMemcpyHostToDevice));
        // k_run1024v2: the interleaved shards only (FM output); its whole-band instantiations are no longer built -- k_run1024v3 takes every
        // whole-band call of whole 4-frame tiles (a call that ends inside a 128-byte line stores the front part of it)
        p->v2_ok = false;
        // whole band, calls of whole 4-frame tiles: k_run1024v3 (CSDR_RUN1024_V3=0: k_run1024 for the comparison)
        p->v3_ok = cfg.c0 == 0 && cfg.C == (uint32_t)PM && cfg.G == 1 && !diag_env("CSDR_RUN1024_V1") && !(diag_env("CSDR_RUN1024_V3") && atoi(diag_env("CSDR_RUN1024_V3")) == 0);
        if (cfg.G > 1) p->v2_ok = cfg.fm && !diag_env("CSDR_RUN1024_V1");       // k_run1024v2<FM, G>; CF32 shards: whole band + row gather (below)
        // until the first call: the kernel a call of max_nf frames would take (csdr_chain_path names it)
        p->s1_ok = (cfg.G == 4 || cfg.G == 8) && !diag_env("CSDR_RUN1024_V1");      // k_shard1024<FM | CF32, G>: every run-sized call of whole tiles
        p->s1_last = p->s1_ok && shard1024_runs(cfg.max_nf, cfg.fm, cfg.G, p->cus) != 0;
        p->v2_last = !p->s1_last && p->v2_ok && (cfg.max_nf & 3u) == 0 && run1024_v2_runs(cfg.max_nf, p->cus) != 0;
        p->v3_last = p->v3_ok && run1024_v3_runs(cfg.max_nf, cfg.fm, p->cus) != 0;
        if (p->v3_last) p->v2_last = false;
    }
    if (p->s1_ok) {
        // k_shard1024 without warm-up windows (as k_run1024v3 below): the state hand-over between the runs, the side copies of the ONE owned channel
        // among the four around DC (510 .. 513; primed k' = k - g, row k' / G -- none for the shards 2 .. 5 of 8), and the chain's response to a
        // unit DC state there (computed on the plan's phasor table, which carries the shard's shift: the primed spectrum)
        p->s1_mfix = -1;
        uint32_t kp = 0;
        for (uint32_t k = 510; k <= 513; k++) if ((k + (uint32_t)PM - cfg.c0) % cfg.G == 0) { kp = (k + (uint32_t)PM - cfg.c0) % (uint32_t)PM; p->s1_mfix = (int)(kp / cfg.G); }
        hipError_t e1 = hipMalloc((void **)&p->d_cpre, sizeof(float2) * (p->cus + 2)), e2 = hipMalloc((void **)&p->d_side, sizeof(float2) * (size_t)(p->cus + 1) * RUN1024_DCFIX_F);
        if (e1 != hipSuccess || e2 != hipSuccess) return fail(hip_fail(e1 != hipSuccess ? e1 : e2, "hipMalloc", __FILE__, __LINE__));
        CSDR_HIP(hipMemset(p->d_cpre, 0, sizeof(float2) * (p->cus + 2)));
        if (cfg.dc_block && !(diag_env("CSDR_NOWU") && atoi(diag_env("CSDR_NOWU")) == 0)) {
            hipError_t e3 = hipMalloc((void **)&p->d_rt, sizeof(float2) * 2 * RUN1024_DCFIX_F * 4);
            if (e3 != hipSuccess) return fail(hip_fail(e3, "hipMalloc", __FILE__, __LINE__));
            std::vector<float2> rt((size_t)2 * RUN1024_DCFIX_F * 4, make_float2(0.f, 0.f));
            if (p->s1_mfix >= 0) dc_state_response(cfg, wpre.data(), 15u, (uint32_t)RUN1024_DCFIX_F, kp, rt.data());      // (channel 0 of the four is the owned one)
            CSDR_HIP(hipMemcpy(p->d_rt, rt.data(), sizeof(float2) * rt.size(), hipMemcpyHostToDevice));
        }
    }
    if (p->v3_ok) {
        // k_run1024v3's state hand-over and the side copies of the four channels around DC (510..513): the kernel always writes them (a few
        // hundred bytes per run); launches without warm-up windows (dc_block, not CSDR_NOWU=0) also get the chain's response to a unit DC state
        // at those channels, frames 15 .. 47 behind a cold start (= frame -1 .. 31 of the run)
        hipError_t e1 = hipMalloc((void **)&p->d_cpre, sizeof(float2) * (p->cus + 2)), e2 = hipMalloc((void **)&p->d_side, sizeof(float2) * (size_t)(p->cus + 1) * 4 * RUN1024_DCFIX_F);
        if (e1 != hipSuccess || e2 != hipSuccess) return fail(hip_fail(e1 != hipSuccess ? e1 : e2, "hipMalloc", __FILE__, __LINE__));
        CSDR_HIP(hipMemset(p->d_cpre, 0, sizeof(float2) * (p->cus + 2)));
    }
    if (p->v3_ok && cfg.dc_block && !(diag_env("CSDR_NOWU") && atoi(diag_env("CSDR_NOWU")) == 0)) {
        hipError_t e3 = hipMalloc((void **)&p->d_rt, sizeof(float2) * 2 * RUN1024_DCFIX_F * 4);
        if (e3 != hipSuccess) return fail(hip_fail(e3, "hipMalloc", __FILE__, __LINE__));
        std::vector<float2> rt((size_t)2 * RUN1024_DCFIX_F * 4);
        dc_state_response(cfg, wpre.data(), 15u, (uint32_t)RUN1024_DCFIX_F, 510u, rt.data());
        CSDR_HIP(hipMemcpy(p->d_rt, rt.data(), sizeof(float2) * rt.size(), hipMemcpyHostToDevice));
    }
    *out = p;
    return 0;
}

int big_reset(BigPlan *p, hipStream_t s)
{
    p->cur = 0; p->frames_done = 0;
    for (int i = 0; i < 2; i++) {
        CSDR_HIP(hipMemsetAsync(p->d_uhist[i], 0, sizeof(float2) * 13 * PM, s));
        CSDR_HIP(hipMemsetAsync(p->d_vend[i], 0, sizeof(float2), s));
        CSDR_HIP(hipMemsetAsync(p->d_rp[i], 0, sizeof(float2) * (p->cfg.G > 1 ? (uint32_t)PM : p->cfg.C), s));
    }
    return 0;
}

void big_seek(BigPlan *p, uint64_t frames) { p->frames_done = frames; }
const char *big_name(const BigPlan *p)
{
    if (p->v3_last) return p->cfg.fm ? "k_run1024v3<FM>" : "k_run1024v3<CF32>";
    if (p->s1_last) return p->cfg.fm ? (p->cfg.G == 8 ? "k_shard1024<FM>/G8" : "k_shard1024<FM>/G4") : (p->cfg.G == 8 ? "k_shard1024<CF32>/G8" : "k_shard1024<CF32>/G4");
    if (p->v2_last) return p->cfg.G == 8 ? "k_run1024v2<FM>/G8" : p->cfg.G == 4 ? "k_run1024v2<FM>/G4" : p->cfg.G == 2 ? "k_run1024v2<FM>/G2" : "k_run1024v2<FM>";
    return p->cfg.fm ? "k_run1024<FM>" : "k_run1024<CF32>";
}

bool big_tile_major_ok(const BigPlan *p, uint32_t nf)
{
    if (p && p->s1_ok) return !p->cfg.fm && !p->cfg.mix && nf % 16u == 0 && shard1024_runs(nf, false, p->cfg.G, p->cus) != 0;      // interleaved shard: k_shard1024<CF32, G>
    return p && p->v3_ok && !p->cfg.fm && !p->cfg.mix && nf % 16u == 0 && (uint64_t)nf * 8192u < (1ull << 31) && run1024_v3_runs(nf, false, p->cus) != 0;
}

int big_process(BigPlan *p, const FusedCall &call, hipStream_t s, KernelTimer *timer)
{
    const FusedConfig &c = p->cfg;
    const uint32_t nf = call.nf;
    if (!nf) return 0;
    int r;
    const uint32_t s1runs = p->s1_ok ? shard1024_runs(nf, c.fm, c.G, p->cus) : 0;
    p->s1_last = s1runs != 0;
    const uint32_t v2runs = (!s1runs && p->v2_ok && (nf & 3u) == 0 && (uint64_t)nf * 8192u < (1ull << 31)) ? run1024_v2_runs(nf, p->cus) : 0;
    p->v2_last = v2runs != 0;
    if (call.tile_major && !big_tile_major_ok(p, nf)) { set_error("big_process: tile-major output asked for a call k_run1024v3<CF32> does not take"); return CSDR_ERR_INVALID; }
    const uint32_t v3runs = (p->v3_ok && (uint64_t)nf * 8192u < (1ull << 31)) ? run1024_v3_runs(nf, c.fm, p->cus) : 0;
    p->v3_last = v3runs != 0;
    if (v3runs) p->v2_last = false;
    if (v2runs || v3runs || s1runs) {
        Run1024v2Host H{};
        H.x = call.d_in; H.out = c.mix ? p->d_premix : call.d_out; H.taps_q = p->d_taps_q; H.tw = p->d_tw;
        H.uhist_in = p->d_uhist[p->cur]; H.uhist_out = p->d_uhist[p->cur ^ 1];
        H.vend_in = p->d_vend[p->cur]; H.vend_out = p->d_vend[p->cur ^ 1];
        H.rp_in = p->d_rp[p->cur]; H.rp_out = p->d_rp[p->cur ^ 1];
        H.nf = nf; H.nruns = v2runs; H.parity0 = (uint32_t)(p->frames_done & 1);
        H.G = c.G; H.g = c.G > 1 ? c.c0 : 0u;
        H.tile_major = call.tile_major && (v3runs || s1runs) && !c.fm;
        H.dc_block = c.dc_block; H.beta = c.dc_block ? (double)c.dc.beta : 0.0; H.fm_ref = c.fm_ref;
        if (v3runs) { H.cpre = p->d_cpre; H.side = p->d_side; H.rt = v3runs <= p->cus ? p->d_rt : nullptr; }
        if (s1runs) { H.cpre = p->d_cpre; H.side = p->d_side; H.rt = p->d_rt; H.mfix = p->s1_mfix; }
        if (s1runs) { if ((r = shard1024_launch(H, c.fm, s1runs, s, timer))) return r; }
        else if (v3runs) { if ((r = run1024_v3_launch(H, c.fm, v3runs, s, timer))) return r; }
        else if ((r = run1024_v2_launch(H, c.fm, s, timer))) return r;
        p->cur ^= 1;
        p->frames_done += nf;
        if (c.mix) {
            if ((r = launch_mix((const float *)p->d_premix, (float *)call.d_out, c.C, c.fm ? nf : 2 * nf, s))) return r;
        }
        return 0;
    }
    Pfb1024Args A{};
    A.u = call.d_in; A.taps = p->d_taps; A.taps_t = p->d_taps_t; A.tw = p->d_tw; A.wpre = p->d_wpre;
    A.out = c.mix ? p->d_premix : call.d_out;
    const bool shard = c.G > 1;
    if (shard) {
        // an interleaved shard's call that k_run1024v2<FM, G> does not take (CF32 output, ragged or short call): the whole primed
        // band through k_run1024 into scratch, then the owned rows k1' = 0 mod G are gathered into the shard's plane
        if (!p->d_full) { hipError_t e = hipMalloc(&p->d_full, (size_t)PM * c.max_nf * (c.fm ? 4 : 8)); if (e != hipSuccess) return hip_fail(e, "hipMalloc", __FILE__, __LINE__); }
        A.out = p->d_full;
    }
    A.rp_in = p->d_rp[p->cur]; A.rp_out = p->d_rp[p->cur ^ 1];
    A.vend_in = p->d_vend[p->cur]; A.vend_out = p->d_vend[p->cur ^ 1];
    A.uhist_in = p->d_uhist[p->cur]; A.uhist_out = p->d_uhist[p->cur ^ 1];
    A.nf = nf; A.nb = (nf + PT - 1) / PT; A.c0 = shard ? 0u : c.c0; A.C = shard ? (uint32_t)PM : c.C; A.ref = c.fm_ref;
    A.parity0 = (uint32_t)(p->frames_done & 1);
    const double beta = c.dc_block ? (double)c.dc.beta : 0.0;
    A.alpha = c.dc_block ? (float)(1.0 - beta) : 0.0f;
    A.l2beta = c.dc_block ? (float)std::log2(beta) : -1000.0f;
    for (int k = 0; k < 4; k++) { A.bp[k] = (float)std::pow(beta, (double)(1 << k)); A.dp[k] = (float)std::pow(beta, 16.0 * (1 << k)); }
    A.dm = (float)std::pow(beta, 1024.0);
    A.pk = phase_consts(c.fm_ref);
    // one run per CU, at least 8 tiles per run (a run >= 1 spends 3 read-only + 2 halo tiles on its start state)
    uint32_t nruns = p->cus;
    if (nruns > A.nb / 8) nruns = A.nb / 8;
    if (nruns < 1) nruns = 1;
    A.nruns = nruns; A.yfirst = p->d_scratch; A.ylast = p->d_scratch + (size_t)nruns * PM;
    if (timer && (r = timer->begin(s))) return r;
    if (c.fm) hipLaunchKernelGGL((k_pfb1024<true, true>), dim3(nruns), dim3(1024), 0, s, A);
    else hipLaunchKernelGGL((k_pfb1024<false, true>), dim3(nruns), dim3(1024), 0, s, A);
    if (timer && (r = timer->end(s))) return r;
    if (c.fm && nruns > 1) hipLaunchKernelGGL(k_pfb1024_fixup, dim3(nruns - 1), dim3(1024), 0, s, A);
    if (shard) hipLaunchKernelGGL(k_shard_gather1024, dim3(c.C), dim3(256), 0, s, (const float *)p->d_full, (float *)(c.mix ? p->d_premix : call.d_out), c.G, nf, c.fm ? 1u : 2u);
    CSDR_HIP(hipGetLastError());
    p->cur ^= 1;
    p->frames_done += nf;
    if (c.mix) {
        if ((r = launch_mix((const float *)p->d_premix, (float *)call.d_out, c.C, c.fm ? nf : 2 * nf, s))) return r;
    }
    return 0;
}

}  // namespace csdr
