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

namespace csdr {

namespace {

constexpr int PM = 1024, PT = 8, PP = 14;       // channels, frames per tile, taps per branch
constexpr int PAS = 17 * 4;                     // padded stride between k1 rows of the pass-1 image (as k_fft_r16<4>)

struct Pfb1024Args {
    const float2 *u;            // first NEW sample; 13 * 1024 samples of history in front
    const float *taps;          // h[(1023 - j) + 1024 n]
    const float2 *tw;           // e^{-j 2 pi i / 1024}
    void *out;                  // [C][nf] F32 (FM) or CF32
    const float2 *rp_in; float2 *rp_out;        // [C] freqdem r'
    float2 *yfirst, *ylast;     // [nruns][1024]
    uint32_t nf, nb, nruns, c0, C;
    float ref;
};

template <bool FM>
__global__ __launch_bounds__(1024) void k_pfb1024(Pfb1024Args A)
{
    __shared__ float2 bufA[PT * 16 * PAS];      // 69 632 B
    __shared__ float2 bufB[PT * PM];            // 65 536 B
    __shared__ float2 tw1[16 * 64];             // pass-1 twiddles W1024^(m k1) at [k1][m]
    __shared__ float2 tw2[16 * 4];              // pass-2 twiddles W1024^(16 n3 k2) at [k2][n3]
    const int tid = threadIdx.x, j = tid;
    { const int k1 = tid >> 6, m = tid & 63; tw1[tid] = A.tw[(m * k1) & 1023]; }
    if (tid < 64) { const int k2 = tid >> 2, n3 = tid & 3; tw2[tid] = A.tw[(16 * n3 * k2) & 1023]; }
    const uint32_t w = blockIdx.x;
    const uint32_t first = (uint32_t)((uint64_t)w * A.nb / A.nruns), last = (uint32_t)((uint64_t)(w + 1) * A.nb / A.nruns);
    if (first >= last) return;

    float2 old[13];
    {
        const int64_t fa = (int64_t)first * PT;
#pragma unroll
        for (int i = 0; i < 13; i++) old[i] = A.u[(fa - 13 + i) * PM + j];          // reaches into the history for run 0
    }
    const bool owned = (uint32_t)tid >= A.c0 && (uint32_t)tid < A.c0 + A.C;
    float2 prev = (FM && w == 0 && owned) ? A.rp_in[tid - A.c0] : make_float2(0.f, 0.f);

    float2 nw[PT];
    {
        const uint32_t t0 = first * PT;
#pragma unroll
        for (int f = 0; f < PT; f++) nw[f] = t0 + f < A.nf ? A.u[(size_t)(t0 + f) * PM + j] : make_float2(0.f, 0.f);
    }
    __syncthreads();                                                // twiddle tables
    for (uint32_t b = first; b < last; b++) {
        // keep the per-phase address arithmetic inside the iteration (hoisted out of the tile loop it pins dozens of VGPRs)
        int tid_i = tid;
        asm volatile("" : "+v"(tid_i));
        const int j_i = tid_i;
        const uint32_t t0 = b * PT;
        const int nvalid = (int)min((uint32_t)PT, A.nf - t0);
        // ---- polyphase FIR, oldest tap first; four frames at a time (independent accumulators).  The 14 taps are
        // re-read per tile (L2-resident, 56 KiB for the whole bank): held across the tile loop they would spill ----
        float h[PP];
#pragma unroll
        for (int n = 0; n < PP; n++) h[n] = A.taps[(PM - 1 - j_i) + n * PM];
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
        // next tile's window: the last 13 of (old | nw)
#pragma unroll
        for (int i = 0; i < 13; i++) old[i] = (i + PT < 13) ? old[i + PT] : nw[i + PT - 13];
        // the next tile's samples fly in during the DFT and the tail
        if (b + 1 < last) {
            const uint32_t t1 = t0 + PT;
#pragma unroll
            for (int f = 0; f < PT; f++) nw[f] = t1 + f < A.nf ? A.u[(size_t)(t1 + f) * PM + j_i] : make_float2(0.f, 0.f);
        }
        __syncthreads();                                            // X complete (in bufB)

        v2fg v[16];
        // ---- pass 1: radix 16 over n1 (stride 64) for (frame, m): threads 0..511 ----
        if (tid_i < 512) {
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
        if (tid_i < 512) {
            const int n3 = tid_i & 3, k1 = (tid_i >> 2) & 15, fr = tid_i >> 6;
#pragma unroll
            for (int n2 = 0; n2 < 16; n2++) { const float2 x = bufA[fr * 16 * PAS + k1 * PAS + 4 * n2 + n3]; v[n2] = (v2fg){x.x, x.y}; }
            g_fft16(v);
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int k2 = GXIDX(i);
                if (k2 && n3) { const float2 tq = tw2[k2 * 4 + n3]; v[i] = g_cmul(v[i], (v2fg){tq.x, tq.y}); }
                bufB[fr * PM + (k1 + 16 * k2) * 4 + n3] = make_float2(v[i].x, v[i].y);
            }
        }
        __syncthreads();
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
                    for (int f = 0; f < 4; f++) m[f] = fm_sample_rn(f ? y[f - 1] : prev, y[f], A.ref);
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
        __syncthreads();                                            // Y consumed before the next tile's X
    }
    if (FM) {
        A.ylast[(size_t)w * PM + tid] = prev;
        if (w + 1 == A.nruns && owned) A.rp_out[tid - A.c0] = prev;
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
    if (fm) hipLaunchKernelGGL(k_pfb1024<true>, dim3(nruns), dim3(1024), 0, s, A);
    else hipLaunchKernelGGL(k_pfb1024<false>, dim3(nruns), dim3(1024), 0, s, A);
    if (fm && nruns > 1) hipLaunchKernelGGL(k_pfb1024_fixup, dim3(nruns - 1), dim3(1024), 0, s, A);
    CSDR_HIP(hipGetLastError());
    return 0;
}

}  // namespace csdr
