// Fused M = 4096 chain (round 5; SURVEY section 7 step 5, BASELINE configs[4]'s channel count with per-channel outputs):
//
//   raw CF32 x --DC blocker--> y --NCO pre-mix, 14-tap polyphase FIR--> X_t[j] --radix-4 split over j = j1 + 1024 q--> z_t[r][j1]
//                                                                                                   (k_front4096, 8 B read + 8 B written)
//   z_t[r][.] --1024-point forward DFT (16 x 16 x 4)--> Y_t[4 k' + r] --[freqdem]--> out[4096][nf] | per-frame partial mixes
//                                                                                                   (k_back4096, 8 B read + 8 / 4 / ~0 B written)
//   (Liquid.chs:575-589 dcBlocker, :846-847 pre-mix, :843 analyzer_execute, :840-862 transpose, :324-328 freqdem; Trans.hs:119-122 mix)
//
// Why two kernels.  A 13-frame FIR window of 4096 branches is 416 KiB: more than a CU's LDS (160 KiB), and with the rest of a kernel's
// registers more than its VGPR file (512 KiB).  So the filter bank is tiled by BRANCH: a front workgroup owns the branches
// j1 + 1024 q, j1 in [256 c, 256 c + 256), q = 0..3 (window: 13 frames x 4 branches per thread = 104 VGPRs), four sibling workgroups
// c = 0..3 cover a frame.  The DC blocker is a scan over the WHOLE stream, so every sibling DMA's the whole frame (a frame is exactly
// one 4096-sample tile of the shared DC machinery; the four reads of a frame meet in the XCD's L2: siblings get workgroup ids 8 apart)
// and scans it -- in full only for the four 256-sample groups it owns (one wave), totals only for the other twelve (three waves).
// Owning q = 0..3 of the same j1 lets the thread do the first (decimation-in-frequency) radix-4 stage of the 4096-point DFT on its own
// registers:  Y[4 k' + r] = sum_j1 z_r[j1] W1024^(j1 k'),  z_r[j1] = W4096^(j1 r) sum_q X[j1 + 1024 q] W4^(q r),
// which cuts the transform into four independent 1024-point DFTs.  That is what makes the back half fit: a back workgroup takes ONE
// residue r and 16 consecutive frames (1024 channels x 16 frames x 8 B = 128 KiB, in LDS), so every output row gets a whole 128-byte
// line (CF32) per workgroup.  Between the two kernels only z (8 B per sample) goes through memory.
//
// Run structure (front): a run = consecutive frames of one sibling group; it starts COLD 20 frames early -- 7 (run 0: 6) frames of group totals
// only (DC state, beta^24576 = 4.6e-6 of anything older, the cut every run kernel makes), then 13 frames that refill the window without
// FIR.  Run 0 of a call replays the previous call's last raw frames (kept by the plan; zeros after create / reset = the reference's initial
// state), and computes frame -1 in full: z slot 0 is the freqdem history of the call's first frame.  No other state crosses calls.
#include "fused_v2_common.h"
#include <string>
#include <type_traits>

#ifndef H_ABLATE
#define H_ABLATE 0       // timing experiments only: 1 no z stores, 2 one FIR tap, 4 no y' pass in the scan (totals only), 8 no tile DMA in the loop, 16 no group-carry scan in the filter
#endif

namespace csdr {
namespace {

// cold start of a run: >= 6 frames of group totals only (DC state: beta^24576), then the 13 window frames (14 for run 0 of a call, whose
// frame -1 is computed in full)
constexpr int H_M = 4096, H_WU = 7, H_WIN = 13, H_COLD = H_WU + H_WIN;
constexpr int H_BUF = 4096;                        // float2 per frame image
constexpr int H_NB = 4;                            // image ring: the frame being scanned + three in flight
constexpr int H_YR = H_NB * H_BUF;                 // y' ring: 2 x (4 owned groups x 256) float2 (the filter never touches the images)
constexpr int H_TT = H_YR + 2 * 1024;              // 2 x 16 group totals
constexpr int H_WO = H_TT + 32;                    // odd-frame pre-mix phasors of the workgroup's 1024 branches (the even ones stay in registers)
constexpr int H_F2 = H_WO + 1024;                  // 155 904 B: one 512-thread workgroup per CU

struct Front4096Args {
    const float4 *x;            // raw input of this call: frame t at x + 2048 t
    const float4 *tail;         // the previous call's last H_COLD raw frames: frame -k at tail + 2048 (H_COLD - k)
    float2 *z;                  // [(nf + 1)][4][1024]: slot t + 1 = frame t; slot 0 = frame -1
    const float *taps;          // [14][4096] prototype taps h[i + 4096 n]
    const float2 *wpre;         // [2][4096] conj(nco phasor) of branch j at even / odd global frames
    const float2 *tw4;          // [3][1024] W4096^(j1 r), r = 1..3
    uint32_t nf, nruns, parity0;
    float alpha, beta, l2beta;
    float b16[16];              // beta^(16 r)
    float b256[17];             // beta^(256 g)
};

// 512 threads, one workgroup per CU, TWO WAVE ROLES a frame apart (the structure of k_run1024v3):
//   waves 0-3, the scan quad: tile DMA three frames ahead into a ring of four images, the DC scan of frame k -- group totals of all
//       sixteen 256-sample groups, y' (x - alpha s, group carry still missing) of the four groups this workgroup owns into a small ring;
//   waves 4-7, the filter quad (thread = j1, branches 256 (c + 4 q) + j1, q = 0..3): group carries of frame k - 1 out of its totals,
//       pre-mix, 14-tap FIR out of a 13-frame register window (104 VGPRs; taps 56), radix 4 over q + twiddle, z stores.
// One workgroup barrier per frame.  The scan quad's vmcnt queue holds the DMA and nothing else (s_waitcnt vmcnt(16) = "all but the two
// youngest frames have landed"); the filter quad's only VMEM are its stores, which nothing waits for.
__global__ __launch_bounds__(512, 1) void k_front4096(Front4096Args A)
{
    __shared__ __attribute__((aligned(16))) float2 L[H_F2];
    float2 *YR = L + H_YR, *Tt = L + H_TT, *WO = L + H_WO;
    const int tid = threadIdx.x;
    // siblings (the four c of a run) sit 8 workgroup ids apart: same XCD (id mod 8), dispatched within 32 ids of each other
    const unsigned w = blockIdx.x;
    unsigned c, run;
    if ((A.nruns & 7u) == 0) { c = (w >> 3) & 3u; run = (w & 7u) + 8u * (w >> 5); }
    else { c = w & 3u; run = w >> 2; }
    const int first = (int)((unsigned long long)run * A.nf / A.nruns), last = (int)((unsigned long long)(run + 1) * A.nf / A.nruns);
    if (first >= last) return;
    const int full_from = first - (run == 0 ? 1 : 0), t_begin = first - H_COLD, n = last - t_begin;
    auto tile_ptr = [&](int t) -> const float4 * { return t >= 0 ? A.x + (size_t)t * 2048 : A.tail + (size_t)(H_COLD + t) * 2048; };
    auto mode_of = [&](int t) -> int { return t < full_from - H_WIN ? 0 : (t < full_from ? 1 : 2); };   // 0: group totals only, 1: window refill, 2: FIR + z
    const unsigned wave_u = (unsigned)__builtin_amdgcn_readfirstlane(tid >> 6);

    if (wave_u < 4u) {
        // ================================================================== scan quad
        const unsigned goff = dma_offset(tid);
        const unsigned lds_wave = (unsigned)(size_t)(__attribute__((address_space(3))) float2 *)L + 1024u * wave_u;
        // my run of 16 consecutive samples: wave 0 takes the four groups this workgroup owns (full scan), waves 1..3 the other twelve (totals)
        const int row = (tid >> 4) & 3, r16 = tid & 15;
        int grp;
        if (wave_u == 0) grp = (int)c + 4 * row;
        else { const int m = ((int)wave_u - 1) * 4 + row; grp = (m / 3) * 4 + (((int)c + 1 + m % 3) & 3); }
        const unsigned raw_a = (unsigned)(16 * grp + r16) * 128u + ((unsigned)((r16 >> 1) & 7) << 4);
        const unsigned y_a = (unsigned)(16 * row + r16) * 128u + ((unsigned)((r16 >> 1) & 7) << 4);   // the same run inside the y' ring's 8 KiB
        const float na = opaque_v(-A.alpha), be = opaque_v(A.beta);
#pragma unroll
        for (int d = 0; d < 3; d++) if (d < n) dma_tile(tile_ptr(t_begin + d), goff, lds_wave + (unsigned)d * (H_BUF * 8u));
        if (n > 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else if (n > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bar();                                          // frame 0 has landed
        for (int k = 0; k <= n; k++) {
            if (k < n) {
                if (!(H_ABLATE & 8) && k + 3 < n) dma_tile(tile_ptr(t_begin + k + 3), goff, lds_wave + (unsigned)((k + 3) & 3) * (H_BUF * 8u));
                const char *B = reinterpret_cast<const char *>(L) + (k & 3) * (H_BUF * 8);
                v4f xr[8];
                float2 s = make_float2(0.f, 0.f);
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    xr[i] = *reinterpret_cast<const v4f *>(B + (raw_a ^ (unsigned)(i << 4)));
                    s = make_float2(fmaf(s.x, be, xr[i].x), fmaf(s.y, be, xr[i].y));
                    s = make_float2(fmaf(s.x, be, xr[i].z), fmaf(s.y, be, xr[i].w));
                }
                float2 u;
                u = dpp2<0x111>(s); s = cfma(u, A.b16[1], s);
                u = dpp2<0x112>(s); s = cfma(u, A.b16[2], s);
                u = dpp2<0x114>(s); s = cfma(u, A.b16[4], s);
                u = dpp2<0x118>(s); s = cfma(u, A.b16[8], s);
                if (r16 == 15) Tt[16 * (k & 1) + grp] = s;
                if (!(H_ABLATE & 4) && wave_u == 0 && mode_of(t_begin + k)) {
                    char *Y = reinterpret_cast<char *>(YR) + (k & 1) * 8192;
                    s = dpp2<0x111>(s);
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        v4f y;
                        y.x = fmaf(s.x, na, xr[i].x); y.y = fmaf(s.y, na, xr[i].y);
                        s = make_float2(fmaf(s.x, be, xr[i].x), fmaf(s.y, be, xr[i].y));
                        y.z = fmaf(s.x, na, xr[i].z); y.w = fmaf(s.y, na, xr[i].w);
                        s = make_float2(fmaf(s.x, be, xr[i].z), fmaf(s.y, be, xr[i].w));
                        *reinterpret_cast<v4f *>(Y + (y_a ^ (unsigned)(i << 4))) = y;
                    }
                }
                // frame k + 1 has landed before the barrier lets anyone scan it
                if (H_ABLATE & 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (k + 3 < n) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else if (k + 2 < n) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            bar();
        }
        return;
    }
    // ====================================================================== filter quad
    const int j = tid & 255;                                               // j1 - 256 c: position inside a 256-sample group
    const int col_off = 16 * (j >> 4) + 2 * (((j & 15) >> 1) ^ (j >> 5)) + (j & 1);
    float h[4][P];
    float2 we[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int br = 256 * (int)(c + 4u * q) + j;
#pragma unroll
        for (int m = 0; m < P; m++) h[q][m] = A.taps[(H_M - 1 - br) + m * H_M];
        we[q] = A.wpre[br]; WO[256 * q + j] = A.wpre[H_M + br];            // (a thread reads back only what it wrote: no barrier needed)
    }
    float2 t4[3];
#pragma unroll
    for (int r = 0; r < 3; r++) t4[r] = A.tw4[r * 1024 + 256 * (int)c + j];
    float2 hist[H_WIN * 4];                                                // window: frames -13 .. -1 of my four branches, [13][4]
#pragma unroll
    for (int i = 0; i < H_WIN * 4; i++) hist[i] = make_float2(0.f, 0.f);
    float2 cst = make_float2(0.f, 0.f);                                    // DC state v before the next frame (same in every lane)
    const float kJ = -A.alpha * exp2f((float)j * A.l2beta);                // -alpha beta^j: group state into position j of a group
    const float bg = A.b256[tid & 15], b1 = A.b256[1], b2 = A.b256[2], b4 = A.b256[4], b8 = A.b256[8], b16g = A.b256[16];
    bar();                                              // (the scan quad's "frame 0 has landed")
    bar();                                              // (its step 0: frame 0 is being scanned)
    // The window is a RING of 13 frames: the frame that arrives in step k goes to slot (k - 1) mod 13 -- over the frame 13 back, which
    // the FIR has just read -- and the loop is unrolled 13 times so that every slot index is a constant (a moving window costs 52
    // 64-bit moves per frame and thread: more than the FIR itself).
    auto step = [&](const int k, auto SC) {
        constexpr int SL = decltype(SC)::value;
        const int t = t_begin + k - 1, mode = mode_of(t);
        // ---- group carries: lane g of every 16-lane row scans the sixteen group totals (decayed, inclusive); the state before
        // group g is cst beta^(256 g) + S[g - 1]; my four groups c + 4 q read theirs with uniform lane indices
        float2 S = Tt[16 * ((k - 1) & 1) + (tid & 15)], u;
        u = dpp2<0x111>(S); S = cfma(u, b1, S);
        u = dpp2<0x112>(S); S = cfma(u, b2, S);
        u = dpp2<0x114>(S); S = cfma(u, b4, S);
        u = dpp2<0x118>(S); S = cfma(u, b8, S);
        const float2 Vl = cfma(cst, bg, dpp2<0x111>(S));                  // lane g: state before group g (row_shr:1 leaves lane 0 at zero)
        v2f Vq[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int ln = (int)c + 4 * q;
            Vq[q] = (v2f){__int_as_float(__builtin_amdgcn_readlane(__float_as_int(Vl.x), ln)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Vl.y), ln))};
        }
        cst = cfma(cst, b16g, make_float2(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(S.x), 15)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(S.y), 15))));
        if (mode) {
            // ---- my four samples of the frame, group carry, pre-mix (phasor of the frame's global parity)
            const float2 *Yf = YR + ((k - 1) & 1) * 1024;
            const bool odd = ((A.parity0 + (unsigned)(t & 1)) & 1u) != 0;             // (t may be negative: & 1 is its parity either way)
            v2f nw[4];
            const v2f kJv = {kJ, kJ};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const v2f yv = __builtin_elementwise_fma(Vq[q], kJv, to_v(Yf[256 * q + col_off]));
                nw[q] = cmul_v(yv, to_v(odd ? WO[256 * q + j] : we[q]));
            }
            if (mode == 2) {
                // ---- polyphase FIR on the pre-mixed window, then the first DFT stage: radix 4 over q and the twiddle W4096^(j1 r)
                v2f X[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    v2f acc = {0.f, 0.f};
#pragma unroll
                    for (int m = ((H_ABLATE & 2) ? 1 : P - 1); m >= 1; m--) {
                        const float2 s2 = hist[4 * ((SL - m + 2 * H_WIN) % H_WIN) + q];       // frame -m
                        acc = __builtin_elementwise_fma((v2f){s2.x, s2.y}, (v2f){h[q][m], h[q][m]}, acc);
                    }
                    X[q] = __builtin_elementwise_fma(nw[q], (v2f){h[q][0], h[q][0]}, acc);
                }
                bfly4_v(X[0], X[1], X[2], X[3]);        // X[r] = sum_q X[q] (-j)^(q r)
#pragma unroll
                for (int r = 1; r < 4; r++) X[r] = cmul_v(X[r], to_v(t4[r - 1]));
                float2 *zp = A.z + ((size_t)(t + 1) * 4) * 1024 + 256 * (int)c + j;
                if (H_ABLATE & 1) asm volatile("" :: "v"(X[0]), "v"(X[1]), "v"(X[2]), "v"(X[3]), "v"(zp));
                else {
#pragma unroll
                    for (int r = 0; r < 4; r++) zp[r * 1024] = to_f2(X[r]);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; q++) hist[4 * SL + q] = to_f2(nw[q]);
        }
        bar();
    };
    for (int kb = 1; kb <= n; kb += H_WIN) {
#define H_STEP(i) if (kb + (i) > n) break; step(kb + (i), std::integral_constant<int, (i)>{});
        H_STEP(0) H_STEP(1) H_STEP(2) H_STEP(3) H_STEP(4) H_STEP(5) H_STEP(6) H_STEP(7) H_STEP(8) H_STEP(9) H_STEP(10) H_STEP(11) H_STEP(12)
#undef H_STEP
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// back half: residue r, block of 16 frames.  1024 threads: wave i = frame 16 blk + i (its 8.5 KiB region of LDS is private to the wave
// through passes 1-3: no barrier); the frame in front of the block (freqdem history) is an extra round of wave 0.
constexpr int K_AS = 68;                           // padded stride between the 16 k1 rows of the pass-1 image (float2)
constexpr int K_FR = 16 * K_AS;                    // 1088 float2 per frame region
constexpr int K_ROWB = 136;                        // bytes per transposed row: 16 frames + the history frame, 8 B each (34 dwords: bank-spread)
constexpr int K_TW = 17 * K_FR;                    // twiddle table behind the 17 frame regions
constexpr int K_F2 = K_TW + 1024;                  // 19 520 float2 = 156 160 B

struct Back4096Args {
    const float2 *z;            // [(nf + 1)][4][1024]
    void *out;                  // MODE 0: [4096][nf] CF32; 1: [4096][nf] F32; 2: partial mixes [nblk][4][16] F32
    const float2 *tw;           // W1024^i
    uint32_t nf, out_stride;
    uint32_t tile_major;        // MODE 0: the 128-byte lines of a 16-frame block back to back, [block][4096][128 B] (the plane k_agc_spec_tm reads); nf % 16 == 0
    PhaseK pk; float fm_ref, tiny;
};

template <int MODE>
__global__ __launch_bounds__(1024, 1) void k_back4096(Back4096Args A)
{
    __shared__ __attribute__((aligned(16))) float2 L[K_F2];
    constexpr bool FM = MODE != 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned wv = (unsigned)__builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned r = blockIdx.x & 3u, blk = blockIdx.x >> 2;
    const int nfb = (int)min(16u, A.nf - 16u * blk);                       // frames of this block
    float2 *tw = L + K_TW;
    tw[tid] = A.tw[tid];
    __syncthreads();
    v2f y[4][4];                                                           // pass 3 results of my frame: y[u][k3] = Y[lane + 64 u + 256 k3]
    float2 yh[4][4];                                                       // wave 0, FM: the same for the history frame
    auto dft_frame = [&](unsigned fi, v2f (&Y)[4][4]) {                    // fi: 0..15 = frames of the block, 16 = the frame in front of it
        float2 *F = L + (size_t)fi * K_FR;
        const unsigned slot = 16u * blk + (fi == 16u ? 0u : fi + 1u);
        const float2 *src = A.z + ((size_t)slot * 4 + r) * 1024;
        v2f v[16];
        // pass 1: radix 16 over n1 (stride 64), lane = m
#pragma unroll
        for (int n1 = 0; n1 < 16; n1++) v[n1] = to_v(src[64 * n1 + lane]);
        fft16_v(v);
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int k1 = XIDX(i);
            if (k1) v[i] = cmul_v(v[i], to_v(tw[(lane * k1) & 1023]));
            F[k1 * K_AS + lane] = to_f2(v[i]);
        }
        // pass 2: radix 16 over n2 (stride 4), lane = (k1, n3)
        const int k1 = lane >> 2, n3 = lane & 3;
#pragma unroll
        for (int n2 = 0; n2 < 16; n2++) v[n2] = to_v(F[k1 * K_AS + 4 * n2 + n3]);
        fft16_v(v);
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int k2 = XIDX(i);
            if (k2) {
                const v2f t = cmul_v(v[i], to_v(tw[(16 * n3 * k2) & 1023]));          // (unconditional load + select: kernels_generic.hip k_fft_r16)
                v[i] = n3 ? t : v[i];
            }
        }
#pragma unroll
        for (int i = 0; i < 16; i++) F[4 * (k1 + 16 * XIDX(i)) + n3] = to_f2(v[i]);
        // pass 3: radix 4 over n3 for kk = lane + 64 u
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const v4f a = *reinterpret_cast<const v4f *>(F + 4 * (lane + 64 * u)), b = *reinterpret_cast<const v4f *>(F + 4 * (lane + 64 * u) + 2);
            Y[u][0] = (v2f){a.x, a.y}; Y[u][1] = (v2f){a.z, a.w}; Y[u][2] = (v2f){b.x, b.y}; Y[u][3] = (v2f){b.z, b.w};
            bfly4_v(Y[u][0], Y[u][1], Y[u][2], Y[u][3]);
        }
    };
    const bool live = (int)wv < nfb;
    if (live) dft_frame(wv, y);
    else {
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int k = 0; k < 4; k++) y[u][k] = (v2f){0.f, 0.f};
    }
    if (FM && wv == 0) {
        v2f t[4][4];
        dft_frame(16u, t);
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int k = 0; k < 4; k++) yh[u][k] = to_f2(t[u][k]);
    }
    __syncthreads();                                    // every frame region has been read out
    // ---- transposed image: row k' = 136 bytes = frames 0..15 of the block, then the history frame
    char *T = reinterpret_cast<char *>(L);
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int k3 = 0; k3 < 4; k3++) {
            const int kp = lane + 64 * u + 256 * k3;
            *reinterpret_cast<float2 *>(T + kp * K_ROWB + 8 * (int)wv) = to_f2(y[u][k3]);
            if (FM && wv == 0) *reinterpret_cast<float2 *>(T + kp * K_ROWB + 128) = yh[u][k3];
        }
    __syncthreads();
    const uint32_t esz = FM ? 4u : 8u;
    const size_t col0 = (size_t)16 * blk;
    if (FM) {
        // ---- freqdem: thread = row k' (its wave's 64 rows are private to the wave from here on)
        char *R = T + tid * K_ROWB;
        float2 prev = *reinterpret_cast<const float2 *>(R + 128);
        const FmK2 fk = {{A.pk.c[0], A.pk.c[1], A.pk.c[2], A.pk.c[3], A.pk.c[4], A.pk.c[5], A.pk.c[6], A.pk.c[7]}, A.tiny, A.fm_ref, A.pk.hp, A.pk.pi};
        float m[16];
#pragma unroll
        for (int f0 = 0; f0 < 16; f0 += 4) {
            float2 rr[4], rp[4];
#pragma unroll
            for (int u = 0; u < 4; u++) rr[u] = *reinterpret_cast<const float2 *>(R + 8 * (f0 + u));
            rp[0] = prev; rp[1] = rr[0]; rp[2] = rr[1]; rp[3] = rr[2];
            prev = rr[3];
            fm_quad(rp, rr, fk, *reinterpret_cast<float (*)[4]>(&m[f0]));
        }
        if (MODE == 2) {
            // ---- per-frame mix over this residue's 1024 channels: ascending k' inside 64 segments of 16 rows, then over the segments
#pragma unroll
            for (int f = 0; f < 16; f += 2) *reinterpret_cast<float2 *>(R + 4 * f) = make_float2(m[f], m[f + 1]);
            __syncthreads();
            float *P1 = reinterpret_cast<float *>(T + 1024 * K_ROWB);      // [64][16]
            {
                const int f = tid & 15, seg = tid >> 4;
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < 16; i++) s += *reinterpret_cast<const float *>(T + (16 * seg + i) * K_ROWB + 4 * f);
                P1[seg * 16 + f] = s;
            }
            __syncthreads();
            if (tid < 16) {
                float s = 0.f;
                for (int seg = 0; seg < 64; seg++) s += P1[seg * 16 + tid];
                reinterpret_cast<float *>(A.out)[((size_t)blk * 4 + r) * 16 + tid] = s;
            }
            return;
        }
        float *orow = reinterpret_cast<float *>(A.out) + (size_t)(4 * tid + (int)r) * A.out_stride + col0;
        if (nfb == 16 && (A.out_stride & 3u) == 0) {
            // whole 64-byte pieces: results back into the row, then four lanes per row store 16 rows x 64 bytes per instruction
#pragma unroll
            for (int f = 0; f < 16; f += 2) *reinterpret_cast<float2 *>(R + 4 * f) = make_float2(m[f], m[f + 1]);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int rowl = 64 * (int)wv + 16 * i + (lane >> 2), piece = lane & 3;
                const float2 a = *reinterpret_cast<const float2 *>(T + rowl * K_ROWB + 16 * piece), b = *reinterpret_cast<const float2 *>(T + rowl * K_ROWB + 16 * piece + 8);
                float *dst = reinterpret_cast<float *>(A.out) + (size_t)(4 * rowl + (int)r) * A.out_stride + col0 + 4 * piece;
                *reinterpret_cast<float4 *>(dst) = make_float4(a.x, a.y, b.x, b.y);
            }
        } else {
            for (int f = 0; f < nfb; f++) orow[f] = m[f];
        }
    } else {
        if (nfb == 16 && (A.out_stride & 1u) == 0) {
            // eight lanes per row: every store instruction writes 8 whole 128-byte lines
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int rowl = 64 * (int)wv + 8 * i + (lane >> 3), piece = lane & 7;
                const float2 a = *reinterpret_cast<const float2 *>(T + rowl * K_ROWB + 16 * piece), b = *reinterpret_cast<const float2 *>(T + rowl * K_ROWB + 16 * piece + 8);
                float2 *dst = reinterpret_cast<float2 *>(A.out) + (A.tile_major ? ((size_t)blk * 4096u + (size_t)(4 * rowl + (int)r)) * 16u + 2 * piece
                                                                                : (size_t)(4 * rowl + (int)r) * A.out_stride + col0 + 2 * piece);
                *reinterpret_cast<float4 *>(dst) = make_float4(a.x, a.y, b.x, b.y);
            }
        } else {
            float2 *orow = reinterpret_cast<float2 *>(A.out) + (size_t)(4 * tid + (int)r) * A.out_stride + col0;
            for (int f = 0; f < nfb; f++) orow[f] = *reinterpret_cast<const float2 *>(T + tid * K_ROWB + 8 * f);
        }
    }
    (void)esz;
}

// out[16 blk + f] = ((P[blk][0][f] + P[blk][1][f]) + P[blk][2][f]) + P[blk][3][f]
__global__ __launch_bounds__(256) void k_mix4096_finish(const float *__restrict__ P, float *__restrict__ out, uint32_t nf)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= nf) return;
    const float *p = P + (size_t)(t >> 4) * 64 + (t & 15u);
    out[t] = ((p[0] + p[16]) + p[32]) + p[48];
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------------
struct HugePlan {
    FusedConfig cfg;
    uint32_t cus = 256;
    uint64_t frames_done = 0;
    float *d_taps = nullptr;
    float2 *d_wpre = nullptr, *d_tw4 = nullptr, *d_tw = nullptr, *d_z = nullptr;
    float4 *d_tail[2] = {nullptr, nullptr};
    float *d_part = nullptr;
    int cur = 0;
    std::string name;
};

bool huge_supported(uint32_t M, uint32_t p) { return M == (uint32_t)H_M && p == (uint32_t)P; }

void huge_destroy(HugePlan *p)
{
    if (!p) return;
    void *ptrs[] = {p->d_taps, p->d_wpre, p->d_tw4, p->d_tw, p->d_z, p->d_tail[0], p->d_tail[1], p->d_part};
    for (void *q : ptrs) if (q) (void)hipFree(q);
    delete p;
}

int huge_reset(HugePlan *p, hipStream_t s)
{
    p->frames_done = 0; p->cur = 0;
    CSDR_HIP(hipMemsetAsync(p->d_tail[0], 0, sizeof(float4) * 2048 * H_COLD, s));
    return 0;
}
void huge_seek(HugePlan *p, uint64_t frames) { p->frames_done = frames; }
const char *huge_name(const HugePlan *p) { return p->name.c_str(); }

int huge_create(const FusedConfig &cfg, HugePlan **out)
{
    HugePlan *p = new HugePlan();
    p->cfg = cfg;
    auto fail = [&](int r) { huge_destroy(p); return r; };
    {
        int dev = 0, cus = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        p->cus = (uint32_t)cus;
    }
#define ALLOC(ptr, bytes) do { hipError_t e = hipMalloc((void **)&(ptr), (bytes) ? (bytes) : 1); if (e != hipSuccess) return fail(hip_fail(e, "hipMalloc", __FILE__, __LINE__)); } while (0)
    ALLOC(p->d_taps, sizeof(float) * H_M * P);
    ALLOC(p->d_wpre, sizeof(float2) * 2 * H_M);
    ALLOC(p->d_tw4, sizeof(float2) * 3 * 1024);
    ALLOC(p->d_tw, sizeof(float2) * 1024);
    ALLOC(p->d_z, sizeof(float2) * ((size_t)cfg.max_nf + 1) * H_M);
    ALLOC(p->d_tail[0], sizeof(float4) * 2048 * H_COLD);
    ALLOC(p->d_tail[1], sizeof(float4) * 2048 * H_COLD);
    if (cfg.mix) ALLOC(p->d_part, sizeof(float) * 64 * ((size_t)(cfg.max_nf + 15) / 16));
#undef ALLOC
    CSDR_HIP(hipMemcpy(p->d_taps, cfg.taps, sizeof(float) * H_M * P, hipMemcpyHostToDevice));
    std::vector<float2> wpre(2 * H_M), tw4(3 * 1024), tw(1024);
    for (uint32_t i = 0; i < 2u * H_M; i++) {           // the NCO phase sequence has period 2M for a power-of-two M (as every fused plan)
        float c, sn;
        nco_phasor(i * cfg.d_theta, &c, &sn);
        wpre[i] = make_float2(c, -sn);
    }
    const double tp = -2.0 * 3.14159265358979323846;
    for (int r = 1; r < 4; r++)
        for (int j1 = 0; j1 < 1024; j1++) tw4[(r - 1) * 1024 + j1] = make_float2((float)std::cos(tp * (double)(j1 * r) / 4096.0), (float)std::sin(tp * (double)(j1 * r) / 4096.0));
    for (int i = 0; i < 1024; i++) tw[i] = make_float2((float)std::cos(tp * (double)i / 1024.0), (float)std::sin(tp * (double)i / 1024.0));
    CSDR_HIP(hipMemcpy(p->d_wpre, wpre.data(), sizeof(float2) * wpre.size(), hipMemcpyHostToDevice));
    CSDR_HIP(hipMemcpy(p->d_tw4, tw4.data(), sizeof(float2) * tw4.size(), hipMemcpyHostToDevice));
    CSDR_HIP(hipMemcpy(p->d_tw, tw.data(), sizeof(float2) * tw.size(), hipMemcpyHostToDevice));
    CSDR_HIP(hipMemset(p->d_tail[0], 0, sizeof(float4) * 2048 * H_COLD));
    p->name = cfg.mix ? "k_front4096+k_back4096<FM,mix>" : (cfg.fm ? "k_front4096+k_back4096<FM>" : "k_front4096+k_back4096<CF32>");
    *out = p;
    return 0;
}

// frames per run: long enough that the 19 cold-start frames (about 8 frames' worth of work) stay below ~8 % of a run
static uint32_t huge_runs(uint32_t nf, uint32_t cus)
{
    uint32_t nruns = cus / 4;                           // 4 siblings per run, one 512-thread workgroup per CU
    if (const char *e = diag_env("CSDR_RUN4096_RUNS")) { const uint32_t v = (uint32_t)atoi(e); if (v >= 1) nruns = v; }
    while (nruns > 1 && nf / nruns < 48) nruns >>= 1;    // (a run pays 20 cold-start frames: the reference chunk of 4096 frames is 64 runs of 64)
    if (nruns > 8) nruns &= ~7u;                        // the XCD-friendly workgroup -> (run, sibling) map wants a multiple of 8
    return nruns ? nruns : 1;
}

int huge_process(HugePlan *p, const FusedCall &call, hipStream_t s, KernelTimer *timer)
{
    const FusedConfig &c = p->cfg;
    const uint32_t nf = call.nf;
    if (!nf) return 0;
    Front4096Args A{};
    A.x = reinterpret_cast<const float4 *>(call.d_in); A.tail = p->d_tail[p->cur]; A.z = p->d_z;
    A.taps = p->d_taps; A.wpre = p->d_wpre; A.tw4 = p->d_tw4;
    A.nf = nf; A.nruns = huge_runs(nf, p->cus); A.parity0 = (uint32_t)(p->frames_done & 1u);
    const double beta = c.dc_block ? (double)c.dc.beta : 0.0;
    A.alpha = c.dc_block ? (float)(1.0 - beta) : 0.0f; A.beta = (float)beta; A.l2beta = c.dc_block ? (float)std::log2(beta) : -1000.0f;
    for (int i = 0; i < 16; i++) A.b16[i] = (float)std::pow(beta, 16.0 * i);
    for (int i = 0; i < 17; i++) A.b256[i] = (float)std::pow(beta, 256.0 * i);
    int r;
    if (timer && (r = timer->begin(s))) return r;
    hipLaunchKernelGGL(k_front4096, dim3(4 * A.nruns), dim3(512), 0, s, A);
    Back4096Args Bk{};
    Bk.z = p->d_z; Bk.tw = p->d_tw; Bk.nf = nf; Bk.out_stride = nf;
    Bk.tile_major = (call.tile_major && !c.fm && !c.mix) ? 1u : 0u;
    if (call.tile_major && (!Bk.tile_major || (nf & 15u))) { set_error("huge_process: tile-major output asked for a call k_back4096<CF32> does not take"); return -1; }
    Bk.pk = phase_consts(1.0f); Bk.pk.hp *= c.fm_ref; Bk.pk.pi *= c.fm_ref; Bk.pk.ref = c.fm_ref;
    Bk.fm_ref = c.fm_ref; Bk.tiny = 1e-37f;
    const uint32_t nblk = (nf + 15) / 16;
    if (c.mix) {
        if (!c.fm) { set_error("k_back4096: --mix without freqdem goes through the mix identity (capi route)"); return -1; }
        Bk.out = p->d_part;
        hipLaunchKernelGGL((k_back4096<2>), dim3(4 * nblk), dim3(1024), 0, s, Bk);
        hipLaunchKernelGGL(k_mix4096_finish, dim3((nf + 255) / 256), dim3(256), 0, s, (const float *)p->d_part, (float *)call.d_out, nf);
    } else {
        Bk.out = call.d_out;
        if (c.fm) hipLaunchKernelGGL((k_back4096<1>), dim3(4 * nblk), dim3(1024), 0, s, Bk);
        else hipLaunchKernelGGL((k_back4096<0>), dim3(4 * nblk), dim3(1024), 0, s, Bk);
    }
    if (timer && (r = timer->end(s))) return r;
    CSDR_HIP(hipGetLastError());
    // the call's last 19 raw frames for the next call's run 0 (older ones move up when the call is shorter than that)
    float4 *nt = p->d_tail[p->cur ^ 1];
    const size_t fb = sizeof(float4) * 2048;
    if (nf >= (uint32_t)H_COLD) {
        CSDR_HIP(hipMemcpyAsync(nt, reinterpret_cast<const char *>(call.d_in) + (size_t)(nf - H_COLD) * fb, fb * H_COLD, hipMemcpyDeviceToDevice, s));
    } else {
        CSDR_HIP(hipMemcpyAsync(nt, reinterpret_cast<const char *>(p->d_tail[p->cur]) + (size_t)nf * fb, fb * (H_COLD - nf), hipMemcpyDeviceToDevice, s));
        CSDR_HIP(hipMemcpyAsync(reinterpret_cast<char *>(nt) + (size_t)(H_COLD - nf) * fb, call.d_in, fb * nf, hipMemcpyDeviceToDevice, s));
    }
    p->cur ^= 1;
    p->frames_done += nf;
    return 0;
}

bool huge_tile_major_ok(const HugePlan *p, uint32_t nf) { return p && !p->cfg.fm && !p->cfg.mix && nf && (nf & 15u) == 0; }

}  // namespace csdr
