// Second-generation run kernel of the fused M = 64 chain, CF32 output (BASELINE configs[1]: 64-ch PFB, DeNo; replaces
// k_run64<CF32> of kernels_fused_small.hip for whole-band calls whose frame count is a multiple of 64).
//
//   raw CF32 x --DC blocker--> y --NCO pre-mix, 14-tap polyphase FIR--> X_t[j] --64-point forward DFT (16 x 4)--> Y_t[k]  -> out[64][nf]
//
// Built the way k_run256v2 / k_run1024v2 are: 256 threads, two workgroups per CU, a TILE is 4096 consecutive samples = 64
// frames, DMA'd (global_load_lds) into one of two 32 KiB LDS buffers a tile ahead and transformed in place:
//   raw image (16-byte XOR swizzle) --serial DC scan per 16-sample run--> y' --column layout: thread (j, q) owns branch j of
//   frames 16 q .. 16 q + 15 (wave q = a block of 16 frames = 8 KiB of the buffer): group state chain, pre-mix; the pre-mixed
//   samples go back in place because the 14-tap window of a thread reaches 13 frames into the block of the wave before it
//   (for wave 0: into a 6.5 KiB history of the previous tile) --> FIR --> X.
// From there a wave only touches its own block: X (written by its lanes as branches) is read by the same lanes as
// (frame, b) for the radix-16 pass, Z goes back, the radix-4 pass reads it as (frame f in the low four lane bits, k) and its
// results leave straight from registers: a 16-lane row writes the 128 contiguous bytes a channel row gets from 16 frames.
// Layouts inside a block are chosen per pass (XOR with 4 x the frame index: every access a permutation of the banks).
// 4 barriers per tile; no LDS round trip for Y; k_run64 spends 76 % of its time in the LDS pipe on three more of them.
#include "fused_v2_common.h"

#ifndef S2_SNOP
// no wait states in front of the asm stores: the kernel has no SGPR spills and the bases are SALU results (see
// kernels_fused_v2.hip: V2_SNOP; tests/test_build_invariants.py checks the spill count)
#define S2_SNOP ""
#endif

namespace csdr {
namespace {

constexpr int S2_BUF = 4096;                        // float2 per tile buffer
constexpr int S2_HIST = 2 * S2_BUF;                 // 13 x 64 pre-mixed samples of the frames in front of the tile
constexpr int S2_TW = S2_HIST + 13 * 64;            // 16 x 4 pass-1 twiddles W64^(k1 b) at [k1][b]
constexpr int S2_TT = S2_TW + 64;                   // 16 group totals
constexpr int S2_RED = S2_TT + 16;
constexpr int S2_F2 = S2_RED + 16;                  // 9120 float2 = 72 960 B: two workgroups per CU
constexpr int S2_WU = 6;                            // read-only warm-up tiles (DC state)

struct Run64v2Args {
    const float2 *x; float2 *out;
    const float *taps;          // h[(63 - j) + 64 n]
    const float2 *tw;           // [16][4] W64^(k1 b)
    const float2 *wpre;         // [2][64] conj(nco phasor) of branch j at even / odd global frames
    const float2 *uhist_in; float2 *uhist_out;    // [13][64] pre-mixed, DC-blocked window before / after the call
    const float2 *vend_in; float2 *vend_out;      // DC blocker state v1
    uint32_t nf, nb, nruns, n0, parity0, out_stride;
    uint32_t tile_major;        // the lines of a 16-frame block back to back, [block][64][128 B] (the plane k_agc_spec_tm reads), instead of rows [64][nf]
    float alpha, beta, l2beta;
    float b16[16], b256[17];
    uint32_t nowu;              // 1: a run starts its halo tile from DC state 0 and leaves the state in front of its last tile in cpre[w + 1];
    float2 *cpre;               //    k_run64_dcfix adds what the true state contributes to the channels 30..33 (as k_run256v2)
};

// first half of the runs (dispatched first: the older workgroups of their CUs) share n0 tiles, the second half the rest
__device__ __forceinline__ void run64_bounds(const Run64v2Args &A, unsigned w, unsigned &first, unsigned &last)
{
    if (A.n0 == 0 || (A.nruns & 1u)) {
        first = (unsigned)((unsigned long long)w * A.nb / A.nruns);
        last = (unsigned)((unsigned long long)(w + 1) * A.nb / A.nruns);
        return;
    }
    const unsigned half = A.nruns / 2, s = w / half, i = w - s * half;
    const unsigned base = s ? A.n0 : 0u, tiles = s ? A.nb - A.n0 : A.n0;
    first = base + (unsigned)((unsigned long long)i * tiles / half);
    last = base + (unsigned)((unsigned long long)(i + 1) * tiles / half);
}

__global__ __launch_bounds__(256, 2) void k_run64v2(Run64v2Args A)
{
    __shared__ __attribute__((aligned(16))) float2 L[S2_F2];
    float2 *hist = L + S2_HIST, *tw_s = L + S2_TW, *Tt = L + S2_TT, *red = L + S2_RED;
    const int tid = threadIdx.x, j = tid & 63;
    const unsigned qw = (unsigned)__builtin_amdgcn_readfirstlane(tid >> 6);        // my wave = my block of 16 frames
    const unsigned w = blockIdx.x;
    unsigned first, last;
    run64_bounds(A, w, first, last);
    const float4 *x4 = reinterpret_cast<const float4 *>(A.x);

    if (tid < 64) tw_s[tid] = A.tw[tid];
    const unsigned goff = dma_offset(tid);
    const unsigned wave_u = qw;
    const unsigned lds_wave = (unsigned)(size_t)(__attribute__((address_space(3))) float2 *)L + 1024u * wave_u;
    // the first tile the loop below walks (a run >= 1: its halo tile, into buffer 1) is requested before anything else (round 3, as
    // k_run256v2): the run start is one burst of memory traffic, not a chain of round trips
    if ((w == 0 ? first : first - 1) < last) dma_tile(x4 + (size_t)(w == 0 ? first : first - 1) * 2048, goff, lds_wave + (w == 0 ? 0u : (unsigned)(S2_BUF * 8u)));
    float2 c;                                           // DC state v before the next tile (same in every lane)
    unsigned tile_begin = first;
    if (w == 0) {
        c = A.vend_in[0];
        for (int i = tid; i < 13 * 64; i += 256) hist[i] = A.uhist_in[i];
    } else if (A.nowu) {
        // no warm-up window (round 5): state 0 in front of the halo tile.  The step this puts on the blocker's output has left the 13-frame
        // FIR inside the halo tile's 64 frames; the slowly decaying DC term behind it only reaches the channels 30..33 (k_run64_dcfix)
        tile_begin = first - 1;
        c = make_float2(0.f, 0.f);
    } else {
        // read-only warm-up: the DC state before the halo tile from the six tiles in front of it (beta^24576 = 4.6e-6 of the
        // older state is dropped, as in every run kernel); the halo tile then leaves the 13-frame history behind
        tile_begin = first - 1;
        const unsigned h0 = tile_begin > (unsigned)S2_WU ? tile_begin - S2_WU : 0u;
        float2 acc = make_float2(0.f, 0.f);
        {
            float4 raw[8];
            // weight of my piece `it` of a tile: beta^(4095 - n), n = n0 + 512 it: a running product (two registers instead of sixteen)
            float wt0, wt1;
            {
                const int wave = tid >> 6, lane = tid & 63;
                const int slot = 64 * wave + lane, q = slot >> 3;
                const int i = (slot & 7) ^ ((q >> 1) & 7);
                const int n = 16 * q + 2 * i;
                wt0 = exp2f((float)(4095 - n) * A.l2beta);
                wt1 = exp2f((float)(4094 - n) * A.l2beta);
            }
            const float wstep = A.l2beta < -100.0f ? 0.0f : exp2f(-512.0f * A.l2beta);
            auto fold = [&](const float4 (&r)[8]) {
                float2 p = make_float2(0.f, 0.f);
                float a0 = wt0, a1 = wt1;
#pragma unroll
                for (int it = 0; it < 8; it++) {
                    p = cfma(make_float2(r[it].x, r[it].y), a0, p);
                    p = cfma(make_float2(r[it].z, r[it].w), a1, p);
                    a0 *= wstep; a1 *= wstep;
                }
                acc = cfma(acc, A.b256[16], p);
            };
            unsigned t = h0;
            if (tile_begin - h0 == (unsigned)S2_WU) {   // the usual window: six tiles in one batch of loads (one memory latency, not six)
                float4 rb[8], rc[8], rd[8], re[8], rf[8];
                tile_load(x4 + (size_t)t * 2048, 256, raw, tid); tile_load(x4 + (size_t)(t + 1) * 2048, 256, rb, tid);
                tile_load(x4 + (size_t)(t + 2) * 2048, 256, rc, tid); tile_load(x4 + (size_t)(t + 3) * 2048, 256, rd, tid);
                tile_load(x4 + (size_t)(t + 4) * 2048, 256, re, tid); tile_load(x4 + (size_t)(t + 5) * 2048, 256, rf, tid);
                fold(raw); fold(rb); fold(rc); fold(rd); fold(re); fold(rf);
                t += 6;
            }
#pragma unroll 1
            for (; t < tile_begin; t++) {
                tile_load(x4 + (size_t)t * 2048, 256, raw, tid);
                fold(raw);
            }
        }
        c = wg_sum(acc, red, tid);
        if (h0 == 0) c = cfma(A.vend_in[0], exp2f((float)(4096u * tile_begin) * A.l2beta), c);
    }
    __syncthreads();                                    // twiddles, history

    // ------------------------------------------------------------------ per-thread constants of the tile loop
    float h[P];
#pragma unroll
    for (int n = 0; n < P; n++) h[n] = A.taps[(63 - j) + n * 64];
    const float2 Wa = A.wpre[(A.parity0 & 1) * 64 + j], Wb = A.wpre[((A.parity0 & 1) ^ 1) * 64 + j];   // frames with even / odd index in the tile
    const v2f Wav = to_v(Wa), Wbv = to_v(Wb);
    float kJ[4];                                        // -alpha beta^(64 r + j): group state into frame r of a group, column j
#pragma unroll
    for (int r = 0; r < 4; r++) kJ[r] = -A.alpha * exp2f((float)(64 * r + j) * A.l2beta);
    const float b256 = A.b256[1];
    const int q = tid, sw = (q >> 1) & 7;
    const unsigned raw_a0 = (unsigned)q * 128u + ((unsigned)sw << 4);            // DC scan: slot i of my run: raw_a ^ (i << 4)
    // column layout of the raw image: sample of frame 16 qw + i, branch j sits at float2 1024 qw + 64 i + (colP ^ ((4 i) & 12))
    const unsigned colP = (unsigned)(16 * (j >> 4) + 2 * (((j & 15) >> 1) ^ (j >> 5)) + (j & 1));
    const unsigned blk = 1024u * qw;                                            // my wave's block (float2)
    // block-local layouts (float2 inside the block): X[i][jj] at 64 i + (jj ^ 4 i); Z[f][k1][b] at 64 f + ((4 k1 + b) ^ 4 f)
    const int fl = (tid & 63) >> 2, b1 = tid & 3;                               // pass 1: frame in the block, b
    const int f2 = tid & 15, k4 = (tid & 63) >> 4;                              // pass 2 / tail: frame in the block, channel group
    // row-major: channel k, frame t at (k nf + t) 8; tile-major: ((t >> 4) 64 + k) 128 + (t & 15) 8 with t >> 4 = 4 b + qw, t & 15 = f2:
    // a 16-lane row stores one 128-byte line either way
    const uint32_t voff = A.tile_major ? (uint32_t)k4 * 128u + 8192u * qw + 8u * (uint32_t)f2
                                       : ((uint32_t)k4 * A.out_stride + 16u * qw + (uint32_t)f2) * 8u;     // + (4 m + 16 k2) rows, + 64 b frames
    const size_t rowb = A.tile_major ? (size_t)128 : (size_t)A.out_stride * 8u;
    const size_t tileb = A.tile_major ? (size_t)32768 : (size_t)512;               // bytes between the first lines of consecutive tiles

    auto tile = [&](unsigned b_, const int par, const bool warm) {
        unsigned b = (unsigned)__builtin_amdgcn_readfirstlane((int)b_);
        asm volatile("" : "+s"(b));
        char *B = reinterpret_cast<char *>(L) + par * (S2_BUF * 8);
        float2 *Bf = reinterpret_cast<float2 *>(B);
        bar();                                          // B_a: the tile image has landed (every wave waited for its own DMA); the other buffer is free
        if (b + 1 < last) dma_tile(x4 + (size_t)(b + 1) * 2048, goff, lds_wave + (unsigned)(par ^ 1) * (S2_BUF * 8u));
        // the state in front of my last tile = in front of the next run's halo tile (my own start error is beta^(>= 13 x 4096) of it by now)
        if (b + 1 == last && tid == 0 && A.nowu) A.cpre[w + 1] = c;
        // ---- DC blocker inside a 256-sample group: thread q owns the run of 16 consecutive samples q (as k_run256v2)
        {
            unsigned raw_a = raw_a0;
            asm volatile("" : "+v"(raw_a));
            const float na = opaque_v(-A.alpha), be = opaque_v(A.beta);
            v4f xr[8];
            float2 s = make_float2(0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                xr[i] = *reinterpret_cast<const v4f *>(B + (raw_a ^ (unsigned)(i << 4)));
                s = make_float2(fmaf(s.x, be, xr[i].x), fmaf(s.y, be, xr[i].y));
                s = make_float2(fmaf(s.x, be, xr[i].z), fmaf(s.y, be, xr[i].w));
            }
            {
                float2 t;
                t = dpp2<0x111>(s); s = cfma(t, A.b16[1], s);
                t = dpp2<0x112>(s); s = cfma(t, A.b16[2], s);
                t = dpp2<0x114>(s); s = cfma(t, A.b16[4], s);
                t = dpp2<0x118>(s); s = cfma(t, A.b16[8], s);
            }
            if ((q & 15) == 15) Tt[q >> 4] = s;
            s = dpp2<0x111>(s);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                v4f y;
                y.x = fmaf(s.x, na, xr[i].x); y.y = fmaf(s.y, na, xr[i].y);
                s = make_float2(fmaf(s.x, be, xr[i].x), fmaf(s.y, be, xr[i].y));
                y.z = fmaf(s.x, na, xr[i].z); y.w = fmaf(s.y, na, xr[i].w);
                s = make_float2(fmaf(s.x, be, xr[i].z), fmaf(s.y, be, xr[i].w));
                *reinterpret_cast<v4f *>(B + (raw_a ^ (unsigned)(i << 4))) = y;
            }
        }
        bar();                                          // B_c: y' (group carry still missing) and the group totals are visible
        // ---- column layout: nw[i] = frame 16 qw + i, branch j; group state chain (uniform), my four groups are 4 qw .. 4 qw + 3
        float2 nw[16];
#pragma unroll
        for (int i = 0; i < 16; i++) nw[i] = Bf[blk + 64 * i + (colP ^ (unsigned)((4 * i) & 12))];
        {
            v2f V = to_v(c), Vm[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
            const v2f bv = {b256, b256};
#pragma unroll
            for (int g = 0; g < 16; g++) {
                if ((unsigned)(g >> 2) == qw) Vm[g & 3] = V;            // wave-uniform
                V = __builtin_elementwise_fma(V, bv, to_v(Tt[g]));
            }
            c = to_f2(V);
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const float k = kJ[i & 3];
                nw[i] = to_f2(__builtin_elementwise_fma(Vm[i >> 2], (v2f){k, k}, to_v(nw[i])));
            }
        }
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            v2f a0 = to_v(nw[i]), a1 = to_v(nw[i + 1]);
            cmul2_v(a0, Wav, a1, Wbv);
            nw[i] = to_f2(a0); nw[i + 1] = to_f2(a1);
        }
        if (b + 1 == A.nb && qw == 3) {                 // the stream's last 13 frames of u
#pragma unroll
            for (int i = 3; i < 16; i++) A.uhist_out[(i - 3) * 64 + j] = nw[i];
        }
        if (!warm) {
            // the pre-mixed samples go back in place: the wave behind mine reads its window out of my block
#pragma unroll
            for (int i = 0; i < 16; i++) Bf[blk + 64 * i + (colP ^ (unsigned)((4 * i) & 12))] = nw[i];
        }
        bar();                                          // B_1: u visible (history of the previous tile still in place)
        float2 win[13];                                 // frames 16 qw - 13 .. 16 qw - 1 of branch j
        if (!warm) {
            if (qw == 0) {
#pragma unroll
                for (int m = 0; m < 13; m++) win[m] = hist[64 * m + j];
            } else {
#pragma unroll
                for (int m = 0; m < 13; m++) win[m] = Bf[blk - 1024u + 64 * (3 + m) + (colP ^ (unsigned)((4 * (3 + m)) & 12))];
            }
        }
        bar();                                          // B_2: every window is in registers: X may overwrite u, the history may move on
        if (qw == 3) {
#pragma unroll
            for (int i = 3; i < 16; i++) hist[64 * (i - 3) + j] = nw[i];
        }
        if (warm) return;
        // ---- polyphase FIR, four frames at a time; X into my block: X[i][j] at 64 i + (j ^ 4 i)
#pragma unroll
        for (int i0 = 0; i0 < 16; i0 += 4) {
            v2f acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
            for (int n = P - 1; n >= 0; n--) {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int t = i0 + u - n;
                    const float2 s2 = (t >= 0) ? nw[t] : win[13 + t];
                    acc[u] = __builtin_elementwise_fma((v2f){s2.x, s2.y}, (v2f){h[n], h[n]}, acc[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) Bf[blk + 64 * (i0 + u) + (unsigned)(j ^ (4 * (i0 + u)))] = to_f2(acc[u]);
        }
        // ---- DFT pass 1 (my wave's block only: no barrier): lane (fl, b1): radix 16 over a, n = 4 a + b1
        v2f vv[16];
#pragma unroll
        for (int a = 0; a < 16; a++) vv[a] = to_v(Bf[blk + 64 * fl + (unsigned)((4 * a + b1) ^ (4 * fl))]);
        fft16_v(vv);
#pragma unroll
        for (int i = 1; i < 16; i++) vv[i] = cmul_v(vv[i], to_v(tw_s[4 * XIDX(i) + b1]));
#pragma unroll
        for (int i = 0; i < 16; i++) Bf[blk + 64 * fl + (unsigned)((4 * XIDX(i) + b1) ^ (4 * fl))] = to_f2(vv[i]);     // Z[fl][k1][b1]
        // ---- pass 2 + stores: lane (f2, k4): radix 4 over b for k1 = k4 + 4 m; Y[k1 + 16 k2] of frame 16 qw + f2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // next tile image (issued at the top of this tile): nothing else is outstanding
        char *obase = reinterpret_cast<char *>(A.out) + tileb * b;
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int k1 = k4 + 4 * m;                  // (runtime k4: the address below is lane arithmetic)
            const unsigned za = blk + 64u * (unsigned)f2 + (unsigned)((4 * k1) ^ (4 * f2));
            const v4f z01 = *reinterpret_cast<const v4f *>(Bf + za), z23 = *reinterpret_cast<const v4f *>(Bf + za + 2);
            v2f y0 = {z01.x, z01.y}, y1 = {z01.z, z01.w}, y2 = {z23.x, z23.y}, y3 = {z23.z, z23.w};
            bfly4_v(y0, y1, y2, y3);
            const v2f yk[4] = {y0, y1, y2, y3};
#pragma unroll
            for (int k2 = 0; k2 < 4; k2++) {
                const char *rowp = obase + (size_t)(4 * m + 16 * k2) * rowb;
                asm volatile(S2_SNOP "global_store_dwordx2 %0, %1, %2" :: "v"(voff), "v"(yk[k2]), "s"(rowp) : "memory");
            }
        }
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (the tile requested at the kernel's entry)
    if (w > 0) {                                        // the halo tile (buffer 1): DC blocker, pre-mix and the history only
        tile(tile_begin, 1, true);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    for (unsigned b = first; b < last; b += 2) {
        tile(b, 0, false);
        if (b + 1 >= last) break;
        tile(b + 1, 1, false);
    }
    if (last == A.nb && tid == 0) A.vend_out[0] = c;
}

// out[30 + ch][64 first_w + t] += cpre[w] x R[parity][t][ch]: what the state run w started without contributes to the four channels
// around DC over its first RUN64_DCFIX_F output frames (the chain is linear up to its CF32 output; 3 MB of traffic per launch)
__global__ __launch_bounds__(256) void k_run64_dcfix(Run64v2Args A, const float2 *__restrict__ rt)
{
    const unsigned w = blockIdx.x + 1u;
    unsigned first, last;
    run64_bounds(A, w, first, last);
    const float2 c = A.cpre[w];
    const unsigned par = A.parity0 & 1u;                // a tile is 64 frames: every halo tile starts on the call's parity
    // one element per thread (grid.y = 7 slices of 256): a loop of read-modify-writes ran one memory round trip after the other (8 us)
    const unsigned e = blockIdx.y * 256u + threadIdx.x;
    if (e >= 4u * RUN64_DCFIX_F) return;
    const unsigned ch = e / RUN64_DCFIX_F, t = e % RUN64_DCFIX_F;
    const size_t fr = (size_t)64 * first + t;
    if (fr >= (size_t)64 * last) return;                // (runs are >= 14 tiles: never)
    float2 *o = A.out + (A.tile_major ? ((fr >> 4) * 64u + (30u + ch)) * 16u + (fr & 15u) : (size_t)(30u + ch) * A.out_stride + fr);
    const float2 r = rt[((size_t)par * RUN64_DCFIX_F + t) * 4u + ch];
    float2 y = *o;
    y.x += c.x * r.x - c.y * r.y; y.y += c.x * r.y + c.y * r.x;
    *o = y;
}

}  // namespace

uint32_t run64_v2_runs(uint32_t nf, uint32_t cus)
{
    // two workgroups per CU; a run >= 1 reads 6 warm-up tiles and walks a halo tile: at least 16 tiles per run on average.  A call pays
    // those ~23 tile times (~60 us) whatever its size, so k_run64 keeps the calls below 3072 tiles (196 608 frames; measured by call size:
    // 4096 frames 17 us against 60 us, 131 072 frames 58 against 66, 262 144 frames 99 against 76)
    if (nf % 64u) return 0;
    const uint32_t nb = nf / 64u;
    if (nb < 3072u && !diag_env("CSDR_RUN64_V2_ALL")) return 0;
    uint32_t nruns = 2 * cus;
    if (nruns > nb / 16) nruns = nb / 16;
    if (nruns > 2) nruns &= ~1u;
    return nruns;                                       // 0: not a call for this kernel
}

int run64_v2_launch(const Run64v2Host &h, hipStream_t s, KernelTimer *timer)
{
    Run64v2Args A{};
    A.x = h.x; A.out = h.out; A.taps = h.taps; A.tw = h.tw; A.wpre = h.wpre;
    A.uhist_in = h.uhist_in; A.uhist_out = h.uhist_out; A.vend_in = h.vend_in; A.vend_out = h.vend_out;
    A.nf = h.nf; A.nb = h.nf / 64u; A.nruns = h.nruns; A.parity0 = h.parity0; A.out_stride = h.nf; A.tile_major = h.tile_major ? 1u : 0u;
    {
        static const double wt = diag_env("CSDR_RUN64_WEIGHT") ? atof(diag_env("CSDR_RUN64_WEIGHT")) : 1.1;   // share of the older workgroup of a CU (1 = even; 1.1 measured best: 243 vs 250 us)
        A.n0 = (h.nruns >= 2 && !(h.nruns & 1u) && wt > 1.0 && wt < 1.5) ? (uint32_t)std::llround(0.5 * wt * (double)A.nb) : 0u;
    }
    const double beta = h.dc_block ? h.beta : 0.0;
    A.alpha = h.dc_block ? (float)(1.0 - beta) : 0.0f; A.beta = (float)beta; A.l2beta = h.dc_block ? (float)std::log2(beta) : -1000.0f;
    for (int i = 0; i < 16; i++) A.b16[i] = (float)std::pow(beta, 16.0 * i);
    for (int i = 0; i < 17; i++) A.b256[i] = (float)std::pow(beta, 256.0 * i);
    int r;
    if (timer && (r = timer->begin(s))) return r;
    A.nowu = (h.cpre && h.rt && h.dc_block && h.nruns >= 2 && A.nb / h.nruns >= 14u) ? 1u : 0u;
    A.cpre = h.cpre;
    hipLaunchKernelGGL(k_run64v2, dim3(h.nruns), dim3(256), 0, s, A);
    if (A.nowu) hipLaunchKernelGGL(k_run64_dcfix, dim3(h.nruns - 1u, (4u * RUN64_DCFIX_F + 255u) / 256u), dim3(256), 0, s, A, h.rt);
    if (timer && (r = timer->end(s))) return r;         // the bracket covers the correction kernel: it is part of every no-warm-up step
    CSDR_HIP(hipGetLastError());
    return 0;
}

}  // namespace csdr
