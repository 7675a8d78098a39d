// Second-generation run kernel of the fused M = 256 chain (replaces k_run256 of kernels_fused.hip for whole-band calls).
//
//   raw CF32 x --DC blocker--> y --NCO pre-mix, 14-tap polyphase FIR--> X_t[j] --256-point forward DFT (16 x 16)--> Y_t[k]
//              --per-channel freqdem--> out[256][nf]                (8 B read + 4 / 8 B written per sample, Liquid.chs:575-589,
//                                                                     828-862, 324-328)
//
// Same run structure as k_run256 (a workgroup walks a run of 16-frame tiles; FIR window, DC state and freqdem history
// never leave the workgroup; read-only warm-up at the run start), but the tile body is rebuilt around what tools/probes/issue_probe*.hip measured on gfx950:
//   * every VALU instruction costs an issue slot of ~3 (plain VOP1/2/3 on VGPRs) to ~4.7 cycles (DPP, packed, SGPR
//     operand, min/max/bfi), v_rcp 9 and a VCC-based v_cndmask 16; k_run256 spent 160 DPP + 190 v_mov + 32 VCC selects
//     per tile and thread.  Here the DC blocker runs as a plain serial scan over 16 consecutive samples per thread out
//     of an LDS image that the tile was DMA'd into (global_load_lds: no VGPR staging, no ds_write), freqdem uses literal
//     coefficients and SGPR-mask selects, and the FIR window changes hands by register renaming (two tiles per loop
//     iteration) instead of moves.
//   * barriers: 4 per tile instead of 10-12.  Pass 2 of the DFT runs with the frame index in the low four lane bits, so
//     the previous frame of a channel is one DPP row shift away and a 16-lane row writes 64 (F32) / 128 (CF32)
//     contiguous bytes of a channel row: no transpose of Y or of the demodulated samples through LDS at all.
//   * two workgroups per CU (70 KiB of LDS each), up to 256 VGPRs: taps and pass-1 twiddles live in registers.
//
// LDS map (float2 units): two tile buffers of 4096 (tile b lives in buffer b & 1 from its DMA to its last Z read, the
// other buffer receives tile b + 1 meanwhile): raw image, 16-byte XOR swizzle (run-major b128 and column-major b64 both
// conflict-free) -> y' in place -> FIR output X in place (thread j overwrites exactly the column it read) -> pass-1
// output Z (own swizzle); STASH 256: last Y frame of the previous tile in pass-2 register order; T 16 frame totals; RED.
#ifndef V2_FM_PACKED
#define V2_FM_PACKED 1    // freqdem on packed pairs (fm_quad) instead of one sample at a time (fm_sample)
#endif
#ifndef V2_ABLATE
#define V2_ABLATE 0      // timing experiments only: 1 no input DMA in the loop, 2 no output stores, 4 no freqdem, 8 one FIR tap, 16 no butterflies in the two DFT passes, 32 no DC scan arithmetic, 64 no y' write-back, 128 no Z write (one LDS pass less each)
#endif

#ifndef V2_COLSCAN
#define V2_COLSCAN 0     // 1 (A/B build, verdict r03 #1 (iii)): no row-layout scan pass over the tile image -- the raw column is read once
                         // (b64, column layout) and the in-run part of the DC blocker runs on the registers as 16-lane DPP row scans
                         // (fused_common.h: col_run_scan / col_run_carries, k_run256's scheme): two LDS passes of the tile less
                         // (scan read + y' write-back), 2 KiB of run totals + 2 KiB of run carries instead
#endif
#ifndef V2_PAIR
#define V2_PAIR 1       // FM: whole 128-byte row lines per tile pair (0: every tile stores its own 64-byte halves, for A/B)
#endif
#ifndef V2_STORE_AUX
#define V2_STORE_AUX ""     // cache policy bits of the output stores (" nt", " sc1", " sc0 sc1") for A/B builds
#endif
#ifndef V2_PAIR_CF
#define V2_PAIR_CF 0    // CF32: 1 = the even tile of a pair keeps its results, the odd tile stores 256 bytes per row (A/B build: measured below)
#endif
#ifndef V2_BAR_E
#define V2_BAR_E 0      // 1: the (redundant) barrier between the X reads and the Z writes of pass 1, for A/B
#endif
#ifndef V2_PAD
#define V2_PAD 1        // 1 (round 5): every frame of a tile buffer is padded by 128 bytes (stride 2176): the two frames a 32-lane b64 group of
                        // pass 1 touches (X reads, Z writes) and the 16 frames a b128 group of pass 2 reads land on different bank halves --
                        // tools/lds_conflicts_run256v2.py: 382 -> 286 LDS cycles per tile and wave, conflict-free.  0: round 4's dense image (A/B)
#endif
#ifndef V2_SNOP
// Wait states in front of the asm stores would cover a scalar base that comes straight out of a spill lane (v_readlane ->
// VMEM hazard, fused_v2_common.h); they cost 1 % of the launch, and this kernel has no SGPR spills (the bases are SALU
// results of the tile index): empty, and tests/test_build_invariants.py fails when a change makes hipcc spill SGPRs here.
#define V2_SNOP ""
#endif

#include "fused_v2_common.h"
#include <cstddef>
#include <type_traits>

namespace csdr {
namespace {

constexpr unsigned V2_FSB = 2048u + (V2_PAD ? 128u : 0u);   // bytes between the frames of a tile buffer
constexpr int V2_FS2 = (int)V2_FSB / 8;            // the same in float2
constexpr int V2_BUF = 16 * V2_FS2;                // float2 per tile buffer (32 KiB + 2 KiB of padding)
// two tile buffers, then: STASH 256, T 16, RED 16, pass-1 twiddles 256
constexpr int V2_STASH = 2 * V2_BUF;
constexpr int V2_F2 = V2_STASH + 256 + 32 + 256 + (V2_COLSCAN ? 512 : 0);   // 9248 float2 = 73 984 B: two workgroups per CU (dense image: 69 888 B)
#define V2_ZSW(f) (V2_PAD ? (((f) >> 1) & 7) : ((f) & 7))       // 16-byte slot swizzle of frame f's Z rows
// 16-byte slot swizzle of run a (= 16 consecutive samples = 128 bytes) of a frame, raw image / y' / X.  The dense image used (a >> 1) & 7:
// conflict-free for the b128 READS (lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... over 64 banks: MI355X_MICROARCH.md, LDS) but
// 2-way conflicted for the y' WRITE-BACK -- ds_write_b128 goes 8 consecutive lanes at a time over 32 banks, and runs 2m, 2m + 1 shared a
// swizzle: 64 of ~570 LDS cycles per tile and wave, the whole of SQ_LDS_BANK_CONFLICT (4.53 M per launch, the same with and without the
// padding and with the tile DMA removed: profiles/r05_headline_*).  With the padded frames a & 7 serves both (the frames of a read group
// sit on different bank halves): tools/lds_conflicts_run256v2.py.
#ifndef V2_RSWN
#define V2_RSWN V2_PAD      // 1: run swizzle a & 7 (needs the padded frames); 0 with V2_PAD=1: round 5's first step, for A/B
#endif
#define V2_RSW(a) (V2_RSWN ? ((a) & 7) : (((a) >> 1) & 7))

struct V2Args {
    RunArgs r;
};

// G > 1: INTERLEAVED CHANNEL SHARD g = A.c0 of G (rank g of G owns the channels g, g + G, ...; G | 16).  With
// k = k1 + 16 k2 ownership only depends on k1, and W16^(a k1) = W16^(a g) W16^(a k1'), k1 = g + k1': the factor W16^(a g) is a
// constant of polyphase branch j = 16 a + b1 and rides on its pre-mix phasor for free, after which the shard needs the pass-1
// outputs k1' = 0, G, 2G, ... only (the radix-16 butterflies prune themselves: G is a template parameter), 16 / G rows of Z per
// frame, and 256 / G radix-16 butterflies + 4096 / G freqdem samples + stores per tile in pass 2.  Those are spread over ALL
// 256 threads: thread (q, f2), q = s NK1 + j1, takes row k1' = G j1 and the slice s of the 16 output slots, and the slice is
// wave-uniform (G = 8: up to one lane bit), so that every wave runs a pass-2 butterfly pruned to its own quad of slots.
template <bool FM, int G>
__global__ __launch_bounds__(256, 2) void k_run256v2(V2Args VA)
{
    static_assert(G == 1 || G == 2 || G == 4 || G == 8, "interleaved shards: G = 2, 4, 8");
    constexpr int NK1 = 16 / G;                         // owned pass-1 rows per frame = output slots per slice
    const RunArgs &RA = VA.r;
    const TileArgs &A = RA.t;
    __shared__ __attribute__((aligned(16))) float2 L[V2_F2];
    float2 *R = L, *ST = L + V2_STASH, *Tt = ST + 256, *red = Tt + 16;
    float2 *tw_s = red + 16;
    float2 *TRc = tw_s + 256, *Ec = TRc + 256;          // V2_COLSCAN: run totals, run carries of the tile in work
    (void)TRc; (void)Ec;
    float2 *H = L + V2_BUF;                             // run start only: the halo tile's image, then scratch of the one-frame DFT (buffer 1 is free until the first tile's B_a)
    float2 *E = ST;                                     // run start only: 256 run carries (the stash area, before the stash is initialised)
    const int tid = threadIdx.x, j = tid;
    const unsigned w = blockIdx.x;
    unsigned first, last;
    run_range(RA.split, w, first, last);
    if (RA.pair_align) {                                // runs start on even tiles: a tile pair fills whole 128-byte lines of the F32 rows
        first &= ~1u;
        if (last != A.nb) last &= ~1u;
    }
    const float4 *x4 = reinterpret_cast<const float4 *>(A.x);
    const int col_off = 16 * (j >> 4) + 2 * (((j & 15) >> 1) ^ V2_RSW(j >> 4)) + (j & 1);
    const unsigned goff = V2_RSWN ? dma_offset_rsw(tid) : dma_offset(tid);
    const unsigned wave_u = (unsigned)__builtin_amdgcn_readfirstlane(tid >> 6);
    // wave w's DMA instruction `it` fills the 1 KiB half (w & 1) of frame 2 it + (w >> 1)
    const unsigned lds_wave = (unsigned)(size_t)(__attribute__((address_space(3))) float2 *)R + V2_FSB * (wave_u >> 1) + 1024u * (wave_u & 1u);

#ifndef V2_TRACE
#define V2_TRACE 0      // 1: the phase stamps of tools/trace_tiles.py (CSDR_TRACE=1 / 2) are compiled in -- a variant build (tools/build_variant.sh trace
                        // kernels_fused_v2.hip -DV2_TRACE=1): in the product kernel their scalar conditions cost the tile loop SGPRs it needs (round 5)
#endif
#if V2_TRACE
#define V2LSTAMP(i) do { if (A.trace && RA.trace_light && tid == 0) A.trace[(size_t)first * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define V2LSTAMP(i) do { } while (0)
#endif
    V2LSTAMP(0);
    // Everything the run start waits for is requested up front, so that the prologue is one burst of memory traffic and not
    // a chain of round trips: the first tile (buffer 0) and the halo tile (buffer 1) by DMA, the table values as plain loads.
    // cold run start: DC state from a read-only warm-up window, FIR window and freqdem history from the halo tile.  Run 0
    // instead carries the exact state of the previous call -- unless the call is INDEPENDENT of the previous launch
    // (RA.indep, the pipelined entry point csdr_chain_submit_device): then run 0 is a cold start like any other, its
    // window being the previous chunk's last WU + 1 tiles, kept in RA.prev_tail (tiles -1 .. -(WU + 1) of this chunk).
    const bool cold = w > 0 || RA.indep;
    const int halo = (int)first - 1;
    auto tile_ptr = [&](int t) -> const float4 * { return t >= 0 ? x4 + (size_t)t * 2048 : RA.prev_tail + (size_t)(WU + 1 + t) * 2048; };
    if (first < last) dma_tile<false, 2 * V2_FSB>(x4 + (size_t)first * 2048, goff, lds_wave);
    if (cold) dma_tile<false, 2 * V2_FSB>(tile_ptr(halo), goff, lds_wave + (unsigned)(V2_BUF * 8));
    float h[P];
#pragma unroll
    for (int n = 0; n < P; n++) h[n] = A.taps[(M256 - 1 - j) + n * M256];
    float2 Wa = A.wpre[(A.parity0 & 1) * M256 + j], Wb = A.wpre[((A.parity0 & 1) ^ 1) * M256 + j];
    if (G > 1) {                                        // W16^(a g), a = j >> 4: one of the sixteen 16th roots of unity, exact constants
        const int n = ((j >> 4) * (int)A.c0) & 15;
        const float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
        const int m4 = n & 3, qd = n >> 2;              // W16^n = (-j)^qd (cos - j sin)(2 pi m4 / 16)
        const float2 e = m4 == 0 ? make_float2(1.f, 0.f) : (m4 == 1 ? make_float2(C1, -S1) : (m4 == 2 ? make_float2(R2, -R2) : make_float2(S1, -C1)));
        const float2 rot = qd == 0 ? e : (qd == 1 ? make_float2(e.y, -e.x) : (qd == 2 ? make_float2(-e.x, -e.y) : make_float2(-e.y, e.x)));
        Wa = cmul(Wa, rot); Wb = cmul(Wb, rot);
    }
    tw_s[tid] = A.tw[tid];
    float2 wa[NB], wb[NB];                              // FIR window halves: one holds the previous tile, the other the new one
#pragma unroll
    for (int f = 0; f < NB; f++) { wa[f] = make_float2(0.f, 0.f); wb[f] = make_float2(0.f, 0.f); }
    float2 c;                                           // DC state v before the next tile (same in every lane)
    float2 w2 = make_float2(0.f, 0.f);

    // ------------------------------------------------------------------ run start (as k_run256)
    if (!cold) {
        c = A.vend_in[0];
#pragma unroll
        for (int f = 3; f < NB; f++) wa[f] = A.yhist_in[(f - 3) * M256 + j];
    } else {
        // (RA.nowu: no warm-up window -- the halo tile starts from state 0 unless it is the call's first tile, whose state is the carried one)
        const int h0 = RA.nowu ? halo : (RA.indep ? halo - (int)RA.wu : (halo > (int)RA.wu ? halo - (int)RA.wu : 0));
        const unsigned nwu = (unsigned)(halo - h0);
        float4 raw[8];
        // weight of my piece `it` of a tile: beta^(4095 - n), n = n0 + 512 it (the swizzle term of the slot does not depend on it): two
        // registers and a uniform ratio instead of sixteen (the six-tile batch below needs the room)
        float wt0, wt1;
        {
            const int wave = tid >> 6, lane = tid & 63;
            const int slot = 64 * wave + lane, q = slot >> 3;
            const int i = (slot & 7) ^ ((q >> 1) & 7);
            const int n = 16 * q + 2 * i;
            wt0 = exp2f((float)(4095 - n) * RA.l2beta);
            wt1 = exp2f((float)(4094 - n) * RA.l2beta);
        }
        const float wstep = RA.l2beta < -100.0f ? 0.0f : exp2f(-512.0f * RA.l2beta);   // beta^-512 (dc_block off, l2beta = -1000: no state to warm up, and no inf * 0)
        // zero-state aggregate of the warm-up tiles: sum over tiles of beta^(4096 (halo - 1 - t)) x (weighted sum inside tile t),
        // order-free, so that run w may start at tile (w mod nwu) of its window: 512 runs that walk their windows in the same order
        // hit the HBM channels in lockstep
        float2 acc = make_float2(0.f, 0.f);
        auto fold = [&](const float4 (&r)[8], int t) {
            float2 p = make_float2(0.f, 0.f);
            float a0 = wt0, a1 = wt1;
#pragma unroll
            for (int it = 0; it < 8; it++) {
                p = cfma(make_float2(r[it].x, r[it].y), a0, p);
                p = cfma(make_float2(r[it].z, r[it].w), a1, p);
                a0 *= wstep; a1 *= wstep;
            }
            acc = cfma(p, exp2f((float)(4096 * (halo - 1 - t)) * RA.l2beta), acc);
        };
        const unsigned rot = RA.wu_rot ? w : 0u;
        unsigned i = 0;
        if (nwu == 6 && RA.wu_batch6) {
            // the usual window: all six tiles requested before the first is folded (192 of the prologue's registers: nothing else is
            // live yet) -- one memory latency instead of two behind the DMA'd halo and first tile
            float4 rb[8], rc[8], rd[8], re[8], rf[8];
            const int t0 = h0 + (int)(rot % 6u), t1 = h0 + (int)((1 + rot) % 6u), t2 = h0 + (int)((2 + rot) % 6u);
            const int t3 = h0 + (int)((3 + rot) % 6u), t4 = h0 + (int)((4 + rot) % 6u), t5 = h0 + (int)((5 + rot) % 6u);
            tile_load(tile_ptr(t0), 256, raw, tid); tile_load(tile_ptr(t1), 256, rb, tid); tile_load(tile_ptr(t2), 256, rc, tid);
            tile_load(tile_ptr(t3), 256, rd, tid); tile_load(tile_ptr(t4), 256, re, tid); tile_load(tile_ptr(t5), 256, rf, tid);
            fold(raw, t0); fold(rb, t1); fold(rc, t2); fold(rd, t3); fold(re, t4); fold(rf, t5);
            i = 6;
        } else if (nwu >= 3) {
            float4 rb[8], rc[8];
#pragma unroll 1
            for (; i + 3 <= nwu; i += 3) {
                const int t0 = h0 + (int)((i + rot) % nwu), t1 = h0 + (int)((i + 1 + rot) % nwu), t2 = h0 + (int)((i + 2 + rot) % nwu);
                tile_load(tile_ptr(t0), 256, raw, tid);
                tile_load(tile_ptr(t1), 256, rb, tid);
                tile_load(tile_ptr(t2), 256, rc, tid);
                fold(raw, t0); fold(rb, t1); fold(rc, t2);
            }
        }
        for (; i < nwu; i++) {
            const int t0 = h0 + (int)((i + rot) % nwu);
            tile_load(tile_ptr(t0), 256, raw, tid);
            fold(raw, t0);
        }
        float2 ch = wg_sum(acc, red, tid);
        V2LSTAMP(1);
        if (h0 == 0 && !RA.indep) ch = cfma(A.vend_in[0], exp2f((float)(4096 * halo) * RA.l2beta), ch);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the two DMA'd tiles (older than every warm-up load)
        __syncthreads();
        scan_staged_fs<V2_FS2, V2_RSWN != 0>(H, E, Tt, A, tid);
#pragma unroll
        for (int f = 3; f < NB; f++) wa[f] = H[V2_FS2 * f + col_off];
        w2 = H[V2_FS2 * 2 + col_off];                      // FM: the 14th tap of the halo tile's last frame (freqdem history of the run)
        const float kj = -A.alpha * A.bj[j & 15];
        const float br = A.b16[tid & 15], bf = A.b256[tid >> 4];
        float2 vb, ve;
        frame_carries(Tt, A, tid, vb, ve);
        E[tid] = cfma(cfma(ch, bf, vb), br, E[tid]);
        c = cfma(ch, A.b256[16], ve);
        __syncthreads();
#pragma unroll
        for (int f = 3; f < NB; f++) wa[f] = cfma(E[16 * f + (j >> 4)], kj, wa[f]);
        w2 = cfma(E[16 * 2 + (j >> 4)], kj, w2);
    }
#pragma unroll
    for (int f = 3; f < NB; f++) wa[f] = cmul(wa[f], (f & 1) ? Wb : Wa);       // the window holds pre-mixed samples
    __syncthreads();                                    // every thread has read its run carries out of the stash area
    // freqdem history: stash[k1][i] = last Y frame of channel k1 + 16 XIDX(i)
    // (interleaved shard: rp_in / rp_out keep FULL-BAND indices, channel k = tid, so that the whole-band tile kernel can finish a
    // ragged call; channel k = (g + G j1) + 16 k2 lives in slot i = XIDX(k2) of thread row q = (i / NK1) NK1 + j1)
    const int st_i = XIDX(tid >> 4), st_j1 = ((tid & 15) - (int)(G > 1 ? A.c0 : 0u)) / G;
    const bool st_own = G == 1 || ((((tid & 15) - (int)A.c0) & (G - 1)) == 0 && (tid & 15) >= (int)A.c0);
    const int st_idx = G == 1 ? (tid & 15) * 16 + st_i : (st_own ? ((st_i / NK1) * NK1 + st_j1) * 16 + st_i : 0);
    if (st_own) ST[st_idx] = (FM && !cold) ? A.rp_in[tid] : make_float2(0.f, 0.f);
    __syncthreads();                                    // H free; stash and twiddles visible
    if (FM && cold) {
        // The run's first freqdem sample needs the frame in front of it: the halo tile's last frame goes through the FIR and
        // a one-frame DFT here (same arithmetic as the tile loop: pass 1 thread b1, pass 2 thread k1), instead of a fix-up
        // kernel after the launch patching one float into every row.
        w2 = cmul(w2, Wa);
        v2f acc = {0.f, 0.f};
#pragma unroll
        for (int n = P - 1; n >= 0; n--) {
            const float2 s2 = (n == P - 1) ? w2 : wa[NB - 1 - n];
            acc = __builtin_elementwise_fma((v2f){s2.x, s2.y}, (v2f){h[n], h[n]}, acc);
        }
        H[j] = to_f2(acc);
        __syncthreads();
        v2f vv[16];
        if (tid < 16) {
#pragma unroll
            for (int a = 0; a < 16; a++) vv[a] = to_v(H[16 * a + tid]);
            fft16_v(vv);
#pragma unroll
            for (int i = (G > 1 ? 0 : 1); i < 16; i++) vv[i] = cmul_v(vv[i], to_v(tw_s[16 * ((XIDX(i) + (G > 1 ? (int)A.c0 : 0)) & 15) + tid]));
#pragma unroll
            for (int i = 0; i < 16; i++) H[256 + 16 * XIDX(i) + tid] = to_f2(vv[i]);     // Z[k1'][b1], k1 = g + k1'
        }
        __syncthreads();
        if (tid < 16) {                                 // thread row q = tid: pass-1 row k1' = G (q mod NK1); every slot is written, the row's own slice is read later
#pragma unroll
            for (int b = 0; b < 16; b++) vv[b] = to_v(H[256 + 16 * (G * (tid % NK1)) + b]);
            fft16_v(vv);                                // vv[i] = Y[k1 + 16 XIDX(i)]
#pragma unroll
            for (int i = 0; i < 16; i++) ST[tid * 16 + i] = to_f2(vv[i]);
            if (G == 1 && RA.nowu) {                    // frame -1 of the run, channels 126..129 = (k1, k2) = (14, 7), (15, 7), (0, 8), (1, 8): slots 13 and 2
                if (tid >= 14) RA.side[((size_t)w * 4 + (tid - 14)) * DCFIX_F] = to_f2(vv[13]);
                else if (tid <= 1) RA.side[((size_t)w * 4 + 2 + tid) * DCFIX_F] = to_f2(vv[2]);
            }
        }
        __syncthreads();
    }

    // ------------------------------------------------------------------ per-thread constants of the tile loop
    const v2f Wav = to_v(Wa), Wbv = to_v(Wb);
    const float na = opaque_v(-A.alpha), be = opaque_v(A.beta);
    const float kJ = -A.alpha * exp2f((float)j * RA.l2beta);                    // -alpha beta^j: frame state into column j
    const float b256 = A.b256[1];
    const FmK2 fk = {{opaque_v(9.999993443e-01f), opaque_v(-3.332985938e-01f), opaque_v(1.994656026e-01f), opaque_v(-1.390860826e-01f),
                      opaque_v(9.642146528e-02f), opaque_v(-5.591168255e-02f), opaque_v(2.186254039e-02f), opaque_v(-4.054457881e-03f)},
                     opaque_v(1e-37f), opaque_v(A.fm_ref), opaque_v(RA.pk.hp), opaque_v(RA.pk.pi)};
    // LDS byte offsets inside a tile buffer that do not change from tile to tile
    const int q = tid, sw = V2_RSW(q & 15);
    const unsigned raw_a = (unsigned)q * 128u + (unsigned)(q >> 4) * (V2_FSB - 2048u) + ((unsigned)sw << 4);   // slot i of my run: raw_a ^ (i << 4)
    const int f1 = tid >> 4, b1 = tid & 15;                                     // pass 1: frame, column digit
    const int k1 = tid >> 4, f2 = tid & 15;                                     // pass 2 / tail: channel digit, frame
    const unsigned x_a = (unsigned)f1 * V2_FSB + (unsigned)b1 * 8u;             // X[f1][16 a + b1]: (x_a ^ (V2_RSW(a) << 4)) + 128 a
    const unsigned zw_a = (unsigned)f1 * V2_FSB + (unsigned)((((b1 >> 1) ^ V2_ZSW(f1)) << 1) | (b1 & 1)) * 8u;   // Z[f1][k1][b1]: + 128 k1
    const unsigned z_a = (unsigned)f2 * V2_FSB + (unsigned)k1 * 128u + ((unsigned)V2_ZSW(f2) << 4);             // pair i of Z[f2][k1][.]: z_a ^ (i << 4)
    // interleaved shard: thread row q = k1 = s NK1 + j1; output row of channel (g + G j1) + 16 k2 in the shard's [M / G][nf] plane: j1 + NK1 k2
    const int j1 = k1 % NK1, hb = (G == 8) ? ((k1 >> 1) & 1) : 0;              // G = 8: lane bit of the slice (slots S0 + 2 hb, S0 + 2 hb + 1)
    const unsigned z_a_g = (unsigned)f2 * V2_FSB + (unsigned)(G * j1) * 128u + ((unsigned)V2_ZSW(f2) << 4);
    const uint32_t voff = ((uint32_t)(G == 1 ? k1 : j1 + 16 * hb) * A.out_stride + A.out_t0 + (uint32_t)f2) * (FM ? 4u : 8u);  // + NK1 k2 rows, + 16 b frames
    const size_t row16 = (size_t)NK1 * A.out_stride * (FM ? 4u : 8u);
#if V2_TRACE
#define V2STAMP(i) do { if (A.trace && !RA.trace_light && tid == 0) A.trace[(size_t)b_ * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define V2STAMP(i) do { } while (0)
#endif

    // FM: the even tile of a pair keeps its 16 results per thread and the odd tile stores both, so that the two 64-byte halves of
    // a row's 128-byte line reach the L2 back to back and leave it as ONE write (halves that arrive a tile apart are evicted
    // separately under the streaming reads: F32 output then costs as much HBM write energy as CF32 output of twice the size)
    float hold[16];
    v2f holdc[V2_PAIR_CF ? 16 : 1];
    auto tile = [&](float2 (&old)[NB], float2 (&nw)[NB], unsigned b_, const int par) {
        unsigned b = (unsigned)__builtin_amdgcn_readfirstlane((int)b_);            // keep the tile index (store / DMA bases) in SGPRs
        asm volatile("" : "+s"(b));
        char *B = reinterpret_cast<char *>(L) + par * (V2_BUF * 8);             // this tile's buffer
        float2 *Bf = reinterpret_cast<float2 *>(B);
        if (RA.prio_div) {
            // the CU issues oldest-wave-first: without this the older of a CU's two workgroups finishes its run ~35 % earlier
            // and the CU is half empty for the rest of the launch; alternating the priority per tile shares the issue slots
            if (((b_ - first) + w / RA.prio_div) & 1u) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        }
        V2STAMP(0);
        if (G == 1 && b + 1 == last && tid == 0) {      // the DC state in front of the next run's halo tile (launches without warm-up windows)
            if (kernarg_u32_s<offsetof(V2Args, r.nowu)>()) kernarg_ptr_s<offsetof(V2Args, r.cpre)>()[w + 1] = c;
        }
        if (V2_TRACE && A.trace && !RA.trace_light && tid == 0) A.trace[(size_t)b_ * 16 + 15] = __builtin_amdgcn_s_memrealtime();
        bar();                                          // B_a: the tile image has landed (every wave waited for its own DMA); the other buffer is free
        V2STAMP(1);
#ifdef V2_SELECTIVE_POLICY
        // A/B builds (verdict r03 #1 (ii)): the last WU + 1 tiles of a run are the next run's warm-up window + halo, already fetched once
        // at the launch's start; every other tile is read exactly once per launch and gets the streaming policy
        if (!(V2_ABLATE & 1) && b + 1 < last) {
            if (b + 1 + (WU + 1) < last) dma_tile<true, 2 * V2_FSB>(x4 + (size_t)(b + 1) * 2048, goff, lds_wave + (unsigned)(par ^ 1) * (V2_BUF * 8u));
            else dma_tile<false, 2 * V2_FSB>(x4 + (size_t)(b + 1) * 2048, goff, lds_wave + (unsigned)(par ^ 1) * (V2_BUF * 8u));
        }
#else
        if (!(V2_ABLATE & 1) && b + 1 < last) dma_tile<false, 2 * V2_FSB>(x4 + (size_t)(b + 1) * 2048, goff, lds_wave + (unsigned)(par ^ 1) * (V2_BUF * 8u));
#endif
#if V2_COLSCAN
        // ---- DC blocker on the column-layout registers: thread j reads its raw column once; a run of 16 consecutive samples is one
        // 16-lane DPP row of one frame, so the zero-state scan inside a run is four v_fmac_dpp steps per component (col_run_scan);
        // the run totals cross the frame through 2 KiB of LDS (col_run_carries: carry into every run from the earlier runs of its
        // frame, frame totals), the frame states V[f] chain in registers as before
#pragma unroll
        for (int f = 0; f < NB; f++) nw[f] = Bf[V2_FS2 * f + col_off];
        col_run_scan(nw, TRc, A, tid);
        V2STAMP(2);
        bar();                                          // B_c: run totals visible
        col_run_carries(TRc, Ec, Tt, A, tid);
        bar();                                          // B_c2: run carries and frame totals visible
        V2STAMP(3);
        {
            v2f V = to_v(c);
            const float kj = -A.alpha * A.bj[j & 15];
            const v2f kJv = {kJ, kJ}, bv = {b256, b256}, kjv = {kj, kj};
#pragma unroll
            for (int f = 0; f < NB; f++) {
                v2f y = __builtin_elementwise_fma(V, kJv, to_v(nw[f]));
                nw[f] = to_f2(__builtin_elementwise_fma(to_v(Ec[16 * f + (j >> 4)]), kjv, y));
                V = __builtin_elementwise_fma(V, bv, to_v(Tt[f]));
            }
            c = to_f2(V);
        }
#else
        // ---- DC blocker inside a frame: thread q owns the run of 16 consecutive samples q.  First the run total (zero state),
        // a decayed DPP row scan over the 16 runs of the frame gives the state e at my run's start (frame state still
        // missing: it is added in column layout below), then the blocker itself from that state
        v4f xr[8];
        float2 s = make_float2(0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            xr[i] = *reinterpret_cast<const v4f *>(B + (raw_a ^ (unsigned)(i << 4)));
            if (V2_ABLATE & 32) { s.x += xr[i].x; continue; }
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
            if (V2_ABLATE & 32) { *reinterpret_cast<v4f *>(B + (raw_a ^ (unsigned)(i << 4))) = xr[i]; continue; }
            y.x = fmaf(s.x, na, xr[i].x); y.y = fmaf(s.y, na, xr[i].y);
            s = make_float2(fmaf(s.x, be, xr[i].x), fmaf(s.y, be, xr[i].y));
            y.z = fmaf(s.x, na, xr[i].z); y.w = fmaf(s.y, na, xr[i].w);
            s = make_float2(fmaf(s.x, be, xr[i].z), fmaf(s.y, be, xr[i].w));
            if (V2_ABLATE & 64) asm volatile("" :: "v"(y));            // timing only: y' is not written back (one LDS pass less)
            else *reinterpret_cast<v4f *>(B + (raw_a ^ (unsigned)(i << 4))) = y;
        }
        V2STAMP(2);
        bar();                                          // B_c: y' (frame carry still missing) and the frame totals are visible
        V2STAMP(3);
        // ---- column layout: thread j owns branch j; frame state chain V[f] (uniform), y = y' - alpha beta^j V[f], pre-mix
#pragma unroll
        for (int f = 0; f < NB; f++) nw[f] = Bf[V2_FS2 * f + col_off];
        {
            v2f V = to_v(c);
            const v2f kJv = {kJ, kJ}, bv = {b256, b256};
#pragma unroll
            for (int f = 0; f < NB; f++) {
                nw[f] = to_f2(__builtin_elementwise_fma(V, kJv, to_v(nw[f])));
                V = __builtin_elementwise_fma(V, bv, to_v(Tt[f]));
            }
            c = to_f2(V);
        }
#endif
        if (b + 1 == A.nb) {                            // the stream's last 13 frames of y
#pragma unroll
            for (int f = 3; f < NB; f++) A.yhist_out[(f - 3) * M256 + j] = nw[f];
        }
#pragma unroll
        for (int f = 0; f < NB; f += 2) {
            v2f a0 = to_v(nw[f]), a1 = to_v(nw[f + 1]);
            cmul2_v(a0, Wav, a1, Wbv);
            nw[f] = to_f2(a0); nw[f + 1] = to_f2(a1);
        }
        V2STAMP(4);
        // ---- polyphase FIR on the pre-mixed window, four frames at a time (independent accumulators); X goes where the
        // thread's column came from
#pragma unroll
        for (int f0 = 0; f0 < NB; f0 += 4) {
            v2f acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
            for (int n = ((V2_ABLATE & 8) ? 0 : P - 1); n >= 0; n--) {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int i = f0 + u - n;
                    const float2 s2 = (i >= 0) ? nw[i] : old[NB + i];
                    const v2f sv = {s2.x, s2.y}, hv = {h[n], h[n]};
                    acc[u] = __builtin_elementwise_fma(sv, hv, acc[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) Bf[V2_FS2 * (f0 + u) + col_off] = to_f2(acc[u]);
        }
        V2STAMP(5);
        bar();                                          // B_d: X complete
        V2STAMP(6);
        // ---- DFT pass 1: thread (f1, b1)
        v2f vv[16];
#pragma unroll
        for (int a = 0; a < 16; a++) vv[a] = to_v(*reinterpret_cast<const float2 *>(B + (x_a ^ (unsigned)(V2_RSW(a) << 4)) + 128 * a));
        if (!(V2_ABLATE & 16)) fft16_v(vv);
        if (G == 1) {
#pragma unroll
            for (int i = 1; i < 16; i++) vv[i] = cmul_v(vv[i], to_v(tw_s[16 * XIDX(i) + b1]));
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++)                // rows k1' = 0, G, 2G, ... only; twiddle of the true k1 = g + k1'
                if (XIDX(i) % G == 0) vv[i] = cmul_v(vv[i], to_v(tw_s[16 * ((XIDX(i) + (int)A.c0) & 15) + b1]));
        }
        V2STAMP(7);
        // no barrier here: Z[f1] goes into frame f1's own 2 KiB of the buffer, which only the 16 lanes that have just read
        // X[f1] (same wave, program order) ever touched since B_d
        if (V2_BAR_E) bar();
        V2STAMP(8);
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (V2_ABLATE & 128) { asm volatile("" :: "v"(vv[i])); continue; }     // timing only: Z is not written (one LDS pass less)
            if (XIDX(i) % G == 0) *reinterpret_cast<float2 *>(B + zw_a + 128 * XIDX(i)) = to_f2(vv[i]);
        }
        V2STAMP(9);
        bar();                                          // B_f: Z complete
        V2STAMP(10);
        if constexpr (G > 1) {
            // ---- interleaved shard: pass 2 + tail of wave `Wv` (compile time): slots S0 .. S0 + NSL - 1 of row k1' = G j1
            auto shard_tail = [&](auto WC) {
                constexpr int Wv = decltype(WC)::value;
                constexpr int S0 = (G == 2) ? 8 * (Wv >> 1) : 4 * Wv;
                constexpr int NSL = (G == 2) ? 8 : 4;           // slots computed per thread (G = 8: the lane keeps two of the four)
                constexpr int NS = (G == 8) ? 2 : NSL;          // slots demodulated and stored per thread
                v2f vz[16];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const v4f v = *reinterpret_cast<const v4f *>(B + (z_a_g ^ (unsigned)(i << 4)));
                    vz[2 * i] = (v2f){v.x, v.y}; vz[2 * i + 1] = (v2f){v.z, v.w};
                }
                fft16_v(vz);                                    // only slots S0 .. S0 + NSL - 1 are used: the rest of the butterfly is dead code
                v2f y[NS];
#pragma unroll
                for (int u = 0; u < NS; u++) {
                    if (G == 8) { y[u].x = hb ? vz[S0 + 2 + u].x : vz[S0 + u].x; y[u].y = hb ? vz[S0 + 2 + u].y : vz[S0 + u].y; }
                    else y[u] = vz[S0 + u];
                }
                // slot u of mine is register slot S0 (+ 2 hb) + u: k2 = XIDX(slot); the lane part of the row (j1, hb) is in voff
                float2 *stp = ST + k1 * 16 + S0 + (G == 8 ? 2 * hb : 0);
                char *obase = reinterpret_cast<char *>(A.out) + (size_t)b * RA.tile_step;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // next tile image (issued a tile ago)
                if (FM) {
                    float2 rp[NS], rr[NS];
#pragma unroll
                    for (int u = 0; u < NS; u += 2) {
                        const v4f sp = *reinterpret_cast<const v4f *>(stp + u);      // previous tile's last frame (lane f2 = 0 uses it)
                        rp[u] = make_float2(dpp_keep<0x111>(sp.x, y[u].x), dpp_keep<0x111>(sp.y, y[u].y));
                        rp[u + 1] = make_float2(dpp_keep<0x111>(sp.z, y[u + 1].x), dpp_keep<0x111>(sp.w, y[u + 1].y));
                        rr[u] = to_f2(y[u]); rr[u + 1] = to_f2(y[u + 1]);
                    }
                    float mq_[NS];
                    float (&mq)[NS] = (V2_PAIR && par == 0) ? *reinterpret_cast<float (*)[NS]>(&hold[0]) : mq_;
                    if (NS >= 4) {
#pragma unroll
                        for (int u = 0; u < NS; u += 4) {
                            const float2 (&rp4)[4] = *reinterpret_cast<const float2 (*)[4]>(&rp[u]);
                            const float2 (&rr4)[4] = *reinterpret_cast<const float2 (*)[4]>(&rr[u]);
                            fm_quad(rp4, rr4, fk, *reinterpret_cast<float (*)[4]>(&mq[u]));
                        }
                    } else {
                        const FmK k1s = {fk.tiny, fk.ref, fk.hp, fk.pi};
#pragma unroll
                        for (int u = 0; u < NS; u++) mq[u] = fm_sample(rp[u], rr[u], k1s);
                    }
                    if (!(V2_PAIR && par == 0 && b + 1 < last)) {
#pragma unroll
                        for (int u = 0; u < NS; u++) {
                            // uniform part of the row: k2 of slot S0 + u without the lane's hb term (G = 8: XIDX(S0 + 2 hb + u) = Wv + 8 hb + 4 u)
                            const char *rowp = obase + (size_t)XIDX(S0 + u) * row16;
                            if (V2_PAIR && par == 1) asm volatile("s_nop 4\n\tglobal_store_dword %0, %1, %2 offset:-64" V2_STORE_AUX :: "v"(voff), "v"(hold[u]), "s"(rowp) : "memory");
                            asm volatile("s_nop 4\n\tglobal_store_dword %0, %1, %2" V2_STORE_AUX :: "v"(voff), "v"(mq[u]), "s"(rowp) : "memory");
                        }
                    }
                    if (f2 == 15) {
#pragma unroll
                        for (int u = 0; u < NS; u += 2) *reinterpret_cast<v4f *>(stp + u) = (v4f){y[u].x, y[u].y, y[u + 1].x, y[u + 1].y};
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < NS; u++) {
                        const char *rowp = obase + (size_t)XIDX(S0 + u) * row16;
                        asm volatile("s_nop 4\n\tglobal_store_dwordx2 %0, %1, %2\n\ts_nop 1" :: "v"(voff), "v"(y[u]), "s"(rowp) : "memory");
                    }
                }
            };
            if (G == 2) { if (wave_u < 2u) shard_tail(std::integral_constant<int, 0>{}); else shard_tail(std::integral_constant<int, 2>{}); }
            else switch (wave_u) {
                case 0: shard_tail(std::integral_constant<int, 0>{}); break;
                case 1: shard_tail(std::integral_constant<int, 1>{}); break;
                case 2: shard_tail(std::integral_constant<int, 2>{}); break;
                default: shard_tail(std::integral_constant<int, 3>{}); break;
            }
            V2STAMP(14);
            return;
        }
        // ---- DFT pass 2: thread (k1, f2) reads its 16 consecutive Z values as eight 16-byte pairs
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const v4f v = *reinterpret_cast<const v4f *>(B + (z_a ^ (unsigned)(i << 4)));
            vv[2 * i] = (v2f){v.x, v.y}; vv[2 * i + 1] = (v2f){v.z, v.w};
        }
        if (!(V2_ABLATE & 16)) fft16_v(vv);             // vv[i] = Y[k1 + 16 XIDX(i)] of frame f2
        V2STAMP(11);
        if (G == 1 && cold && b - first < 7u && kernarg_u32_s<offsetof(V2Args, r.nowu)>()) {   // (uniform) the run's first seven tiles: Y of the channels around DC for k_run256_dcfix
            // (the kernel arguments are re-read here, behind an opaque copy of the pointer: as loop invariants they cost the tile loop SGPRs it
            // does not have -- the asm stores below rely on a kernel without SGPR spills)
            float2 *sd = kernarg_ptr_s<offsetof(V2Args, r.side)>() + ((size_t)w * 4) * DCFIX_F + 1 + 16 * (size_t)(b - first) + f2;
            if (k1 >= 14) sd[(k1 - 14) * DCFIX_F] = to_f2(vv[13]);
            else if (k1 <= 1) sd[(2 + k1) * DCFIX_F] = to_f2(vv[2]);
        }
        // ---- tail
        char *obase = reinterpret_cast<char *>(A.out) + (size_t)b * RA.tile_step;
        if (FM) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // next tile image (issued a tile ago): nothing else is outstanding
#pragma unroll
            for (int i = 0; i < 16; i += 4) {
                float2 rp[4], rr[4];
#pragma unroll
                for (int u = 0; u < 4; u += 2) {
                    const v4f sp = *reinterpret_cast<const v4f *>(ST + k1 * 16 + i + u);  // previous tile's last frame (lane f2 = 0 uses it)
                    rp[u] = make_float2(dpp_keep<0x111>(sp.x, vv[i + u].x), dpp_keep<0x111>(sp.y, vv[i + u].y));
                    rp[u + 1] = make_float2(dpp_keep<0x111>(sp.z, vv[i + u + 1].x), dpp_keep<0x111>(sp.w, vv[i + u + 1].y));
                    rr[u] = to_f2(vv[i + u]); rr[u + 1] = to_f2(vv[i + u + 1]);
                }
                float mq_[4];
                float (&mq)[4] = (V2_PAIR && par == 0) ? *reinterpret_cast<float (*)[4]>(&hold[i]) : mq_;
                if (V2_ABLATE & 4) { mq[0] = rp[0].x + rr[0].y; mq[1] = rp[1].y + rr[1].x; mq[2] = rp[2].x + rr[2].y; mq[3] = rp[3].y + rr[3].x; }
                else if (V2_FM_PACKED) fm_quad(rp, rr, fk, mq);
                else {
                    const FmK k1s = {fk.tiny, fk.ref, fk.hp, fk.pi};
#pragma unroll
                    for (int u = 0; u < 4; u++) mq[u] = fm_sample(rp[u], rr[u], k1s);
                }
                if (V2_PAIR && par == 0 && b + 1 < last) continue;   // (uniform) the odd tile of the pair stores these
#pragma unroll
                for (int u = 0; u < 4; u++) {                       // stores go out between the quads
                    const char *rowp = obase + (size_t)XIDX(i + u) * row16;
                    if (V2_ABLATE & 2) asm volatile("" :: "v"(mq[u]), "s"(rowp));
                    else {
                        if (V2_PAIR && par == 1) asm volatile(V2_SNOP "global_store_dword %0, %1, %2 offset:-64" V2_STORE_AUX :: "v"(voff), "v"(hold[i + u]), "s"(rowp) : "memory");
                        asm volatile(V2_SNOP "global_store_dword %0, %1, %2" V2_STORE_AUX :: "v"(voff), "v"(mq[u]), "s"(rowp) : "memory");
                    }
                }
            }
            if (f2 == 15) {
#pragma unroll
                for (int i = 0; i < 16; i += 2) *reinterpret_cast<v4f *>(ST + k1 * 16 + i) = (v4f){vv[i].x, vv[i].y, vv[i + 1].x, vv[i + 1].y};
            }
            V2STAMP(12);
            V2STAMP(13);
        } else {
            V2STAMP(12);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            V2STAMP(13);
            if (V2_PAIR_CF && par == 0 && b + 1 < last) {           // CF32 pairs (experiment): 256 bytes of a row per tile pair
#pragma unroll
                for (int i = 0; i < 16; i++) holdc[i] = vv[i];
            } else {
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const char *rowp = obase + (size_t)XIDX(i) * row16;
                    if (V2_ABLATE & 2) asm volatile("" :: "v"(vv[i]), "s"(rowp));
                    else {
                        if (V2_PAIR_CF && par == 1) asm volatile(V2_SNOP "global_store_dwordx2 %0, %1, %2 offset:-128" V2_STORE_AUX :: "v"(voff), "v"(holdc[i]), "s"(rowp) : "memory");
                        asm volatile(V2_SNOP "global_store_dwordx2 %0, %1, %2" V2_STORE_AUX :: "v"(voff), "v"(vv[i]), "s"(rowp) : "memory");
                    }
                }
            }
        }
        V2STAMP(14);
    };

    V2LSTAMP(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the first tile was requested at the kernel's entry
    V2LSTAMP(3);
    for (unsigned b = first; b < last; b += 2) {
        tile(wa, wb, b, 0);
        if (b + 1 >= last) break;
        tile(wb, wa, b + 1, 1);
    }

    // ------------------------------------------------------------------ state after the run
    V2LSTAMP(4);
    bar();                                              // stash of the last tile visible to every wave
    if (FM) {
        const float2 lastY = ST[st_idx];
        if (last == A.nb && st_own) A.rp_out[tid] = lastY;
    }
    if (last == A.nb && tid == 0) A.vend_out[0] = c;
}

}  // namespace

// k_run256_dcfix: run w >= 1 of a launch that ran without warm-up windows (RunArgs::nowu).  Thread t = frame t - 1 of the run, t = 0 .. 112;
// Y = side + c_true x R[t][ch] for the channels 126..129, then the run's own tail for them (freqdem against the corrected frame in front).
template <bool FM>
__global__ __launch_bounds__(128) void k_run256_dcfix(V2Args VA, const float2 *__restrict__ Rt)
{
    const RunArgs &RA = VA.r;
    const TileArgs &A = RA.t;
    __shared__ float2 Ys[4][DCFIX_F + 1];
    const unsigned w = blockIdx.x + 1;
    unsigned first, last;
    run_range(RA.split, w, first, last);
    if (RA.pair_align) { first &= ~1u; if (last != A.nb) last &= ~1u; }
    if (first >= last || first <= 1) return;            // first == 1: the halo tile is the call's first tile, the run started from the carried state (h0 == 0 folds vend_in in): exact, and cpre[1] holds that same state -- correcting would add it twice
    const int t = threadIdx.x;
    const float2 c = RA.cpre[w];
    const float2 *R = Rt + (size_t)(A.parity0 & 1u) * DCFIX_F * 4;
    float2 y[4];
    if (t < DCFIX_F) {
#pragma unroll
        for (int ch = 0; ch < 4; ch++) {
            const float2 r = R[t * 4 + ch], s0 = RA.side[((size_t)w * 4 + ch) * DCFIX_F + t];
            y[ch] = make_float2(fmaf(c.x, r.x, fmaf(-c.y, r.y, s0.x)), fmaf(c.x, r.y, fmaf(c.y, r.x, s0.y)));
            Ys[ch][t] = y[ch];
        }
    }
    __syncthreads();
    if (t < 1 || t >= DCFIX_F) return;
    const unsigned b = first + (unsigned)(t - 1) / 16u, f2 = (unsigned)(t - 1) & 15u;
    if (b >= last) return;
    char *obase = reinterpret_cast<char *>(A.out) + (size_t)b * RA.tile_step;
    const FmK fk = {1e-37f, A.fm_ref, RA.pk.hp, RA.pk.pi};
#pragma unroll
    for (int ch = 0; ch < 4; ch++) {
        const size_t off = ((size_t)(126 + ch) * A.out_stride + A.out_t0 + f2) * (FM ? 4u : 8u);
        if (FM) *reinterpret_cast<float *>(obase + off) = fm_sample(Ys[ch][t - 1], y[ch], fk);
        else *reinterpret_cast<float2 *>(obase + off) = y[ch];
    }
}

int run256_dcfix_launch(const void *run_args, bool fm, unsigned nruns, const float2 *Rt, hipStream_t s)
{
    if (nruns < 2) return 0;
    V2Args VA;
    VA.r = *static_cast<const RunArgs *>(run_args);
    if (fm) hipLaunchKernelGGL(k_run256_dcfix<true>, dim3(nruns - 1), dim3(128), 0, s, VA, Rt);
    else hipLaunchKernelGGL(k_run256_dcfix<false>, dim3(nruns - 1), dim3(128), 0, s, VA, Rt);
    return 0;
}

static V2Args make_v2(const void *run_args)
{
    V2Args VA;
    VA.r = *static_cast<const RunArgs *>(run_args);
    return VA;
}

int run256_v2_launch(const void *run_args, bool fm, unsigned G, unsigned nruns, hipStream_t s)
{
    const V2Args VA = make_v2(run_args);
#define V2_LAUNCH(F, GG) hipLaunchKernelGGL((k_run256v2<F, GG>), dim3(nruns), dim3(256), 0, s, VA)
    if (G == 1) { if (fm) V2_LAUNCH(true, 1); else V2_LAUNCH(false, 1); }
    else if (G == 2) { if (fm) V2_LAUNCH(true, 2); else V2_LAUNCH(false, 2); }
    else if (G == 4) { if (fm) V2_LAUNCH(true, 4); else V2_LAUNCH(false, 4); }
    else if (G == 8) { if (fm) V2_LAUNCH(true, 8); else V2_LAUNCH(false, 8); }
    else { set_error("k_run256v2: interleaved shards of stride %u are not built (2, 4, 8)", G); return -1; }
#undef V2_LAUNCH
    return 0;
}

int run256_v2_blocks_per_cu(bool fm)
{
    int occ = 0;
    if (fm) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (k_run256v2<true, 1>), 256, 0);
    else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (k_run256v2<false, 1>), 256, 0);
    return occ < 1 ? 1 : occ;
}

}  // namespace csdr
