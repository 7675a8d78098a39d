// Third-generation run kernel of the fused M = 1024 FM chain (BASELINE configs[3] shape): ONE 512-thread workgroup per compute
// unit, split by wave role, the output staged in REGISTERS until a row's whole 128-byte line is complete.
//
//   raw CF32 x --DC blocker--> y --NCO pre-mix, 14-tap polyphase FIR--> X_t[j] --1024-point forward DFT (16 x 16 x 4)--> Y_t[k]
//              --per-channel freqdem--> out[1024][nf] F32           (8 B read + 4 B written per sample, Liquid.chs:575-589,
//                                                                     828-862, 324-328)
//
// Why (verdict r03 #4).  A 4-frame tile holds 16 bytes of each of the 1024 F32 rows; a whole 128-byte line of every row is 8
// tiles = 128 KiB.  k_run1024v2 (two 256-thread workgroups per CU, 256 VGPRs each, 2 x 80 KiB of LDS: nothing left on chip)
// parks those 128 KiB per workgroup in a global staging block and reads them back transposed: + 8 B per sample through the
// L2 -> fabric counters, 1.89x the algorithmic bytes with the run-start re-reads.  Here a CU runs ONE workgroup whose waves are
// a two-stage pipeline (the structure of k_run256v3), three tiles deep:
//   FRONT waves 0-3 (thread j = branches j + 256 q; wave f = frame f in pass 1): DC blocker, pre-mix, FIR of tile s -> X(s) in the
//                   tile's buffer; DFT pass 1 of tile s-1
//   BACK  waves 4-7 (wave f = frame f; thread kk): tile DMA; DFT pass 2 of tile s-1; pass 3 + freqdem of tile s-2; row stores
// The front waves carry the window -- a ring of 16 frames x 4 branches in registers, the step unrolled four times, so the window never
// moves (k_run1024v2: 52 v_mov per tile) -- and re-read their taps per step as v2 does; the back waves have no window, so the 8 tiles x 4
// channels x 4 frames a thread produces per block wait in 128 VGPRs.  When the block is complete a wave turns its 64 rows x 128 bytes
// through its own 8 KiB of the tile buffer it has just consumed (its pass-3 reads cover exactly 2 KiB of each frame block: no barrier)
// and stores whole lines, eight lanes per row.  No staging block, no read-back; runs are twice as long (one per CU), so the read-only
// run-start tiles halve as well.  Runs start and end on 8-tile blocks (the launcher takes calls of nf = 0 mod 32 frames).
//
// A step has two workgroup barriers:        front                                      back
//   bar P  ----------------------------------------------------------------------------------------------------------------
//          taps of branch 0; pass 1 of tile s-1; taps of branch 1;               DMA of image s+1 (pieces 0-3); pass 3 + freqdem of
//          DC scan of tile s (raw image -> y', group totals)                     tile s-2 [+ block flush]; DMA pieces 4-7
//   bar Q  ----------------------------------------------------------------------------------------------------------------
//          column layout, pre-mix, FIR -> X(s) (taps of branches 2, 3 on the way)   pass 2 of tile s-1; wait for image s+1
// Tile i lives in buffer i % 4 from its DMA to the back waves' pass 3.  Arithmetic: k_run1024v2's, instruction for instruction.
// Measured (CSDR_RUN1024_V3_TRACE on a -DB3_TRACE=1 build, tools/trace_run1024v3.py): a step is ~7000 cycles, the two roles within 5 %
// of each other in both phases, i.e. each SIMD's two waves issue back to back (~1550 VALU instructions per step and SIMD); the step of
// a completed block takes ~4000 more (128 KiB of row stores = 2048 cycles of the CU's address unit, behind the LDS turn).
// What did not help: starting the runs of a CU group in different steps so that the blocks do not complete together (+3 %: the burst is
// not an HBM problem), the tile DMA in the front waves' queue (in-order vmcnt: the first tap use then waits for the image).
#include "fused_v2_common.h"
#include <type_traits>

#ifndef B3_TRACE
#define B3_TRACE 0       // 1: s_memtime stamps of run 1's wave 0 / wave 4 per phase (12 KiB of LDS), written to Run1024v3Args::trace
#endif
#ifndef B3_ABLATE
#define B3_ABLATE 0      // timing experiments only (wrong results): 2 no output stores, 4 no freqdem, 8 row stores straight out of the registers (no LDS
                         // turn: the lines go to the wrong rows), 16 no DFT passes 1-2, 32 no FIR, 64 no pass 3 / tail at all
#endif

namespace csdr {
namespace {

constexpr int B3_M = 1024, B3_T4 = 4, B3_NBUF = 4;
constexpr int B3_TBF = 8, B3_TBC = 4;            // tiles per block = a row's 128-byte line: 32 F32 / 16 CF32 frames
constexpr int B3_BUF = 4096;                       // float2 per tile buffer (32 KiB)
constexpr int B3_TW1 = B3_NBUF * B3_BUF;           // twiddles W1024^(k1 b) at [k1 - 1][b], k1 = 1..15: 960
constexpr int B3_ST = B3_TW1 + 960;                // last Y frame of channel kk + 256 k3 at [kk][k3]: 1024 (the prologue's reduction scratch before that)
constexpr int B3_TT = B3_ST + 1024;                // 16 group totals
constexpr int B3_TW2 = B3_TT + 16;                 // pass-2 twiddles W64^(d k2) at [k2][d]: 64 (the three d of an instruction in three banks; tw1 has them 2 KiB apart)
constexpr int B3_F2 = B3_TW2 + 64;                 // 18 448 float2 = 147 584 B: one workgroup per CU
constexpr int B3_WU = 6, B3_HALO = 4;              // read-only warm-up tiles (DC state); 3 window-refill tiles + the muted tile in front of a run

struct Run1024v3Args {
    const float2 *x;            // raw input of this call
    void *out;                  // [1024][nf] F32 (FM) / CF32
    const float4 *taps_q;       // k_run1024v2's table: [4 q][4 pieces][256 j] taps + even-frame phasor; behind the 64 KiB: [4 q][256 j] odd-frame phasor
    const float2 *tw;           // e^{-j 2 pi i / 1024}
    const float2 *uhist_in; float2 *uhist_out;    // [13][1024] pre-mixed, DC-blocked window before / after the call
    const float2 *vend_in; float2 *vend_out;      // DC blocker state v1
    const float2 *rp_in; float2 *rp_out;          // [1024] freqdem r'
    uint32_t nf, nb, nruns, parity0;
    uint32_t tile_major;        // CF32: lines of a block back to back, [block][1024][128 B] (the plane k_agc_spec_tm reads), instead of rows [1024][nf]
    unsigned long long *trace;  // debug (CSDR_RUN1024_V3_TRACE=file): s_memtime stamps of run 1's wave 0 (front) and wave 4 (back), [role][step][4]
    float alpha, beta, l2beta, fm_ref, tiny;
    float b16[16];              // beta^(16 r)
    float b256[17];             // beta^(256 g)
    PhaseK pk;
    uint32_t nowu;              // 1: a run starts cold from DC state 0 (no read-only warm-up tiles), leaves the state in front of tile last - 4 in
    float2 *cpre, *side;        //    cpre[w + 1] and (FM) the uncorrected Y of the channels 510..513 of its frames -1 .. 31 in side; k_run1024_dcfix
};

// run w: blocks of TB tiles, evenly; the call's last block may be a partial one (nb % TB tiles: its rows get the front part of a line)
__host__ __device__ __forceinline__ void run3_bounds(uint32_t nb, uint32_t nruns, unsigned w, unsigned TB, unsigned &first, unsigned &last)
{
    const unsigned nblk = (nb + TB - 1) / TB;
    first = TB * (unsigned)((unsigned long long)w * nblk / nruns);
    last = TB * (unsigned)((unsigned long long)(w + 1) * nblk / nruns);
    if (last > nb) last = nb;
}

template <bool FM>
__global__ __launch_bounds__(512) void k_run1024v3(Run1024v3Args A)
{
    constexpr unsigned B3_TB = FM ? B3_TBF : B3_TBC;
    __shared__ __attribute__((aligned(16))) float2 L[B3_F2];
    __shared__ float2 cpre_s[8];                        // the DC state in front of the last eight steps' tiles: the one in front of tile last - 4 is the next run's cold start
    __shared__ unsigned long long trc[B3_TRACE ? 1536 : 1];            // debug stamps (8 KiB): collected in LDS, written out when the run is over (a global store per stamp would sit in the traced wave's vmcnt queue)
    float2 *tw1 = L + B3_TW1, *ST = L + B3_ST, *Tt = L + B3_TT, *red = ST;
    const int tid = threadIdx.x;
    const bool back = tid >= 256;                       // wave-uniform role
    const int lt = tid & 255, j = lt;                   // thread index inside the role; front: polyphase branches j + 256 q
    const unsigned w = blockIdx.x;
    unsigned first, last;
    run3_bounds(A.nb, A.nruns, w, B3_TB, first, last);
    const float4 *x4 = reinterpret_cast<const float4 *>(A.x);
    const int col_off = 16 * (j >> 4) + 2 * (((j & 15) >> 1) ^ (j >> 5)) + (j & 1);
    const unsigned goff = dma_offset(lt);
    const unsigned wave_u = (unsigned)__builtin_amdgcn_readfirstlane(lt >> 6);      // wave index inside the role
    const unsigned lds_wave = (unsigned)(size_t)(__attribute__((address_space(3))) float2 *)L + 1024u * wave_u;

    for (int e = tid; e < 960; e += 512) tw1[e] = A.tw[(((e >> 6) + 1) * (e & 63)) & 1023];
    if (tid < 64) L[B3_TW2 + tid] = A.tw[(16 * (tid & 3) * (tid >> 2)) & 1023];

    // ------------------------------------------------------------------ run start
    // items of the run: tile tile_begin + i, i = 0 .. n_items - 1; the first nwarm only refill the window (front waves), the next
    // one (tile first - 1) is muted: FIR and DFT for its last frame, the freqdem history of the run's first sample
    const unsigned tile_begin = w == 0 ? first : first - B3_HALO;
    const unsigned nwarm = w == 0 ? 0u : (unsigned)(FM ? B3_HALO - 1 : B3_HALO);      // (CF32 output has no freqdem history: the tile in front of the run only refills the window)
    const unsigned n_items = last - tile_begin;
    // TILE DMA: by the BACK waves (image s + 1 is requested in pieces between the arithmetic of phase P of step s and waited for at the end of
    // the step).  The front waves load their taps per step; loads return in order, so a DMA in their queue makes the first tap use wait for the
    // whole image (hipcc's waits cannot see an asm DMA), and eighteen VMEM instructions per wave in one burst cost the front waves -- the
    // role that paces the step -- ~850 cycles (the CU's address unit takes 16 cycles per 1 KiB instruction: CSDR_RUN1024_V3_TRACE).
    if (back) dma_tile(x4 + (size_t)tile_begin * 2048, goff, lds_wave);         // the first image: landed at the __syncthreads below
    float2 c = make_float2(0.f, 0.f);                   // DC state v before the next tile (same in every lane)
    {
        float2 acc = make_float2(0.f, 0.f);
        if (w > 0 && !back && !A.nowu) {
            // read-only warm-up (as k_run1024v2): the DC state before tile_begin from the six tiles in front of it, one batch of loads.  A run
            // that starts fewer than six tiles into the call (short calls: runs of one block) folds the tiles there are -- zeros stand for the
            // others -- and takes the rest from the stream's state below, which is then exact
            const int h0 = (int)tile_begin - B3_WU;
            float4 raw[8], rb[8], rc[8], rd[8], re[8], rf[8];
            float wt0, wt1;
            {
                const int wave = lt >> 6, lane = lt & 63;
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
            static_assert(B3_WU == 6, "one batch of six warm-up tiles");
            auto wload = [&](int t, float4 (&r)[8]) { tile_load(x4 + (size_t)(t > 0 ? t : 0) * 2048, t >= 0 ? 256 : 0, r, lt); };
            wload(h0, raw); wload(h0 + 1, rb); wload(h0 + 2, rc); wload(h0 + 3, rd); wload(h0 + 4, re); wload(h0 + 5, rf);
            fold(raw); fold(rb); fold(rc); fold(rd); fold(re); fold(rf);
        }
        const float2 sum = wg_sum(acc, red, tid);       // red[0..3]: the front waves (the back waves park zeros in red[4..7])
        if (w == 0) c = A.vend_in[0];
        else {
            c = sum;
            if ((int)tile_begin - B3_WU <= 0) c = cfma(A.vend_in[0], exp2f((float)(4096u * tile_begin) * A.l2beta), c);
        }
    }
    if (back) {                                         // freqdem history (after the reduction scratch is done with)
#pragma unroll
        for (int k3 = 0; k3 < 4; k3++) ST[4 * lt + (k3 ^ (2 * ((lt >> 3) & 1)))] = (FM && w == 0) ? A.rp_in[lt + 256 * k3] : make_float2(0.f, 0.f);   // (halves swapped where kk & 8: see z2r)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the first image: hipcc's own waits do not know about an asm DMA)
    __syncthreads();                                    // twiddle tables, stash, image 0

    if (!back) {
        // ================================================================== FRONT
        // WINDOW.  A ring of 16 frames x 4 branches in registers (128 VGPRs), the step loop unrolled four times so that every slot index is a
        // compile-time constant: in a step of phase PH = s & 3 the tile's frames f = 0, 1, 2 are written straight into slots 4 PH + f (they held
        // frames -16, -15, -14: dead), frame 3 waits in n3 until the FIR has read frame -13 out of slot 4 PH + 3 and is moved there at the end
        // of the step: four moves per step instead of k_run1024v2's 52 (a tenth of the front waves' time).  Frame -d of a step sits in slot
        // (4 PH - d) & 15.  The tile buffer index is s & 3 = PH as well.
        float2 ring[64];                                // [slot][branch]
#pragma unroll
        for (int i = 0; i < 64; i++) ring[i] = make_float2(0.f, 0.f);
        if (w == 0) {                                   // step 0 has phase 0: frame -13 + i in slot 3 + i
#pragma unroll
            for (int i = 0; i < 52; i++) ring[4 * (3 + (i >> 2)) + (i & 3)] = A.uhist_in[(i >> 2) * B3_M + 256 * (i & 3) + j];
        }
        // taps (14) and even / odd frame pre-mix phasors of a branch: four 16-byte loads + one 8-byte load per tile out of the L2-resident
        // table, as in k_run1024v2 (the window leaves no room for all 72 registers): branches 0 and 1 are requested at the top of the step
        // and fly during the DC scan, 2 and 3 between the FIR passes; the DMA of tile s + 2 is issued after the last of them has been
        // used (the waits the compiler places count vmcnt in order and would otherwise wait for the DMA as well)
        const unsigned joff = 16u * (unsigned)j;
        const __amdgpu_buffer_rsrc_t taps_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(A.taps_q), 0, 65536 + 8192, 0x00020000);
        auto load_taps = [&](v4f (&t)[4], v2f &wodd, const int qq) {
#pragma unroll
            for (int p = 0; p < 4; p++) {
                typedef unsigned v4u_ __attribute__((ext_vector_type(4)));
                const v4u_ v = __builtin_amdgcn_raw_buffer_load_b128(taps_rsrc, (int)joff, (qq * 4 + p) * 4096, 0);
                t[p] = __builtin_bit_cast(v4f, v);
            }
            typedef unsigned v2u_ __attribute__((ext_vector_type(2)));
            const v2u_ v = __builtin_amdgcn_raw_buffer_load_b64(taps_rsrc, (int)(joff >> 1), 65536 + qq * 2048, 0);
            wodd = __builtin_bit_cast(v2f, v);
        };
        const float kJ = -A.alpha * exp2f((float)j * A.l2beta);                  // -alpha beta^j: group state into column j
        const float b256 = A.b256[1];
        const bool odd0 = (A.parity0 & 1) != 0;
        const int q = lt, sw = (q >> 1) & 7;
        const unsigned raw_a = (unsigned)q * 128u + ((unsigned)sw << 4);         // slot i of my run: raw_a ^ (i << 4)

        const unsigned nsteps = n_items + 2;
        auto fstep = [&](const unsigned s, auto phc) {
            constexpr int PH = decltype(phc)::value;    // s & 3
            const bool have = s < n_items;
            const bool warm = s < nwarm;
            char *B = reinterpret_cast<char *>(L) + PH * (B3_BUF * 8);           // item s's buffer
            float2 *Bf = reinterpret_cast<float2 *>(B);
            const bool tr = B3_TRACE && A.trace && w == 1 && tid == 0 && s < 128;
            if (tr) trc[8 * s + 4] = __builtin_amdgcn_s_memtime();
            if (tr) trc[8 * s + 5] = __builtin_amdgcn_s_memtime();
            v4f tq0[4], tq1[4], tq2[4], tq3[4];
            v2f wo0, wo1, wo2, wo3;
            if (tr) trc[8 * s + 6] = __builtin_amdgcn_s_memtime();
            bar();                                      // P
            if (tr) trc[8 * s + 0] = __builtin_amdgcn_s_memtime();
            // no warm-up windows: the state in front of tile last - 4 is the next run's cold start (my own start error is beta^(>= 12 x 4096) of it)
            // the DC state in front of item s, parked in an LDS ring every step by every lane (the same value): a conditional global store, or
            // only a lane mask kept across the loop, costs the FM kernel an SGPR it does not have; the entry in front of tile last - 4 is
            // picked when the run is over
            cpre_s[s & 7u] = c;
            if (have) load_taps(tq0, wo0, 0);           // (the taps of branches 0 and 1 fly during pass 1 and the DC scan)
            asm volatile("" ::: "memory");
            if (s >= 1 + nwarm && s - 1 < n_items && !(B3_ABLATE & 16)) {
                // ---- DFT pass 1 of item s - 1 (its X is complete since the barrier): wave f = frame f, lane b1: radix 16 over a (n = 64 a + b1);
                // Z1 goes back into the frame block, where back wave f finds it in phase Q
                char *B1 = reinterpret_cast<char *>(L) + ((PH + 3) & 3) * (B3_BUF * 8);
                unsigned b1 = (unsigned)lt & 63u;
                asm volatile("" : "+v"(b1));            // (offsets derived here: as loop invariants they would pin VGPRs next to the window)
                const unsigned fb = 8192u * wave_u;
                const unsigned x_a = 8u * ((16u * (b1 >> 4)) | (b1 & 1u) | (2u * ((((b1 & 15u) >> 1) ^ (b1 >> 5)) & 7u)));
                const unsigned z1w = 128u * (b1 & 3u) + 8u * ((b1 >> 2) & 1u) + 16u * (((b1 >> 3) ^ ((b1 & 3u) >> 1) ^ (((b1 & 3u) >> 1) << 2)) & 7u);
                v2f vv[16];
#pragma unroll
                for (int a = 0; a < 16; a++) vv[a] = to_v(*reinterpret_cast<const float2 *>(B1 + fb + 512 * a + (x_a ^ (unsigned)((a & 3) << 5))));
                fft16_v(vv);
#pragma unroll
                for (int i = 1; i < 16; i++) vv[i] = cmul_v(vv[i], to_v(tw1[64 * (XIDX(i) - 1) + b1]));
#pragma unroll
                for (int i = 0; i < 16; i++)
                    *reinterpret_cast<float2 *>(B1 + fb + 512 * XIDX(i) + (z1w ^ (unsigned)(((2 * XIDX(i)) & 6) << 4))) = to_f2(vv[i]);
            }
            asm volatile("" ::: "memory");
            if (have) {
                load_taps(tq1, wo1, 1);
                // ---- DC blocker inside a 256-sample group: thread q owns the run of 16 consecutive samples q (as k_run256v2)
                const float na = opaque_v(-A.alpha), be = opaque_v(A.beta);
                v4f xr[8];
                float2 sc = make_float2(0.f, 0.f);
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    xr[i] = *reinterpret_cast<const v4f *>(B + (raw_a ^ (unsigned)(i << 4)));
                    sc = make_float2(fmaf(sc.x, be, xr[i].x), fmaf(sc.y, be, xr[i].y));
                    sc = make_float2(fmaf(sc.x, be, xr[i].z), fmaf(sc.y, be, xr[i].w));
                }
                {
                    float2 t;
                    t = dpp2<0x111>(sc); sc = cfma(t, A.b16[1], sc);
                    t = dpp2<0x112>(sc); sc = cfma(t, A.b16[2], sc);
                    t = dpp2<0x114>(sc); sc = cfma(t, A.b16[4], sc);
                    t = dpp2<0x118>(sc); sc = cfma(t, A.b16[8], sc);
                }
                if ((q & 15) == 15) Tt[q >> 4] = sc;
                sc = dpp2<0x111>(sc);
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    v4f y;
                    y.x = fmaf(sc.x, na, xr[i].x); y.y = fmaf(sc.y, na, xr[i].y);
                    sc = make_float2(fmaf(sc.x, be, xr[i].x), fmaf(sc.y, be, xr[i].y));
                    y.z = fmaf(sc.x, na, xr[i].z); y.w = fmaf(sc.y, na, xr[i].w);
                    sc = make_float2(fmaf(sc.x, be, xr[i].z), fmaf(sc.y, be, xr[i].w));
                    *reinterpret_cast<v4f *>(B + (raw_a ^ (unsigned)(i << 4))) = y;
                }
            }
            if (tr) trc[8 * s + 1] = __builtin_amdgcn_s_memtime();
            bar();                                      // Q: y' (group carry still missing), the group totals and Z1 of item s - 1 are visible
            if (tr) trc[8 * s + 2] = __builtin_amdgcn_s_memtime();
            if (!have) return;
            // ---- column layout: sample of frame f, branch j + 256 qq -> ring slot 4 PH + f (f < 3) / n3 (f = 3); group state chain V[g] (uniform)
            float2 n3[4];
#define NW(f, qq) (*((f) < 3 ? &ring[4 * ((4 * PH + (f)) & 15) + (qq)] : &n3[qq]))
            {
                v2f V = {c.x, c.y};
                const v2f kJv = {kJ, kJ}, bv = {b256, b256};
#pragma unroll
                for (int f = 0; f < 4; f++) {
                    const v4f t01 = *reinterpret_cast<const v4f *>(Tt + 4 * f), t23 = *reinterpret_cast<const v4f *>(Tt + 4 * f + 2);
                    const v2f tg[4] = {{t01.x, t01.y}, {t01.z, t01.w}, {t23.x, t23.y}, {t23.z, t23.w}};
#pragma unroll
                    for (int qq = 0; qq < 4; qq++) {
                        const int g = 4 * f + qq;
                        NW(f, qq) = to_f2(__builtin_elementwise_fma(V, kJv, to_v(Bf[256 * g + col_off])));
                        V = __builtin_elementwise_fma(V, bv, tg[qq]);
                    }
                }
                c = make_float2(__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(V.x))), __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(V.y))));
            }
            // pre-mix, then the polyphase FIR on the pre-mixed window: one branch at a time, four accumulators = its four frames
            auto branch = [&](const v4f (&t)[4], const v2f wodd, const int qq) {
                const v2f we = {t[3].z, t[3].w};
                const v2f Wa = odd0 ? wodd : we, Wb = odd0 ? we : wodd;
#pragma unroll
                for (int f = 0; f < 4; f += 2) {
                    v2f a0 = to_v(NW(f, qq)), a1 = to_v(NW(f + 1, qq));
                    cmul2_v(a0, Wa, a1, Wb);
                    NW(f, qq) = to_f2(a0); NW(f + 1, qq) = to_f2(a1);
                }
                if (warm || (B3_ABLATE & 32)) return;
                const float h[16] = {t[0].x, t[0].y, t[0].z, t[0].w, t[1].x, t[1].y, t[1].z, t[1].w, t[2].x, t[2].y, t[2].z, t[2].w, t[3].x, t[3].y, 0.f, 0.f};
                v2f acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
                for (int n = P - 1; n >= 0; n--) {
#pragma unroll
                    for (int f = 0; f < 4; f++) {
                        const int i = f - n;
                        const float2 s2 = (i >= 0) ? NW(i, qq) : ring[4 * ((4 * PH + i) & 15) + qq];
                        const v2f sv = {s2.x, s2.y}, hv = {h[n], h[n]};
                        acc[f] = __builtin_elementwise_fma(sv, hv, acc[f]);
                    }
                }
#pragma unroll
                for (int f = 0; f < 4; f++) Bf[256 * (4 * f + qq) + col_off] = to_f2(acc[f]);
            };
            // taps of branches 2 and 3: requested here / after branch 0 (they fly during the FIR of the branches before them)
            load_taps(tq2, wo2, 2);
            asm volatile("" ::: "memory");
            branch(tq0, wo0, 0);
            asm volatile("" ::: "memory");
            load_taps(tq3, wo3, 3);
            asm volatile("" ::: "memory");
            branch(tq1, wo1, 1);
            branch(tq2, wo2, 2);
            branch(tq3, wo3, 3);
            if (tr) trc[8 * s + 3] = __builtin_amdgcn_s_memtime();
            // frame 3 takes the slot of frame -13
#pragma unroll
            for (int qq = 0; qq < 4; qq++) ring[4 * ((4 * PH + 3) & 15) + qq] = n3[qq];
#undef NW
        };
        for (unsigned s0 = 0; s0 < nsteps; s0 += 4) {
            fstep(s0, std::integral_constant<int, 0>());
            if (s0 + 1 < nsteps) fstep(s0 + 1, std::integral_constant<int, 1>());
            if (s0 + 2 < nsteps) fstep(s0 + 2, std::integral_constant<int, 2>());
            if (s0 + 3 < nsteps) fstep(s0 + 3, std::integral_constant<int, 3>());
        }
        if (B3_TRACE && A.trace && w == 1 && tid == 0) for (int i = 0; i < 1024; i++) A.trace[i] = trc[i];
        if (lt == 0 && n_items >= (unsigned)B3_HALO) A.cpre[w + 1] = cpre_s[(n_items - (unsigned)B3_HALO) & 7u];     // (same thread wrote it)
        if (last == A.nb) {
            if (lt == 0) A.vend_out[0] = c;
            // the next call's window: frame -d behind the last item (phase (n_items - 1) & 3) sits in slot (4 (n_items & 3) - d) & 15
            auto put = [&](auto phc) {
                constexpr int PN = decltype(phc)::value;
#pragma unroll
                for (int i = 0; i < 52; i++) A.uhist_out[(i >> 2) * B3_M + 256 * (i & 3) + j] = ring[4 * ((4 * PN - 13 + (i >> 2)) & 15) + (i & 3)];
            };
            switch (n_items & 3u) {
            case 0: put(std::integral_constant<int, 0>()); break;
            case 1: put(std::integral_constant<int, 1>()); break;
            case 2: put(std::integral_constant<int, 2>()); break;
            default: put(std::integral_constant<int, 3>()); break;
            }
        }
        return;
    }

    // ====================================================================== BACK
    const FmK2 fk = {{A.pk.c[0], A.pk.c[1], A.pk.c[2], A.pk.c[3], A.pk.c[4], A.pk.c[5], A.pk.c[6], A.pk.c[7]}, A.tiny, A.fm_ref, A.pk.hp, A.pk.pi};
    const unsigned fb = 8192u * wave_u;                                         // passes 1-2: my wave's frame block
    // (pass 1, front waves: X[f][64 a + b1] sits at fb + 512 a + (x_a ^ ((a & 3) << 5)), the column layout of the raw image; Z1[k1][b = 4 c + d] is
    // written to fb + 512 k1 + (z1w ^ (((2 k1) & 6) << 4)) so that reader lane l2 = 4 k1 + d sees its 16 values as eight swizzled 16-byte pairs;
    // the swizzle of a reader row, ((l2 >> 1) & 7) ^ (4 for d >= 2), keeps the b128 reads conflict-free and puts the rows d and d + 2 that a writer's
    // 32-lane half touches into different bank quarters: tools/lds_conflicts_run1024v3.py)
    const int l2 = lt & 63, d2 = l2 & 3;                                        // pass 2: k1 = l2 >> 2, d = l2 & 3
    const unsigned z1r = fb + (unsigned)l2 * 128u + ((unsigned)(((l2 >> 1) & 7) ^ (((l2 >> 1) & 1) << 2)) << 4);    // pair i: z1r ^ (i << 4)
    // Z2[k1][k2][d]: thread kk = k1 + 16 k2 of pass 3 owns the 32 bytes at 32 kk; its two 16-byte halves are swapped where kk & 8, so that the
    // sixteen lanes a b128 read is served with cover all 64 banks (unswapped: 32-byte stride, two-way conflict)
    const unsigned z2w = fb + 32u * (unsigned)(l2 >> 2) + 16u * (unsigned)(((l2 >> 1) & 1) ^ ((l2 >> 5) & 1)) + 8u * (unsigned)(l2 & 1);     // + 512 k2
    const unsigned z2r = 32u * (unsigned)lt + 16u * (unsigned)((lt >> 3) & 1);   // pass 3: thread kk reads d = 0, 1 of frame f at 8192 f + z2r, d = 2, 3 at (8192 f + z2r) ^ 16
    // block flush: my wave's 8 KiB of a consumed tile buffer = 2 KiB of each frame block (exactly what its pass-3 reads covered), as
    // 512 16-byte slots: writer lane l, piece p -> slot 8 l + (p ^ ((l >> 1) & 7)); reader instruction m, lane l -> row r = 8 m + (l >> 3),
    // piece l & 7.  Slot sigma lies at 8192 (sigma >> 7) + 2048 wave + 16 (sigma & 127).
    const unsigned row_b = A.nf * (FM ? 4u : 8u);                               // bytes per output row
    const bool tmaj = !FM && A.tile_major != 0;
    const unsigned line_row = tmaj ? 128u : row_b, blk_step = tmaj ? 1024u * 128u : 128u;      // bytes between the lines of neighbouring channels / blocks
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(A.out, 0, (int)(1024u * row_b), 0x00020000);
    typedef unsigned v4u __attribute__((ext_vector_type(4)));

    // my four channels' results of the block's tiles: per k3 a 32-float register tuple [16-byte piece of the line][4] (128 VGPRs), written at a
    // uniform runtime index (v_movreld: the tile's position in the block lives in an SGPR) and read at constant indices by the flush
    typedef float v32f __attribute__((ext_vector_type(32)));
    v32f stg[4];
#pragma unroll
    for (int k3 = 0; k3 < 4; k3++)
#pragma unroll
        for (int i = 0; i < 32; i++) stg[k3][i] = 0.f;

    for (unsigned s = 0; s < n_items + 2; s++) {
        bar();                                          // P: Z2 of item s - 2 is complete
        const bool tr = B3_TRACE && A.trace && w == 1 && tid == 256 && s < 128;
        if (tr) trc[1024 + 4 * s + 0] = __builtin_amdgcn_s_memtime();
        // image s + 1 into the buffer item s - 3 left in phase P of the previous step: four pieces here, four behind pass 3's arithmetic
        const bool dma = s + 1 < n_items;
        const float4 *dsrc = x4 + (size_t)(tile_begin + s + 1) * 2048;
        const unsigned ddst = lds_wave + ((s + 1u) & 3u) * (B3_BUF * 8u);
        if (dma) { dma_piece(dsrc, goff, ddst, 0); dma_piece(dsrc, goff, ddst, 1); dma_piece(dsrc, goff, ddst, 2); dma_piece(dsrc, goff, ddst, 3); }
        if (s >= 2 + nwarm && !(B3_ABLATE & 64)) {
            // ---- DFT pass 3 + tail of item s - 2: thread kk = k1 + 16 k2, all four frames; Y[f][k3] = channel kk + 256 k3
            const unsigned b = (unsigned)__builtin_amdgcn_readfirstlane((int)(tile_begin + s - 2));
            const char *B = reinterpret_cast<const char *>(L) + ((s - 2u) & 3u) * (B3_BUF * 8u);
            v2f y[4][4];
#pragma unroll
            for (int f = 0; f < 4; f++) {
                const v4f v0 = *reinterpret_cast<const v4f *>(B + 8192 * f + z2r);
                const v4f v1 = *reinterpret_cast<const v4f *>(B + ((8192 * f + z2r) ^ 16u));
                y[f][0] = (v2f){v0.x, v0.y}; y[f][1] = (v2f){v0.z, v0.w}; y[f][2] = (v2f){v1.x, v1.y}; y[f][3] = (v2f){v1.z, v1.w};
                bfly4_v(y[f][0], y[f][1], y[f][2], y[f][3]);
            }
            const unsigned ts = b & (B3_TB - 1u);
            bool keep = true;
            if (FM && w > 0 && b + 1u >= first && b < first + 8u) {     // (always: a launch with warm-up windows just does not read them)
                // the uncorrected Y of the four channels around DC, frames -1 .. 31 of the run, for k_run1024_dcfix (channel kk + 256 k3:
                // 510, 511 = threads 254, 255 at k3 = 1; 512, 513 = threads 0, 1 at k3 = 2)
                const int chs = lt >= 254 ? lt - 254 : (lt < 2 ? lt + 2 : -1);
                if (chs >= 0) {
                    float2 *sd = kernarg_ptr_s<offsetof(Run1024v3Args, side)>() + ((size_t)w * 4 + (unsigned)chs) * RUN1024_DCFIX_F;
                    const int i0 = 4 * (int)(b + 1u - first) - 3;          // frame f of this tile -> slot i0 + f (slot 0 = frame -1)
#pragma unroll
                    for (int f = 0; f < 4; f++)
                        if (i0 + f >= 0) sd[i0 + f] = lt >= 254 ? to_f2(y[f][1]) : to_f2(y[f][2]);
                }
            }
            if (FM) {
                char *stp = reinterpret_cast<char *>(ST) + 32u * (unsigned)lt;    // my 32 bytes of the stash: [kk][k3], halves swapped like Z2's
                const unsigned sth = 16u * (unsigned)((lt >> 3) & 1);
                const v4f p01 = *reinterpret_cast<const v4f *>(stp + sth), p23 = *reinterpret_cast<const v4f *>(stp + (sth ^ 16u));
                const float2 prev[4] = {make_float2(p01.x, p01.y), make_float2(p01.z, p01.w), make_float2(p23.x, p23.y), make_float2(p23.z, p23.w)};
                *reinterpret_cast<v4f *>(stp + sth) = (v4f){y[3][0].x, y[3][0].y, y[3][1].x, y[3][1].y};
                *reinterpret_cast<v4f *>(stp + (sth ^ 16u)) = (v4f){y[3][2].x, y[3][2].y, y[3][3].x, y[3][3].y};
                keep = b >= first;                      // (the muted tile in front of the run: only its last frame was wanted)
                if (keep) {
                    v4f mv[4];
                    FmK2 fkt = fk;                      // tile-local copies of the uniform scalings (SGPRs)
                    asm volatile("" : "+s"(fkt.ref), "+s"(fkt.hp), "+s"(fkt.pi), "+s"(fkt.tiny));
#pragma unroll
                    for (int k3 = 0; k3 < 4; k3++) {
                        const float2 rp[4] = {prev[k3], to_f2(y[0][k3]), to_f2(y[1][k3]), to_f2(y[2][k3])};
                        const float2 rr[4] = {to_f2(y[0][k3]), to_f2(y[1][k3]), to_f2(y[2][k3]), to_f2(y[3][k3])};
                        float mq[4];
                        if (B3_ABLATE & 4) { mq[0] = rp[0].x + rr[0].y; mq[1] = rp[1].y + rr[1].x; mq[2] = rp[2].x + rr[2].y; mq[3] = rp[3].y + rr[3].x; }
                        else fm_quad(rp, rr, fkt, mq);
                        mv[k3] = (v4f){mq[0], mq[1], mq[2], mq[3]};
                    }
#pragma unroll
                    for (int k3 = 0; k3 < 4; k3++) {
                        stg[k3][4 * ts + 0] = mv[k3].x; stg[k3][4 * ts + 1] = mv[k3].y; stg[k3][4 * ts + 2] = mv[k3].z; stg[k3][4 * ts + 3] = mv[k3].w;
                    }
                }
            } else {
                // CF32: a tile is 32 bytes of a row = pieces 2 ts and 2 ts + 1 of its line
#pragma unroll
                for (int k3 = 0; k3 < 4; k3++) {
#pragma unroll
                    for (int f = 0; f < 4; f++) { stg[k3][8 * ts + 2 * f] = y[f][k3].x; stg[k3][8 * ts + 2 * f + 1] = y[f][k3].y; }
                }
            }
            if (keep) {
                if ((ts == B3_TB - 1u || b + 1 == last) && !(B3_ABLATE & 2)) {
                    // ---- the block is complete (or the call ends inside it: pieces 0 .. ts only): every row's 128 bytes leave in one piece
                    char *Bw = const_cast<char *>(B);
                    unsigned lf = (unsigned)l2;                                  // per-lane offsets of the flush, derived here: as loop invariants they would pin five more VGPRs
                    asm volatile("" : "+v"(lf));
                    const unsigned fl_w = 8192u * (unsigned)(lf >> 4) + 2048u * wave_u + 128u * (unsigned)(lf & 15) + ((unsigned)((lf >> 1) & 7) << 4);   // ^ (p << 4)
                    const unsigned fl_r0 = 2048u * wave_u + 128u * (unsigned)(lf >> 3) + 16u * (unsigned)((lf & 7) ^ (lf >> 4));                           // m even: + 8192 (m >> 1)
                    const unsigned fl_r1 = 2048u * wave_u + 1024u + 128u * (unsigned)(lf >> 3) + 16u * (unsigned)((lf & 7) ^ (4 + (lf >> 4)));            // m odd
                    const unsigned st_v = (64u * wave_u + (unsigned)(lf >> 3)) * line_row + 16u * (unsigned)(lf & 7);
                    const unsigned o0 = (b / B3_TB) * blk_step;                  // the block's line in a row (row-major: 128 bytes per block)
                    const bool pok = (FM ? (lf & 7u) : ((lf & 7u) >> 1)) <= ts;  // my 16-byte piece of the line exists (always, but in the call's last block)
#pragma unroll
                    for (int k3 = 0; k3 < 4; k3++) {
                        if (B3_ABLATE & 8) {
#pragma unroll
                            for (int m = 0; m < 8; m++)
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, (v4f){stg[k3][4 * m], stg[k3][4 * m + 1], stg[k3][4 * m + 2], stg[k3][4 * m + 3]}), ors, (int)st_v, (int)(o0 + (unsigned)(8 * m + 256 * k3) * line_row), 0);
                            continue;
                        }
#pragma unroll
                        for (int p = 0; p < 8; p++) *reinterpret_cast<v4f *>(Bw + (fl_w ^ (unsigned)(p << 4))) = (v4f){stg[k3][4 * p], stg[k3][4 * p + 1], stg[k3][4 * p + 2], stg[k3][4 * p + 3]};
#pragma unroll
                        for (int m = 0; m < 8; m++) {
                            const v4f v = *reinterpret_cast<const v4f *>(Bw + ((m & 1) ? fl_r1 : fl_r0) + 8192 * (m >> 1));
                            if (pok) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), ors, (int)st_v, (int)(o0 + (unsigned)(8 * m + 256 * k3) * line_row), 0);
                        }
                    }
                }
            }
        }
        if (dma) { dma_piece(dsrc, goff, ddst, 4); dma_piece(dsrc, goff, ddst, 5); dma_piece(dsrc, goff, ddst, 6); dma_piece(dsrc, goff, ddst, 7); }
        if (tr) trc[1024 + 4 * s + 1] = __builtin_amdgcn_s_memtime();
        bar();                                          // Q: Z1 of item s - 1 is complete
        if (tr) trc[1024 + 4 * s + 2] = __builtin_amdgcn_s_memtime();
        if (s >= 1 + nwarm && s - 1 < n_items && !(B3_ABLATE & 16)) {
            // ---- DFT pass 2 of item s - 1 (front wave f left Z1 in my frame block in phase P): lane (k1, d): radix 16 over c (b = 4 c + d)
            char *B = reinterpret_cast<char *>(L) + ((s - 1u) & 3u) * (B3_BUF * 8u);
            v2f vv[16];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const v4f v = *reinterpret_cast<const v4f *>(B + (z1r ^ (unsigned)(i << 4)));
                vv[2 * i] = (v2f){v.x, v.y}; vv[2 * i + 1] = (v2f){v.z, v.w};
            }
            fft16_v(vv);                                // vv[i] = k2 = XIDX(i)
            if (d2) {                                   // W64^(d k2); lanes d = 0 sit this out
#pragma unroll
                for (int i = 1; i < 16; i++) vv[i] = cmul_v(vv[i], to_v(L[B3_TW2 + 4 * XIDX(i) + d2]));
            }
#pragma unroll
            for (int i = 0; i < 16; i++) *reinterpret_cast<float2 *>(B + z2w + 512 * XIDX(i)) = to_f2(vv[i]);
        }
        if (tr) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); trc[1024 + 4 * s + 3] = __builtin_amdgcn_s_memtime(); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // my pieces of image s + 1 have landed (and the block's row stores, if this step had them)
    }
    if (B3_TRACE && A.trace && w == 1 && tid == 256) for (int i = 1024; i < 1536; i++) A.trace[i] = trc[i];
    if (FM && last == A.nb) {                           // a thread reads back what it wrote
#pragma unroll
        for (int k3 = 0; k3 < 4; k3++) A.rp_out[lt + 256 * k3] = ST[4 * lt + (k3 ^ (2 * ((lt >> 3) & 1)))];
    }
}

// What the DC state a run started without contributes to the channels 510..513 over the run's first 32 frames: Y += cpre[w] x R (the chain
// is linear up to Y).  CF32: in place on the rows (row-major or the tile-major plane); FM: freqdem of the corrected side copies.
template <bool FM>
__global__ __launch_bounds__(128) void k_run1024_dcfix(Run1024v3Args A, const float2 *__restrict__ rt)
{
    constexpr unsigned B3_TB = FM ? B3_TBF : B3_TBC;
    const unsigned w = blockIdx.x + 1u, ch = threadIdx.x >> 5, fr = threadIdx.x & 31u;
    unsigned first, last;
    run3_bounds(A.nb, A.nruns, w, B3_TB, first, last);
    const float2 c = A.cpre[w];
    const unsigned par = A.parity0 & 1u;                // a tile is 4 frames: every cold start begins on the call's parity
    const float2 *R = rt + (size_t)par * RUN1024_DCFIX_F * 4;
    auto corr = [&](unsigned i) { const float2 r = R[i * 4u + ch]; return make_float2(c.x * r.x - c.y * r.y, c.x * r.y + c.y * r.x); };
    const size_t t = (size_t)4 * first + fr;            // the frame in the call
    if (t >= A.nf) return;
    const unsigned k = 510u + ch;
    if (FM) {
        const float2 *sd = A.side + ((size_t)w * 4 + ch) * RUN1024_DCFIX_F;
        const float2 p0 = sd[fr], p1 = sd[fr + 1u], d0 = corr(fr), d1 = corr(fr + 1u);
        const FmK fk = {A.tiny, A.fm_ref, A.pk.hp, A.pk.pi};
        reinterpret_cast<float *>(A.out)[(size_t)k * A.nf + t] =
            fm_sample(make_float2(p0.x + d0.x, p0.y + d0.y), make_float2(p1.x + d1.x, p1.y + d1.y), fk);
    } else {
        float2 *o = reinterpret_cast<float2 *>(A.out) + (A.tile_major ? ((t >> 4) * 1024u + k) * 16u + (t & 15u) : (size_t)k * A.nf + t);
        const float2 d1 = corr(fr + 1u);
        float2 y = *o;
        y.x += d1.x; y.y += d1.y;
        *o = y;
    }
}

}  // namespace

int run1024_v3_launch(const Run1024v2Host &h, bool fm, uint32_t nruns, hipStream_t s, KernelTimer *timer)
{
    Run1024v3Args A{};
    A.x = h.x; A.out = h.out; A.taps_q = h.taps_q; A.tw = h.tw;
    A.uhist_in = h.uhist_in; A.uhist_out = h.uhist_out; A.vend_in = h.vend_in; A.vend_out = h.vend_out;
    A.rp_in = h.rp_in; A.rp_out = h.rp_out;
    A.nf = h.nf; A.nb = h.nf / B3_T4; A.nruns = nruns; A.parity0 = h.parity0; A.tile_major = (!fm && h.tile_major) ? 1u : 0u;
    const double beta = h.dc_block ? h.beta : 0.0;
    A.alpha = h.dc_block ? (float)(1.0 - beta) : 0.0f; A.beta = (float)beta; A.l2beta = h.dc_block ? (float)std::log2(beta) : -1000.0f;
    for (int i = 0; i < 16; i++) A.b16[i] = (float)std::pow(beta, 16.0 * i);
    for (int i = 0; i < 17; i++) A.b256[i] = (float)std::pow(beta, 256.0 * i);
    A.fm_ref = h.fm_ref; A.tiny = 1e-37f;
    A.pk = phase_consts(1.0f);                          // unscaled polynomial (fm_quad scales a = min / max by ref)
    A.pk.hp *= h.fm_ref; A.pk.pi *= h.fm_ref; A.pk.ref = h.fm_ref;
    static const char *trace_file = diag_env("CSDR_RUN1024_V3_TRACE");
    static unsigned long long *d_trace = nullptr;
    if (trace_file && !d_trace) CSDR_HIP(hipMalloc(&d_trace, 1536 * sizeof(unsigned long long)));
    if (trace_file) { CSDR_HIP(hipMemsetAsync(d_trace, 0, 1536 * sizeof(unsigned long long), s)); A.trace = d_trace; }
    int r;
    if (timer && (r = timer->begin(s))) return r;
    // no warm-up windows: whole runs of >= 16 tiles (a cold start is 4 tiles deep and the correction covers 4 more), the state arrays there
    if (!h.cpre || !h.side) { set_error("run1024_v3_launch: internal: no state arrays"); return -1; }
    A.nowu = (h.rt && h.dc_block && nruns >= 2 && A.nb / nruns >= 16u) ? 1u : 0u;
    A.cpre = h.cpre; A.side = h.side;
    if (fm) hipLaunchKernelGGL((k_run1024v3<true>), dim3(nruns), dim3(512), 0, s, A);
    else hipLaunchKernelGGL((k_run1024v3<false>), dim3(nruns), dim3(512), 0, s, A);
    if (A.nowu) {
        if (fm) hipLaunchKernelGGL((k_run1024_dcfix<true>), dim3(nruns - 1u), dim3(128), 0, s, A, h.rt);
        else hipLaunchKernelGGL((k_run1024_dcfix<false>), dim3(nruns - 1u), dim3(128), 0, s, A, h.rt);
    }
    if (timer && (r = timer->end(s))) return r;         // the bracket covers the correction kernel: it is part of every no-warm-up step
    CSDR_HIP(hipGetLastError());
    if (trace_file) {                                   // debug: the last launch's stamps, raw uint64: front [128][8], back [128][4]
        std::vector<unsigned long long> hbuf(1536);
        CSDR_HIP(hipStreamSynchronize(s));
        CSDR_HIP(hipMemcpy(hbuf.data(), d_trace, 1536 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        if (FILE *f = fopen(trace_file, "wb")) { fwrite(hbuf.data(), sizeof(unsigned long long), 1536, f); fclose(f); }
    }
    return 0;
}

uint32_t run1024_v3_runs(uint32_t nf, bool fm, uint32_t cus)
{
    const uint32_t B3_TB = fm ? B3_TBF : B3_TBC;
    // one workgroup per CU; runs are whole blocks of 8 (F32) / 4 (CF32) tiles (a row's 128-byte line), at least one (a run >= 1 walks 4
    // halo tiles in front of its first tile and folds the up to 6 tiles in front of those: short calls get as many runs as they have
    // blocks); the call's last block may be partial (nf = 0 mod 4: whole tiles)
    if (nf % B3_T4) return 0;
    const uint32_t nblk = (nf / B3_T4 + B3_TB - 1) / B3_TB;
    uint32_t nruns = cus;
    if (const char *e = diag_env("CSDR_RUN1024_V3_RUNS")) { const uint32_t v = (uint32_t)atoi(e); if (v >= 1 && v < nruns) nruns = v; }   // experiments
    if (nruns > nblk) nruns = nblk;
    return nruns;                                       // 0: a ragged call (not whole 4-frame tiles)
}

}  // namespace csdr
