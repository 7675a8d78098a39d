// Run kernel of an INTERLEAVED CHANNEL SHARD of the fused M = 1024 chain: what ONE rank of BASELINE configs[3] ("1024-ch PFB + FM demod,
// channels sharded across 8 x MI355X") runs -- rank g of G owns the channels g, g + G, g + 2G, ... (mux's per-channel independence,
// Trans.hs:124-129; one sink per channel, SoapySDR.hs:209-212, 223-225).
//
//   raw CF32 x --DC blocker--> y --NCO pre-mix (x W1024^(r g): the shift by g channels rides on the phasor table), 14-tap polyphase FIR-->
//   X'_t[r], r < 1024 --fold: Z_t[n1] = sum_{n2 < G} X'_t[n1 + (1024/G) n2]-- (1024/G)-point forward DFT--> Y_t[g + G m], m < 1024/G
//   --per-channel freqdem--> out[1024/G][nf]       (8 B read + 4/G (F32) or 8/G (CF32) B written per input sample; Liquid.chs:575-589, 828-862, 324-328)
//
// Every rank still reads, DC-blocks and FIR-filters the WHOLE stream (the front end does not shard by channel); what a shard
// saves is the transform and the tail: the G branches n1 + (1024/G) n2 that alias onto the same bin of the short DFT are SUMMED behind the FIR,
// and one 128-point DFT per frame (G = 8) replaces the 1024-point one.  k_run1024v2<FM, G> (round 3) pruned the three passes of the long
// transform instead and kept their LDS images and four barriers per tile; here the step is the front half of k_run1024v3's, on all eight waves:
//
//   512 threads, ONE workgroup per CU.  Thread (rho = tid >> 8, j = tid & 255) owns the two polyphase branches j + 256 (2 rho + q'), q' = 0, 1:
//   their window is a ring of 16 frames x 2 branches in registers (64 VGPRs, slot indices compile-time constants: the step loop is unrolled four
//   times), and -- unlike k_run1024v2 / v3, whose threads own four branches -- their 28 taps and 4 phasors STAY in registers: no tap re-reads
//   (v3: 72 KiB of L2 reads per 32 KiB tile).  A step = one tile of 4 frames, two workgroup barriers:
//
//     bar P  ------------------------------------------------------------------------------------------------------------------------
//            waves 0-3 (rho = 0): DC scan of tile s (raw image -> y',     waves 4-7 (rho = 1): DMA of tile s + 2 (pieces between the DFT's stages);
//            16 group totals); freqdem + row stores of tile s - 2         fold + pruned DFT of tile s - 1 (wave f = frame f)
//     bar Q  ------------------------------------------------------------------------------------------------------------------------
//            all eight waves: column layout + group carries, pre-mix, FIR of the own two branches, their sum (G >= 4: both alias onto
//            the same n1) -> partial folds P[part][f][n1] in LDS
//
//   The pruned DFT runs ACROSS THE LANES of a wave: lane l holds Z[l + 64 a], a < NP / 64; a radix-(NP/64) step on the registers, then six
//   radix-2 decimation-in-frequency stages whose partner values come through DPP row operations (distances 1 .. 8) and
//   v_permlane16_swap / v_permlane32_swap (16, 32): no LDS image, no LDS crossbar, no barrier; lane l ends up with
//   the bins a' + (NP/64) bitrev6(l), stored as one 16-byte piece per pair.  The tail thread m = channel g + G m reads its four frames,
//   freqdem against the stash (fm_quad); G = 8: its 16-byte pieces wait in 128 bytes of LDS per row and leave as a whole line per block of
//   8 tiles (stored piece by piece the L2s evicted the half-written lines: 4.2x the output bytes in WRITE_SIZE).
//
// State arrays and tables are the plan's (kernels_pfb1024.hip): window uhist [13][1024] pre-mixed, DC state, freqdem history rp[] indexed by
// the PRIMED channel k' = G m, tap table taps_q with the shard's phasors -- so ragged and short calls of the same handle can take the
// whole-band kernel + row gather and carry the same state.
// Run starts: cold, from DC state 0 (Shard1024Args::nowu; the six read-only warm-up tiles of the first version were 9 % of the reads): three
// window-refill tiles and (FM) the muted tile in front of the run for the freqdem history; what the missing state contributes to the ONE owned
// channel among the four around DC (none for the shards 2 .. 5 of 8) is put back by k_shard1024_dcfix behind the launch, as k_run1024_dcfix does.
#include "fused_v2_common.h"
#include <type_traits>

#ifndef S1_TRACE
#define S1_TRACE 0       // 1: s_memtime stamps of run 1's wave 0 (role 0) / wave 4 (role 1) per phase, written to Shard1024Args::trace (CSDR_SHARD1024_TRACE=file)
#endif

namespace csdr {
namespace {

constexpr int S1_BUF = 4096;                       // float2 per tile buffer (32 KiB)
constexpr int S1_WU = 6, S1_HALO = 4;

template <int G> struct S1 {
    static constexpr int NP = 1024 / G;            // points of the pruned DFT = owned channels
    static constexpr int PPL = NP / 64;            // points per lane
    static constexpr int NPART = G == 8 ? 4 : 2;   // partial folds per n1: (rho, j >> 7) at G = 8, rho otherwise
    static constexpr int NBUF = 3, AHEAD = NBUF - 1;   // tile buffers; the DMA runs two tiles ahead
    static constexpr int PB = NBUF * S1_BUF;       // partial folds [NPART][4][NP]
    static constexpr int YB = PB + NPART * 4 * NP; // Y [2][4][NP]: written by the DFT of step s, read by the tail of step s + 1
    static constexpr int ST = YB + 2 * 4 * NP;     // last Y frame of every owned channel [NP] (FM)
    static constexpr int TT = ST + NP;             // 16 group totals
    static constexpr int RED = TT + 16;            // 8 reduction slots of the prologue
    static constexpr int SGB = G == 8 ? 128 : 64;  // output bytes of a row staged in LDS before they leave in one piece: a whole 128-byte line at G = 8, half a
                                                   // line at G = 4 (a CU has no 32 KiB left there; 64-byte pieces are what k_run256v2 stores, at 1.0x WRITE_SIZE)
    static constexpr int SG = RED + 8;             // [NP][SGB]: the tail thread's own row, 16-byte slot p at p ^ (m & (SGB / 16 - 1))
    static constexpr int F2 = SG + NP * SGB / 8;
    static_assert(G == 4 || G == 8, "built for strides 4 and 8 (stride 2 keeps k_run1024v2<FM, 2>)");
    static_assert(F2 * 8 <= 160 * 1024, "one workgroup per CU");
};

struct Shard1024Args {
    const float2 *x;            // raw input of this call
    void *out;                  // [1024 / G][nf] F32 (FM) / CF32
    const float4 *taps_q;       // the plan's table: [4 q][4 pieces][256 j] taps + even-frame phasor; behind the 64 KiB: [4 q][256 j] odd-frame phasor
    const float2 *tw;           // e^{-j 2 pi i / 1024}
    const float2 *uhist_in; float2 *uhist_out;    // [13][1024] pre-mixed, DC-blocked window before / after the call
    const float2 *vend_in; float2 *vend_out;      // DC blocker state v1
    const float2 *rp_in; float2 *rp_out;          // [1024] freqdem r', indexed by the primed channel k' = G m
    uint32_t nf, nb, nruns, parity0;
    uint32_t tile_major;        // CF32: the 128-byte lines of a 16-frame block back to back, [block][1024 / G][128 B] (the plane k_agc_spec_tm reads), instead of rows
    float alpha, beta, l2beta, fm_ref, tiny;
    float b16[16];              // beta^(16 r)
    float b256[17];             // beta^(256 g)
    PhaseK pk;
    unsigned long long *trace;  // debug (S1_TRACE build): [role][step < 96][4] stamps
    uint32_t nowu;              // 1: a run starts cold from DC state 0 (no read-only warm-up tiles), leaves the state in front of tile last - 4 in
    int mfix;                   //    cpre[w + 1] and (FM) the uncorrected Y of row mfix (the owned channel among 510 .. 513; -1: none) for its frames
    float2 *cpre, *side;        //    -1 .. 31 in side [w][RUN1024_DCFIX_F]: k_shard1024_dcfix
};

// value of lane l ^ D (upper: bit D of my lane index is set), without a trip through the LDS crossbar: DPP inside a row of 16 lanes,
// v_permlane16_swap / v_permlane32_swap (gfx950) across rows and halves -- a ds_bpermute / ds_swizzle per value put ~450 cycles of latency on
// each of the first two butterfly stages (phase trace: 1274 cycles for the two of them, 110 per DPP stage)
template <int D> __device__ __forceinline__ float lane_xor(float x, bool upper)
{
    const int v = __float_as_int(x);
    if (D == 1) return __int_as_float(__builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));        // quad_perm [1,0,3,2]
    if (D == 2) return __int_as_float(__builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));        // quad_perm [2,3,0,1]
    if (D == 4) {
        const int a = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xf, 0x5, false);                        // row_shl:4 into the banks 0, 2 (lane i <- i + 4)
        return __int_as_float(__builtin_amdgcn_update_dpp(a, v, 0x114, 0xf, 0xA, false));               // row_shr:4 into the banks 1, 3 (lane i <- i - 4)
    }
    if (D == 8) return __int_as_float(__builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false));        // row_ror:8
    if (D == 16) {
        // swap the odd rows of the first copy with the even rows of the second: first = [r0, r0, r2, r2], second = [r1, r1, r3, r3]
        const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
        return __uint_as_float(upper ? r[0] : r[1]);
    }
    // swap the upper half of the first copy with the lower half of the second: first = [lo, lo], second = [hi, hi]
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return __uint_as_float(upper ? r[0] : r[1]);
}

// run w: whole blocks of TB tiles (a row's 128-byte line), evenly; the call's last block may be a partial one
__host__ __device__ __forceinline__ void shard_bounds(uint32_t nb, uint32_t nruns, unsigned w, unsigned TB, unsigned &first, unsigned &last)
{
    const unsigned nblk = (nb + TB - 1) / TB;
    first = TB * (unsigned)((unsigned long long)w * nblk / nruns);
    last = TB * (unsigned)((unsigned long long)(w + 1) * nblk / nruns);
    if (last > nb) last = nb;
}

template <bool FM, int G>
__global__ __launch_bounds__(512) void k_shard1024(Shard1024Args A)
{
    using K = S1<G>;
    constexpr int NP = K::NP, PPL = K::PPL, NBUF = K::NBUF, AHEAD = K::AHEAD;
    constexpr unsigned TB = FM ? 8u : 4u;
    __shared__ __attribute__((aligned(16))) float2 L[K::F2];
    float2 *Pb = L + K::PB, *Yb = L + K::YB, *ST = L + K::ST, *Tt = L + K::TT, *red = L + K::RED;
    __shared__ unsigned long long trc[S1_TRACE ? 768 : 1];
    const int tid = threadIdx.x;
    const int rho = __builtin_amdgcn_readfirstlane(tid >> 8);          // wave-uniform role
    const int lt = tid & 255, j = lt;
    const unsigned w = blockIdx.x;
    unsigned first, last;
    shard_bounds(A.nb, A.nruns, w, TB, first, last);
    const float4 *x4 = reinterpret_cast<const float4 *>(A.x);
    const int col_off = 16 * (j >> 4) + 2 * (((j & 15) >> 1) ^ (j >> 5)) + (j & 1);
    const int colb = col_off + 512 * rho;               // my first branch's group: 2 rho
    const unsigned goff = dma_offset(lt);
    const unsigned wave_u = (unsigned)__builtin_amdgcn_readfirstlane(lt >> 6);      // wave index inside the role
    const unsigned lds_wave = (unsigned)(size_t)(__attribute__((address_space(3))) float2 *)L + 1024u * wave_u;

    // items of the run: tile tile_begin + i; the first nwarm only refill the window; (FM) the next one, tile first - 1, is muted: FIR and DFT for
    // its last frame, the freqdem history of the run's first sample
    const unsigned tile_begin = w == 0 ? first : first - S1_HALO;
    const unsigned nwarm = w == 0 ? 0u : (unsigned)(FM ? S1_HALO - 1 : S1_HALO);
    const unsigned n_items = last - tile_begin;
    if (rho) {                                          // the first images: items 0 and 1
        dma_tile(x4 + (size_t)tile_begin * 2048, goff, lds_wave);
        if (AHEAD >= 2 && n_items > 1) dma_tile(x4 + (size_t)(tile_begin + 1) * 2048, goff, lds_wave + S1_BUF * 8u);
    }

    // ------------------------------------------------------------------ run start: DC state
    float2 c = make_float2(0.f, 0.f);                   // DC state v before the next tile (same in every lane)
    {
        float2 acc = make_float2(0.f, 0.f);
        if (w > 0 && !rho && !A.nowu) {
            // read-only warm-up (as k_run1024v3): the DC state before tile_begin from the six tiles in front of it, one batch of loads.  A run
            // that starts fewer than six tiles into the call folds the tiles there are -- zeros stand for the others -- and takes the rest from
            // the stream's state below, which is then exact
            const int h0 = (int)tile_begin - S1_WU;
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
            static_assert(S1_WU == 6, "one batch of six warm-up tiles");
            auto wload = [&](int t, float4 (&r)[8]) { tile_load(x4 + (size_t)(t > 0 ? t : 0) * 2048, t >= 0 ? 256 : 0, r, lt); };
            wload(h0, raw); wload(h0 + 1, rb); wload(h0 + 2, rc); wload(h0 + 3, rd); wload(h0 + 4, re); wload(h0 + 5, rf);
            fold(raw); fold(rb); fold(rc); fold(rd); fold(re); fold(rf);
        }
        const float2 sum = wg_sum(acc, red, tid);       // red[0..3]: the waves of role 0 (role 1 parks zeros in red[4..7])
        if (w == 0) c = A.vend_in[0];
        else {
            c = sum;
            if ((int)tile_begin - S1_WU <= 0) c = cfma(A.vend_in[0], exp2f((float)(4096u * tile_begin) * A.l2beta), c);
        }
    }
    // ------------------------------------------------------------------ my two branches' taps and phasors (registers for the whole run)
    float h[2][16];
    v2f Wa[2], Wb[2];                                   // phasors of the frames 0, 2 / 1, 3 of a tile (a tile starts on the call's parity)
    {
        const bool odd0 = (A.parity0 & 1) != 0;
#pragma unroll
        for (int qp = 0; qp < 2; qp++) {
            const int qq = 2 * rho + qp;
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const float4 t = A.taps_q[(qq * 4 + p) * 256 + j];
                h[qp][4 * p] = t.x; h[qp][4 * p + 1] = t.y; h[qp][4 * p + 2] = t.z; h[qp][4 * p + 3] = t.w;
            }
            const float2 wo = reinterpret_cast<const float2 *>(A.taps_q + 4096)[qq * 256 + j];
            const v2f we = {h[qp][14], h[qp][15]}, wov = {wo.x, wo.y};
            Wa[qp] = odd0 ? wov : we; Wb[qp] = odd0 ? we : wov;
        }
    }
    // ------------------------------------------------------------------ the DFT's per-lane twiddles (role 1)
    const int l6 = lt & 63;
    v2f tw0[PPL], tws[5];
#pragma unroll
    for (int k = 0; k < PPL; k++) { const float2 t = A.tw[(G * l6 * k) & 1023]; tw0[k] = (v2f){t.x, t.y}; }        // W_NP^(l k)
#pragma unroll
    for (int i = 0; i < 5; i++) {                       // stage distance d = 32 >> i: upper lanes W_2d^(l & (d - 1)), lower lanes 1
        const int d = 32 >> i;
        const float2 t = A.tw[((512 / d) * (l6 & (d - 1))) & 1023];
        tws[i] = (l6 & d) ? (v2f){t.x, t.y} : (v2f){1.f, 0.f};
    }

    if (!rho) {                                         // freqdem history of the owned channels (the tail runs on the waves of role 0, behind the scan)
        for (int m = lt; m < NP; m += 256) ST[m] = (FM && w == 0) ? A.rp_in[G * m] : make_float2(0.f, 0.f);
    } else {
        // the first image has landed (hipcc's own waits do not know about an asm DMA)
        if (AHEAD >= 2 && n_items > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    // WINDOW.  A ring of 16 frames x 2 branches in registers, the step loop unrolled four times so that every slot index is a compile-time
    // constant (k_run1024v3's): in a step of phase PH = s & 3 the tile's frames f = 0, 1, 2 go straight into slots 4 PH + f (they held frames
    // -16, -15, -14: dead), frame 3 waits in n3 until the FIR has read frame -13 out of slot 4 PH + 3.  Frame -d of a step sits in slot
    // (4 PH - d) & 15.
    float2 ring[32];                                    // [slot][branch]
#pragma unroll
    for (int i = 0; i < 32; i++) ring[i] = make_float2(0.f, 0.f);
    if (w == 0) {                                       // step 0 has phase 0: frame -13 + i in slot 3 + i
#pragma unroll
        for (int i = 0; i < 13; i++)
#pragma unroll
            for (int qp = 0; qp < 2; qp++) ring[2 * (3 + i) + qp] = A.uhist_in[i * 1024 + 256 * (2 * rho + qp) + j];
    }

    const float kJ = -A.alpha * exp2f((float)j * A.l2beta);                  // -alpha beta^j: group state into column j
    const float b256 = A.b256[1];
    const int q = lt, sw = (q >> 1) & 7;
    const unsigned raw_a = (unsigned)q * 128u + ((unsigned)sw << 4);         // role 0, DC scan: slot i of my run of 16 samples: raw_a ^ (i << 4)
    const FmK2 fk = {{A.pk.c[0], A.pk.c[1], A.pk.c[2], A.pk.c[3], A.pk.c[4], A.pk.c[5], A.pk.c[6], A.pk.c[7]}, A.tiny, A.fm_ref, A.pk.hp, A.pk.pi};
    const unsigned row_b = A.nf * (FM ? 4u : 8u);                            // bytes per output row
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(A.out, 0, (int)((unsigned)NP * row_b), 0x00020000);
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    // where my partial fold goes: [part][f][n1]
    const int part = G == 8 ? 2 * rho + (j >> 7) : rho;
    const int n1 = G == 8 ? (j & 127) : j;
    const int br6 = (int)(__builtin_bitreverse32((unsigned)l6) >> 26);

    const unsigned nsteps = n_items + 2;
    auto step = [&](const unsigned s, auto phc) {
        constexpr int PH = decltype(phc)::value;        // s & 3
        const bool have = s < n_items;
        const bool warm = s < nwarm;
        const unsigned bi = (unsigned)__builtin_amdgcn_readfirstlane((int)(s % (unsigned)NBUF));
        char *B = reinterpret_cast<char *>(L) + bi * (S1_BUF * 8u);           // item s's buffer
        float2 *Bf = reinterpret_cast<float2 *>(B);
        const bool tr = S1_TRACE && A.trace && w == 1 && (tid & 255) == 0 && s < 96;
        unsigned long long *tq = trc + 384 * rho + 4 * s;
        bar();                                          // P: the partial folds of item s - 1 and Y of item s - 2 are complete; image s has landed
        if (tr) tq[0] = __builtin_amdgcn_s_memtime();
        if (!rho) {
            if (have) {
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
            if (S1_TRACE == 2 && tr) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tq[1] = __builtin_amdgcn_s_memtime(); }
            // ---- tail of item s - 2: thread m = owned channel m (global channel g + G m): its four frames, freqdem against the stash.  A tile is
            // 16 bytes (F32) / 32 bytes (CF32) of row m.  Stored as such (first version), the L2s do NOT hold the half-written lines until they are
            // complete: WRITE_SIZE showed 140 MB per launch for 33.5 MB of output (G = 8), 154 for 67 (G = 4).  So the pieces wait in the
            // thread's own SGB bytes of LDS (slot p at p ^ (m & (SGB / 16 - 1))) and leave together, SGB / 16 stores in a row, when they are
            // complete (or the call ends inside them); no other thread touches them, so no barrier is involved.
            if (s >= 2 + nwarm && s - 2 < n_items) {
                const unsigned b = (unsigned)__builtin_amdgcn_readfirstlane((int)(tile_begin + s - 2));
                const float2 *Y = Yb + (s & 1u) * (4 * NP);
                constexpr unsigned SGB = K::SGB, NSL = SGB / 16u, TF = SGB / (FM ? 16u : 32u);      // staged bytes per row, their 16-byte slots, tiles per flush
                const unsigned ts = b & (TF - 1u);
                // (the tile-major plane keeps direct stores: a block's lines of all rows are one contiguous 16 KiB there, written within four steps
                // by one workgroup -- the L2s do merge those, and staging them measured 3 % slower)
                const bool stage = FM || !A.tile_major;
                const bool flush = stage && b >= first && (ts == TF - 1u || b + 1u == last);
                for (int m = lt; m < NP; m += 256) {
                    float2 y[4];
#pragma unroll
                    for (int f = 0; f < 4; f++) y[f] = Y[f * NP + m];
                    char *sg = reinterpret_cast<char *>(L + K::SG) + SGB * m;
                    const unsigned sw = (unsigned)m & (NSL - 1u);
                    if (FM && w > 0 && m == A.mfix && b + 1u >= first && b < first + 8u) {
                        // the uncorrected Y of the owned channel next to DC, frames -1 .. 31 of the run, for k_shard1024_dcfix
                        float2 *sd = A.side + (size_t)w * RUN1024_DCFIX_F;
                        const int i0 = 4 * (int)(b + 1u - first) - 3;          // frame f of this tile -> slot i0 + f (slot 0 = frame -1)
#pragma unroll
                        for (int f = 0; f < 4; f++)
                            if (i0 + f >= 0) sd[i0 + f] = y[f];
                    }
                    if (FM) {
                        const float2 prev = ST[m];
                        ST[m] = y[3];
                        if (b >= first) {               // (the muted tile in front of the run: only its last frame was wanted)
                            const float2 rp[4] = {prev, y[0], y[1], y[2]};
                            float mq[4];
                            fm_quad(rp, y, fk, mq);
                            *reinterpret_cast<v4f *>(sg + 16u * (ts ^ sw)) = (v4f){mq[0], mq[1], mq[2], mq[3]};
                        }
                    } else {
                        const v4f v0 = {y[0].x, y[0].y, y[1].x, y[1].y}, v1 = {y[2].x, y[2].y, y[3].x, y[3].y};
                        if (stage) {
                            *reinterpret_cast<v4f *>(sg + 16u * ((2u * ts) ^ sw)) = v0;
                            *reinterpret_cast<v4f *>(sg + 16u * ((2u * ts + 1u) ^ sw)) = v1;
                        } else {                        // tile-major: pieces 2 (b & 3), 2 (b & 3) + 1 of channel m's line in block b >> 2
                            const unsigned so = (b >> 2) * ((unsigned)NP * 128u) + 32u * (b & 3u);
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v0), ors, (int)((unsigned)m * 128u), (int)so, 0);
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v1), ors, (int)((unsigned)m * 128u), (int)(so + 16u), 0);
                        }
                    }
                    if (flush) {
                        const unsigned pmax = FM ? ts : 2u * ts + 1u;       // (a call that ends inside a block: the front part of it)
                        const unsigned so = (b / TF) * SGB;                  // my row's staged piece of this block
#pragma unroll
                        for (unsigned pc = 0; pc < NSL; pc++) {
                            const v4f v = *reinterpret_cast<const v4f *>(sg + 16u * (pc ^ sw));
                            if (pc <= pmax) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), ors, (int)((unsigned)m * row_b), (int)(so + 16u * pc), 0);
                        }
                    }
                }
            }
        } else {
            // ---- image s + AHEAD into the buffer item s - 1 left at barrier P: two pieces here, the others between the stages of the DFT (the CU's
            // address unit takes 16 cycles per 1 KiB instruction: eight in a row stall the wave for ~800)
            const bool dma = s + AHEAD < n_items;
            const float4 *dsrc = x4 + (size_t)(tile_begin + s + AHEAD) * 2048;
            const unsigned ddst = lds_wave + (unsigned)((s + AHEAD) % (unsigned)NBUF) * (S1_BUF * 8u);
            if (dma) { dma_piece(dsrc, goff, ddst, 0); dma_piece(dsrc, goff, ddst, 1); }
            // ---- fold + pruned DFT of item s - 1: wave f = frame f, lane l holds Z[l + 64 a]
            if (s >= 1 + nwarm && s - 1 < n_items) {
                const int f = (int)wave_u;
                v2f z[PPL];
#pragma unroll
                for (int a = 0; a < PPL; a++) {
                    v2f acc = to_v(Pb[(0 * 4 + f) * NP + l6 + 64 * a]);
#pragma unroll
                    for (int p = 1; p < K::NPART; p++) acc = acc + to_v(Pb[(p * 4 + f) * NP + l6 + 64 * a]);
                    z[a] = acc;
                }
                // radix-PPL step on the registers (decimation in frequency): u_k = sum_a z_a W_PPL^(a k), then x W_NP^(l k); chain k holds
                // the 64-point problem whose bin k64 is Y[k + PPL k64]
                if (PPL == 2) { const v2f u0 = z[0] + z[1], u1 = z[0] - z[1]; z[0] = u0; z[1] = cmul_v(u1, tw0[1]); }
                else {
                    bfly4_v(z[0], z[1], z[2], z[3]);
#pragma unroll
                    for (int k = 1; k < PPL; k++) z[k] = cmul_v(z[k], tw0[k]);
                }
                // six radix-2 stages across the lanes: partner l ^ d; lower lane x + p, upper lane (p - x) W_2d^(l & (d - 1))
                auto stage = [&](auto dc, const int i) {
                    constexpr int D = decltype(dc)::value;
                    const bool upper = (l6 & D) != 0;
                    const unsigned sgn = upper ? 0x80000000u : 0u;
#pragma unroll
                    for (int k = 0; k < PPL; k++) {
                        const v2f p = {lane_xor<D>(z[k].x, upper), lane_xor<D>(z[k].y, upper)};
                        const v2f xs = {__uint_as_float(__float_as_uint(z[k].x) ^ sgn), __uint_as_float(__float_as_uint(z[k].y) ^ sgn)};
                        v2f y = p + xs;
                        if (D > 1) y = cmul_v(y, tws[i]);
                        z[k] = y;
                    }
                };
                if (S1_TRACE == 2 && tr) tq[1] = __builtin_amdgcn_s_memtime();
                stage(std::integral_constant<int, 32>(), 0);
                if (dma) { dma_piece(dsrc, goff, ddst, 2); dma_piece(dsrc, goff, ddst, 3); }
                stage(std::integral_constant<int, 16>(), 1);
                if (dma) { dma_piece(dsrc, goff, ddst, 4); dma_piece(dsrc, goff, ddst, 5); }
                if (S1_TRACE == 2 && tr) tq[2] = __builtin_amdgcn_s_memtime();
                stage(std::integral_constant<int, 8>(), 2);
                if (dma) { dma_piece(dsrc, goff, ddst, 6); dma_piece(dsrc, goff, ddst, 7); }
                stage(std::integral_constant<int, 4>(), 3);
                stage(std::integral_constant<int, 2>(), 4);
                stage(std::integral_constant<int, 1>(), 5);
                // lane l holds the bins k + PPL bitrev6(l), k < PPL: contiguous
                float2 *Yw = Yb + ((s + 1u) & 1u) * (4 * NP) + f * NP + PPL * br6;
                *reinterpret_cast<v4f *>(Yw) = (v4f){z[0].x, z[0].y, z[1].x, z[1].y};
                if (PPL == 4) *reinterpret_cast<v4f *>(Yw + 2) = (v4f){z[2].x, z[2].y, z[3].x, z[3].y};
            } else if (dma) {
#pragma unroll
                for (int it = 2; it < 8; it++) dma_piece(dsrc, goff, ddst, it);
            }
        }
        if (tr) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tq[S1_TRACE == 2 ? (rho ? 3 : 2) : 1] = __builtin_amdgcn_s_memtime(); }
        bar();                                          // Q: y' (group carry still missing) and the group totals of item s are visible
        if (S1_TRACE != 2 && tr) tq[2] = __builtin_amdgcn_s_memtime();
        if (have) {
            // no warm-up windows: the state in front of tile last - 4 is the next run's cold start (my own start error is beta^(>= 12 x 4096) of it)
            if (tid == 0 && tile_begin + s + (unsigned)S1_HALO == last) A.cpre[w + 1] = c;
            // ---- column layout: sample of frame f, branch j + 256 (2 rho + qp) -> ring slot 4 PH + f (f < 3) / n3 (f = 3); group state chain V[g] (uniform)
            float2 n3[2];
#define NW(f, qp) (*((f) < 3 ? &ring[2 * ((4 * PH + (f)) & 15) + (qp)] : &n3[qp]))
            {
                v2f V = {c.x, c.y};
                const v2f kJv = {kJ, kJ}, bv = {b256, b256};
#pragma unroll
                for (int f = 0; f < 4; f++) {
                    const v4f t01 = *reinterpret_cast<const v4f *>(Tt + 4 * f), t23 = *reinterpret_cast<const v4f *>(Tt + 4 * f + 2);
                    const v2f V0 = V;
                    const v2f V1 = __builtin_elementwise_fma(V0, bv, (v2f){t01.x, t01.y});
                    const v2f V2 = __builtin_elementwise_fma(V1, bv, (v2f){t01.z, t01.w});
                    const v2f V3 = __builtin_elementwise_fma(V2, bv, (v2f){t23.x, t23.y});
                    V = __builtin_elementwise_fma(V3, bv, (v2f){t23.z, t23.w});
                    const v2f Va = rho ? V2 : V0, Vb2 = rho ? V3 : V1;       // my groups 4 f + 2 rho, 4 f + 2 rho + 1
                    NW(f, 0) = to_f2(__builtin_elementwise_fma(Va, kJv, to_v(Bf[256 * (4 * f) + colb])));
                    NW(f, 1) = to_f2(__builtin_elementwise_fma(Vb2, kJv, to_v(Bf[256 * (4 * f + 1) + colb])));
                }
                c = make_float2(__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(V.x))), __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(V.y))));
            }
            // pre-mix, then the polyphase FIR on the pre-mixed window: one branch at a time, four accumulators = its four frames
            v2f X[2][4];
#pragma unroll
            for (int qp = 0; qp < 2; qp++) {
#pragma unroll
                for (int f = 0; f < 4; f += 2) {
                    v2f a0 = to_v(NW(f, qp)), a1 = to_v(NW(f + 1, qp));
                    cmul2_v(a0, Wa[qp], a1, Wb[qp]);
                    NW(f, qp) = to_f2(a0); NW(f + 1, qp) = to_f2(a1);
                }
                if (warm) continue;
                v2f acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
                for (int n = P - 1; n >= 0; n--) {
#pragma unroll
                    for (int f = 0; f < 4; f++) {
                        const int i = f - n;
                        const float2 s2 = (i >= 0) ? NW(i, qp) : ring[2 * ((4 * PH + i) & 15) + qp];
                        const v2f sv = {s2.x, s2.y}, hv = {h[qp][n], h[qp][n]};
                        acc[f] = __builtin_elementwise_fma(sv, hv, acc[f]);
                    }
                }
#pragma unroll
                for (int f = 0; f < 4; f++) X[qp][f] = acc[f];
            }
            if (!warm) {
                // my two branches alias onto the same bin n1 of the short DFT: their sum is my part of the fold
#pragma unroll
                for (int f = 0; f < 4; f++) Pb[(part * 4 + f) * NP + n1] = to_f2(X[0][f] + X[1][f]);
            }
            // frame 3 takes the slot of frame -13
#pragma unroll
            for (int qp = 0; qp < 2; qp++) ring[2 * ((4 * PH + 3) & 15) + qp] = n3[qp];
#undef NW
        }
        if (tr && (S1_TRACE != 2 || !rho)) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tq[3] = __builtin_amdgcn_s_memtime(); }
        if (rho) {
            // image s + 1 has landed: everything older than the eight DMA instructions of image s + 2 (when there is one)
            if (s + AHEAD < n_items) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };
    for (unsigned s0 = 0; s0 < nsteps; s0 += 4) {
        step(s0, std::integral_constant<int, 0>());
        if (s0 + 1 < nsteps) step(s0 + 1, std::integral_constant<int, 1>());
        if (s0 + 2 < nsteps) step(s0 + 2, std::integral_constant<int, 2>());
        if (s0 + 3 < nsteps) step(s0 + 3, std::integral_constant<int, 3>());
    }
    if (S1_TRACE && A.trace && w == 1) { __syncthreads(); for (int i = tid; i < 768; i += 512) A.trace[i] = trc[i]; }
    if (last == A.nb) {
        if (tid == 0) A.vend_out[0] = c;
        // the next call's window: frame -d behind the last item (phase (n_items - 1) & 3) sits in slot (4 (n_items & 3) - d) & 15
        auto put = [&](auto phc) {
            constexpr int PN = decltype(phc)::value;
#pragma unroll
            for (int i = 0; i < 13; i++)
#pragma unroll
                for (int qp = 0; qp < 2; qp++) A.uhist_out[i * 1024 + 256 * (2 * rho + qp) + j] = ring[2 * ((4 * PN - 13 + i) & 15) + qp];
        };
        switch (n_items & 3u) {
        case 0: put(std::integral_constant<int, 0>()); break;
        case 1: put(std::integral_constant<int, 1>()); break;
        case 2: put(std::integral_constant<int, 2>()); break;
        default: put(std::integral_constant<int, 3>()); break;
        }
        if (FM && !rho) {                               // a thread reads back what it wrote
            for (int m = lt; m < NP; m += 256) A.rp_out[G * m] = ST[m];
        }
    }
}

// What the DC state a run started without contributes to the owned channel next to DC over the run's first 32 frames: Y += cpre[w] x R (the
// chain is linear up to Y).  CF32: in place on the row (or the tile-major plane); FM: freqdem of the corrected side copies.
template <bool FM>
__global__ __launch_bounds__(64) void k_shard1024_dcfix(Shard1024Args A, const float2 *__restrict__ rt, unsigned NP)
{
    const unsigned TB = FM ? 8u : 4u;
    const unsigned w = blockIdx.x + 1u, fr = threadIdx.x;
    unsigned first, last;
    shard_bounds(A.nb, A.nruns, w, TB, first, last);
    if (fr >= 32u || A.mfix < 0) return;
    const float2 c = A.cpre[w];
    const float2 *R = rt + (size_t)(A.parity0 & 1u) * RUN1024_DCFIX_F * 4;      // a tile is 4 frames: every cold start begins on the call's parity
    auto corr = [&](unsigned i) { const float2 r = R[i * 4u]; return make_float2(c.x * r.x - c.y * r.y, c.x * r.y + c.y * r.x); };
    const size_t t = (size_t)4 * first + fr;            // the frame in the call
    if (t >= A.nf) return;
    const unsigned k = (unsigned)A.mfix;
    if (FM) {
        const float2 *sd = A.side + (size_t)w * RUN1024_DCFIX_F;
        const float2 p0 = sd[fr], p1 = sd[fr + 1u], d0 = corr(fr), d1 = corr(fr + 1u);
        const FmK fk = {A.tiny, A.fm_ref, A.pk.hp, A.pk.pi};
        reinterpret_cast<float *>(A.out)[(size_t)k * A.nf + t] =
            fm_sample(make_float2(p0.x + d0.x, p0.y + d0.y), make_float2(p1.x + d1.x, p1.y + d1.y), fk);
    } else {
        float2 *o = reinterpret_cast<float2 *>(A.out) + (A.tile_major ? ((t >> 4) * NP + k) * 16u + (t & 15u) : (size_t)k * A.nf + t);
        const float2 d1 = corr(fr + 1u);
        float2 y = *o;
        y.x += d1.x; y.y += d1.y;
        *o = y;
    }
}

}  // namespace

// whole 4-frame tiles; at least two blocks (16 / 8 tiles) per run: a run >= 1 spends six read-only tiles and four halo tiles on its start
uint32_t shard1024_runs(uint32_t nf, bool fm, uint32_t G, uint32_t cus)
{
    if ((G != 4 && G != 8) || (nf & 3u) || nf == 0) return 0;
    if (diag_env("CSDR_NO_SHARD1024")) return 0;
    const uint32_t nb = nf / 4, TB = fm ? 8u : 4u, nblk = (nb + TB - 1) / TB;
    if ((uint64_t)(1024 / G) * nf * (fm ? 4u : 8u) >= (1ull << 31)) return 0;       // the row stores use 32-bit buffer offsets
    uint32_t nruns = cus;
    if (const char *e = diag_env("CSDR_SHARD1024_RUNS")) { const uint32_t v = (uint32_t)atoi(e); if (v >= 1 && v < nruns) nruns = v; }   // tests: force runs onto short inputs
    const uint32_t per = fm ? 2u : 4u;                  // >= 16 tiles per run
    if (nruns > nblk / per) nruns = nblk / per;
    return nruns;                                       // 0: too short for this kernel
}

int shard1024_launch(const Run1024v2Host &h, bool fm, uint32_t nruns, hipStream_t s, KernelTimer *timer)
{
    Shard1024Args A{};
    A.x = h.x; A.out = h.out; A.taps_q = h.taps_q; A.tw = h.tw;
    A.uhist_in = h.uhist_in; A.uhist_out = h.uhist_out; A.vend_in = h.vend_in; A.vend_out = h.vend_out;
    A.rp_in = h.rp_in; A.rp_out = h.rp_out;
    A.nf = h.nf; A.nb = h.nf / 4; A.nruns = nruns; A.parity0 = h.parity0; A.tile_major = (!fm && h.tile_major) ? 1u : 0u;
    if (A.tile_major && (h.nf & 15u)) { set_error("k_shard1024: a tile-major plane takes whole 16-frame blocks"); return -1; }
    const double beta = h.dc_block ? h.beta : 0.0;
    A.alpha = h.dc_block ? (float)(1.0 - beta) : 0.0f; A.beta = (float)beta; A.l2beta = h.dc_block ? (float)std::log2(beta) : -1000.0f;
    for (int i = 0; i < 16; i++) A.b16[i] = (float)std::pow(beta, 16.0 * i);
    for (int i = 0; i < 17; i++) A.b256[i] = (float)std::pow(beta, 256.0 * i);
    A.fm_ref = h.fm_ref; A.tiny = 1e-37f;
    A.pk = phase_consts(1.0f);                          // unscaled polynomial (fm_quad scales a = min / max by ref)
    A.pk.hp *= h.fm_ref; A.pk.pi *= h.fm_ref; A.pk.ref = h.fm_ref;
    static const char *trace_file = S1_TRACE ? diag_env("CSDR_SHARD1024_TRACE") : nullptr;
    static unsigned long long *d_trace = nullptr;
    if (trace_file && !d_trace) CSDR_HIP(hipMalloc(&d_trace, 768 * sizeof(unsigned long long)));
    if (trace_file) { CSDR_HIP(hipMemsetAsync(d_trace, 0, 768 * sizeof(unsigned long long), s)); A.trace = d_trace; }
    // no warm-up windows: whole runs of >= 16 tiles (a cold start is 4 tiles deep and the correction covers 8 more), the state arrays there
    A.nowu = (h.rt && h.cpre && h.side && h.dc_block && nruns >= 2) ? 1u : 0u;
    A.cpre = h.cpre; A.side = h.side; A.mfix = h.mfix;
    int r;
    if (timer && (r = timer->begin(s))) return r;
    if (h.G == 8) {
        if (fm) hipLaunchKernelGGL((k_shard1024<true, 8>), dim3(nruns), dim3(512), 0, s, A);
        else hipLaunchKernelGGL((k_shard1024<false, 8>), dim3(nruns), dim3(512), 0, s, A);
    } else if (h.G == 4) {
        if (fm) hipLaunchKernelGGL((k_shard1024<true, 4>), dim3(nruns), dim3(512), 0, s, A);
        else hipLaunchKernelGGL((k_shard1024<false, 4>), dim3(nruns), dim3(512), 0, s, A);
    } else { set_error("k_shard1024: chan_stride %u is not built (4, 8)", h.G); return -1; }
    if (A.nowu && A.mfix >= 0) {
        if (fm) hipLaunchKernelGGL((k_shard1024_dcfix<true>), dim3(nruns - 1u), dim3(64), 0, s, A, h.rt, 1024u / h.G);
        else hipLaunchKernelGGL((k_shard1024_dcfix<false>), dim3(nruns - 1u), dim3(64), 0, s, A, h.rt, 1024u / h.G);
    }
    if (timer && (r = timer->end(s))) return r;         // (the bracket covers the correction kernel)
    CSDR_HIP(hipGetLastError());
    if (trace_file) {                                   // debug: the last launch's stamps, raw uint64 [role][96][4]
        std::vector<unsigned long long> hbuf(768);
        CSDR_HIP(hipStreamSynchronize(s));
        CSDR_HIP(hipMemcpy(hbuf.data(), d_trace, 768 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        if (FILE *f = fopen(trace_file, "wb")) { fwrite(hbuf.data(), sizeof(unsigned long long), hbuf.size(), f); fclose(f); }
    }
    return 0;
}

}  // namespace csdr
