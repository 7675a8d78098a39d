// Second-generation run kernel of the fused M = 1024 chain, INTERLEAVED-SHARD instantiations k_run1024v2<FM, G> only (round 3).
// Since round 6 the route table sends the strides G = 4, 8 to k_shard1024 (kernels_shard1024.hip) and only G = 2 comes here; the G = 4, 8
// instantiations stay built behind CSDR_NO_SHARD1024=1 (the A/B of profiles/r06_shard1024_call_sizes.txt).  The whole-band form of this
// kernel (a 128 KiB global staging block per workgroup, read back transposed: DESIGN of round 3, `git show 5b4db89:DESIGN.md` 4.2c) was
// replaced by k_run1024v3 in round 4 and its code was removed from this file in round 6.
//
//   raw CF32 x --DC blocker--> y --NCO pre-mix, 14-tap polyphase FIR--> X_t[j] --1024-point forward DFT (16 x 16 x 4), pruned to the
//              shard's bins--> Y_t[k], k = g (mod G) --per-channel freqdem--> out[1024 / G][nf]     (Liquid.chs:575-589, 828-862, 324-328)
//
// Built the way k_run256v2 is (kernels_fused_v2.hip): 256 threads, two workgroups per CU, a TILE is 4096 consecutive
// samples = 4 frames, DMA'd into one of two 32 KiB LDS buffers a tile ahead and transformed in place:
//   raw image (16-byte XOR swizzle) --serial DC scan per 16-sample run--> y' --column layout: thread j owns branches
//   j + 256 q, q = 0..3; frame state chain, pre-mix, FIR out of a 13-frame register window (52 float2) --> X in place
//   --pass 1, wave f = frame f: radix 16 over n = 64 a + b--> Z1 --pass 2, same wave: radix 16 over b = 4 c + d--> Z2
//   --pass 3, thread kk = k1 + 16 k2: radix 4 over d for all four frames--> Y[kk + 256 k3], frames 0..3 in registers.
// Passes 1 and 2 read and write only their wave's frame block (an LDS image a wave writes and then reads needs no
// barrier), so a tile takes 4 barriers.  The tail thread holds four consecutive frames of four channels: the previous
// frame of a channel is a register-wide read of the stash (no DPP).  A tile's 16 bytes per owned row are stored directly.
// The 56 taps of a thread's four branches do not fit next to the window: they are re-read per tile from a 64 KiB table
// (L2-resident, fully coalesced 16-byte loads; the row of a branch ends with its even-frame pre-mix phasor, the odd-frame one
// comes from a second table) at the top of the tile and
// are waited for before the next tile's DMA is issued, so that no wait the compiler places can reach the DMA.
#include "fused_v2_common.h"

#ifndef B2_ABLATE
#define B2_ABLATE 0      // timing experiments only: 1 no input DMA in the loop, 2 no output stores (staging and rows), 4 no freqdem, 8 no tap loads, 16 no FIR,
                         // 32 no DFT passes 1-2, 64 no DC scan, 128 no window shift, 256 block flush without its loads, 512 without its stores, 1024 no flush
#endif

#ifndef B2_FM_PACKED
#define B2_FM_PACKED 1    // freqdem on packed pairs (fm_quad) or one sample at a time (fm_sample)
#endif
// Every asm store of more than 64 bits ends with s_nop 1: the store reads its data VGPRs for two more wait states, hipcc pads
// that for its own stores but cannot see one inside inline asm (it reused the registers in the very next instruction: lanes
// 12..15 of every row of 16 lost the second dword).
#ifndef B2_STORE_MOD
#define B2_STORE_MOD ""  // cache policy bits of the output stores (experiments: " nt", " sc1", " sc0 sc1")
#endif

namespace csdr {
namespace {

constexpr int B2_M = 1024, B2_T4 = 4;              // channels, frames per tile
constexpr int B2_BUF = 4096;                       // float2 per tile buffer
constexpr int B2_TW1 = 2 * B2_BUF;                 // twiddles W1024^(k1 b) at [k1 - 1][b], k1 = 1..15: 960 (pass 2 finds its W64^(d k2) at [4 d - 1][4 k2])
constexpr int B2_ST = B2_TW1 + 960;                // FM: last Y frame of channel kk + 256 k3 at [kk][k3]: 1024 (the prologue's reduction scratch before that)
constexpr int B2_TT = B2_ST + 1024;                // 16 group totals
constexpr int B2_F2 = B2_TT + 16;                  // 10 192 float2 = 81 536 B: two workgroups per CU (81 920 each)
constexpr int B2_WU = 6, B2_HALO = 4;              // read-only warm-up tiles (DC state), window refill tiles (13 frames of history)

struct Run1024v2Args {
    const float2 *x;            // raw input of this call
    void *out;                  // [1024][out_stride] F32 (FM) or CF32
    const float4 *taps_q;       // [4 q][4 pieces][256 j]: the 14 taps h[(1023 - r) + 1024 n] of branch r = 256 q + j, then conj(nco phasor) of r at even frames; behind the 64 KiB: [4 q][256 j] float2, the phasor at odd frames
    const float2 *tw;           // e^{-j 2 pi i / 1024}
    const float2 *uhist_in; float2 *uhist_out;    // [13][1024] pre-mixed, DC-blocked window before / after the call
    const float2 *vend_in; float2 *vend_out;      // DC blocker state v1
    const float2 *rp_in; float2 *rp_out;          // [1024] freqdem r'
    uint32_t nf, nb, nruns, n0, parity0, out_stride;   // n0: tiles of the first half of the runs (0: even split)
    uint32_t g;                 // G > 1: shard index (informational: the tables carry the shift by g channels, see the kernel)
    float alpha, beta, l2beta, fm_ref, tiny;
    float b16[16];              // beta^(16 r)
    float b256[17];             // beta^(256 g)
    PhaseK pk;
};

// Run w of a launch.  nruns = two per CU: the first half (the workgroups dispatched first, the older ones of their CUs) win the
// issue arbitration and get n0 of the nb tiles, the second half the rest (k_run256v2: both end together at about 1.2 : 0.8).
// Output blocks are aligned to absolute tile indices, so a run may start and end inside a block (partial lines there).
__host__ __device__ __forceinline__ void run_bounds(uint32_t nb, uint32_t nruns, uint32_t n0, unsigned w, unsigned &first, unsigned &last)
{
    if (n0 == 0 || (nruns & 1u)) {
        first = (unsigned)((unsigned long long)w * nb / nruns);
        last = (unsigned)((unsigned long long)(w + 1) * nb / nruns);
        return;
    }
    const unsigned half = nruns / 2, s = w / half, i = w - s * half;
    const unsigned base = s ? n0 : 0u, tiles = s ? nb - n0 : n0;
    first = base + (unsigned)((unsigned long long)i * tiles / half);
    last = base + (unsigned)((unsigned long long)(i + 1) * tiles / half);
}

// G > 1: INTERLEAVED CHANNEL SHARD g of G (G = 2, 4, 8; FM output).  The host folds a shift of the spectrum by g channels,
// W1024^(r g) on branch r, into the pre-mix phasors of the tap table (free: the FIR is linear), so that this kernel -- and the
// whole-band kernels of the same plan -- see Y'[k'] = Y[k' + g] and the shard owns k' = k1' + 16 k2 + 256 k3 with k1' = 0 mod G:
// it wants the pass-1 rows k1' = 0, G, 2G, ... only (the butterfly prunes itself: G is a template parameter), pass 2 runs
// in the lanes of those rows, pass 3 + freqdem in the threads kk = k1' + 16 k2 with k1' = 0 mod G, and every state array of the
// plan (freqdem history, window) is indexed by the PRIMED channel k' = k1' + 16 k2 + 256 k3, as the whole-band kernels see it
// behind the same tables.  Output: the shard's own [1024 / G][nf] plane, row k1'/G + (16/G) k2 + (256/G) k3, 16-byte pieces stored
// directly: 512 workgroups x 1024/G rows keep 64/G MiB of lines open, which the L2s hold for G = 8 and mostly for G = 4 (the
// staging block of the whole-band path exists because 64 MiB are not held).
template <bool FM, int G>
__global__ __launch_bounds__(256, 2) void k_run1024v2(Run1024v2Args A)
{
    static_assert(FM && (G == 2 || G == 4 || G == 8), "interleaved shards only: F32 output, G = 2, 4, 8");
    __shared__ __attribute__((aligned(16))) float2 L[B2_F2];
    float2 *tw1 = L + B2_TW1, *ST = L + B2_ST, *Tt = L + B2_TT, *red = ST;
    const int tid = threadIdx.x, j = tid;
    const unsigned w = blockIdx.x;
    unsigned first, last;
    run_bounds(A.nb, A.nruns, A.n0, w, first, last);
    const float4 *x4 = reinterpret_cast<const float4 *>(A.x);
    const int col_off = 16 * (j >> 4) + 2 * (((j & 15) >> 1) ^ (j >> 5)) + (j & 1);

#pragma unroll
    for (int i = 0; i < 4; i++) { const int e = tid + 256 * i; if (e < 960) tw1[e] = A.tw[(((e >> 6) + 1) * (e & 63)) & 1023]; }

    float2 hist[52];                                    // window: frames -13 .. -1 of my four branches, [13][4]
    float2 c;                                           // DC state v before the next tile (same in every lane)

    // ------------------------------------------------------------------ run start
    unsigned tile_begin = first;
    if (w == 0) {
        c = A.vend_in[0];
#pragma unroll
        for (int i = 0; i < 52; i++) hist[i] = A.uhist_in[(i >> 2) * B2_M + 256 * (i & 3) + j];
    } else {
        // read-only warm-up: the DC state before tile first - 4 from the six tiles in front of it (beta^24576 = 4.6e-6 of
        // the older state is dropped, as in every run kernel); then four halo tiles refill the window without output
        tile_begin = first - B2_HALO;
        const unsigned h0 = tile_begin - B2_WU;
        float2 acc = make_float2(0.f, 0.f);
        {
            // round 3 (as k_run256v2): the six tiles in ONE batch of loads (the window's registers are not live yet), weights of my
            // pieces as a running product: beta^(4095 - n), n = n0 + 512 it
            float4 raw[8], rb[8], rc[8], rd[8], re[8], rf[8];
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
            static_assert(B2_WU == 6, "one batch of six warm-up tiles");
            tile_load(x4 + (size_t)h0 * 2048, 256, raw, tid); tile_load(x4 + (size_t)(h0 + 1) * 2048, 256, rb, tid);
            tile_load(x4 + (size_t)(h0 + 2) * 2048, 256, rc, tid); tile_load(x4 + (size_t)(h0 + 3) * 2048, 256, rd, tid);
            tile_load(x4 + (size_t)(h0 + 4) * 2048, 256, re, tid); tile_load(x4 + (size_t)(h0 + 5) * 2048, 256, rf, tid);
            fold(raw); fold(rb); fold(rc); fold(rd); fold(re); fold(rf);
        }
        c = wg_sum(acc, red, tid);
        if (h0 == 0) c = cfma(A.vend_in[0], exp2f((float)(4096 * B2_WU) * A.l2beta), c);
#pragma unroll
        for (int i = 0; i < 52; i++) hist[i] = make_float2(0.f, 0.f);
    }
    if (FM) {                                           // freqdem history (after the reduction scratch is done with)
#pragma unroll
        for (int k3 = 0; k3 < 4; k3++) ST[4 * tid + k3] = (w == 0) ? A.rp_in[tid + 256 * k3] : make_float2(0.f, 0.f);
    }
    __syncthreads();                                    // twiddle tables, stash

    // ------------------------------------------------------------------ per-thread constants of the tile loop
    const float kJ = -A.alpha * exp2f((float)j * A.l2beta);                     // -alpha beta^j: group state into column j
    const float b256 = A.b256[1];
    const bool odd0 = (A.parity0 & 1) != 0;
    // atan polynomial: uniform values out of the kernel arguments (SGPR operands of the packed ops; the window leaves no VGPRs for them)
    const FmK2 fk = {{A.pk.c[0], A.pk.c[1], A.pk.c[2], A.pk.c[3], A.pk.c[4], A.pk.c[5], A.pk.c[6], A.pk.c[7]}, A.tiny, A.fm_ref, A.pk.hp, A.pk.pi};
    const unsigned goff = dma_offset(tid);
    const unsigned joff = 16u * (unsigned)j;                                    // my float4 in a 256-entry row of the tap table
    const __amdgpu_buffer_rsrc_t taps_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(A.taps_q), 0, 65536 + 8192, 0x00020000);
    const unsigned wave_u = (unsigned)__builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_wave = (unsigned)(size_t)(__attribute__((address_space(3))) float2 *)L + 1024u * wave_u;
    // LDS byte offsets inside a tile buffer that do not change from tile to tile
    const int q = tid, sw = (q >> 1) & 7;
    const unsigned raw_a0 = (unsigned)q * 128u + ((unsigned)sw << 4);            // slot i of my run: raw_a ^ (i << 4)
    const unsigned fb = 8192u * wave_u;                                         // passes 1-2: my wave's frame block
    const int b1 = tid & 63;                                                    // pass 1: n = 64 a + b1
    // X[f][64 a + b1] sits at fb + 512 a + (x_a ^ ((a & 3) << 5))  (column layout of the raw image, see col_off)
    const unsigned x_a0 = 8u * (unsigned)((16 * (b1 >> 4)) | (b1 & 1) | (2 * ((((b1 & 15) >> 1) ^ (b1 >> 5)) & 7)));
    // Z1[k1][b = 4 c + d]: reader lane l2 = 4 k1 + d sees its 16 values as eight swizzled 16-byte pairs, slot 8 l2 + ((c >> 1) ^ ((l2 >> 1) & 7))
    // writer (k1, b1): fb + 512 k1 + (z1w ^ (((2 k1) & 6) << 4))
    const unsigned z1w0 = 128u * (unsigned)(b1 & 3) + 8u * (unsigned)((b1 >> 2) & 1) + 16u * (unsigned)(((b1 >> 3) ^ ((b1 & 3) >> 1)) & 7);
    const int l2 = tid & 63, d2 = l2 & 3;                                       // pass 2: k1 = l2 >> 2, d = l2 & 3
    const unsigned z1r0 = fb + (unsigned)l2 * 128u + ((unsigned)((l2 >> 1) & 7) << 4);   // pair i: z1r ^ (i << 4)
    const unsigned z2w = fb + 8u * (unsigned)l2;                                // Z2[k1][k2][d] at 4 (k1 + 16 k2) + d: + 512 k2
    const unsigned z2r = 32u * (unsigned)tid;                                   // pass 3: thread kk reads 4 d's of frame f at 8192 f + 32 kk
    // two 16-bit LDS / table offsets per register, unpacked next to their use: the window leaves no VGPRs for nine of them
    const unsigned pk0 = raw_a0 | (x_a0 << 16), pk1 = z1r0 | (z1w0 << 16), pk2 = z2w | (z2r << 16), pk3 = goff | (joff << 16);
    const uint32_t esz = FM ? 4u : 8u;

    auto tile = [&](unsigned b_, const int par, const bool warm, const bool mute) {
        unsigned b = (unsigned)__builtin_amdgcn_readfirstlane((int)b_);            // keep the tile index (store / DMA bases) in SGPRs
        asm volatile("" : "+s"(b));
        char *B = reinterpret_cast<char *>(L) + par * (B2_BUF * 8);             // this tile's buffer
        float2 *Bf = reinterpret_cast<float2 *>(B);
        // the swizzled addresses (base ^ constant) are re-derived inside the tile: hoisted out of the loop they pin ~24 VGPRs
        auto unpack = [](unsigned pk, unsigned &lo, unsigned &hi) { asm volatile("" : "+v"(pk)); lo = pk & 0xffffu; hi = pk >> 16; };
        unsigned goff_t, joff_t;
        unpack(pk3, goff_t, joff_t);
        // taps (and the pre-mix phasor) of a branch: four 16-byte loads.  Branches 0 and 1 fly during the DC scan, 2 and 3 are
        // requested between the FIR passes (the window leaves no room for all 64 registers at once); the fences keep the
        // scheduler from hoisting them.  The next tile's DMA is only issued after the last of them has been used: the
        // compiler's waits count vmcnt in order and would otherwise wait for the DMA as well.
        auto load_taps = [&](v4f (&t)[4], v2f &wodd, const int qq) {
            if (B2_ABLATE & 8) { for (int p = 0; p < 4; p++) t[p] = (v4f){kJ, kJ, kJ, kJ}; wodd = (v2f){kJ, kJ}; return; }
#pragma unroll
            for (int p = 0; p < 4; p++) {               // buffer load: resource + 32-bit lane offset + scalar offset (one VGPR for all twenty loads)
                typedef unsigned v4u __attribute__((ext_vector_type(4)));
                const v4u v = __builtin_amdgcn_raw_buffer_load_b128(taps_rsrc, (int)joff_t, (qq * 4 + p) * 4096, 0);
                t[p] = __builtin_bit_cast(v4f, v);
            }
            typedef unsigned v2u __attribute__((ext_vector_type(2)));
            const v2u v = __builtin_amdgcn_raw_buffer_load_b64(taps_rsrc, (int)(joff_t >> 1), 65536 + qq * 2048, 0);
            wodd = __builtin_bit_cast(v2f, v);
        };
        v4f tq0[4], tq1[4], tq2[4], tq3[4];
        v2f wo0, wo1, wo2, wo3;
        load_taps(tq0, wo0, 0); load_taps(tq1, wo1, 1);
        bar();                                          // B_a: the tile image has landed (every wave waited for its own DMA); the other buffer is free
        // ---- DC blocker inside a 256-sample group: thread q owns the run of 16 consecutive samples q (as k_run256v2)
        unsigned raw_a, x_a;
        unpack(pk0, raw_a, x_a);
        const float na = opaque_v(-A.alpha), be = opaque_v(A.beta);             // VGPR copies for the scan only (an SGPR operand costs an issue slot more)
        v4f xr[8];
        float2 s = make_float2(0.f, 0.f);
        if (!(B2_ABLATE & 64)) {
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
        if ((raw_a & 0x780u) == 0x780u) Tt[raw_a >> 11] = s;    // (q & 15) == 15: Tt[q >> 4], from the tile-local register
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
        // ---- column layout: nw[4 f + qq] = sample of frame f, branch j + 256 qq; group state chain V[g] (uniform), a frame
        // (four groups) at a time: left to itself the scheduler fetches all sixteen totals first, 32 registers the window does not leave
        float2 nw[16];
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
                    nw[g] = to_f2(__builtin_elementwise_fma(V, kJv, to_v(Bf[256 * g + col_off])));
                    V = __builtin_elementwise_fma(V, bv, tg[qq]);
                }
                asm volatile("" ::: "memory");
            }
            // the state is the same in every lane: it waits for the next tile in SGPRs
            c = make_float2(__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(V.x))), __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(V.y))));
        }
        // pre-mix, then the polyphase FIR on the pre-mixed window: one branch at a time, four independent accumulators = its
        // four frames; X goes where the thread's column came from
        auto branch = [&](const v4f (&t)[4], const v2f wodd, const int qq) {
            // phasors of the frames with even / odd index in the tile: the table's even / odd global frames, swapped when the call starts odd
            const v2f we = {t[3].z, t[3].w};
            const v2f Wa = odd0 ? wodd : we, Wb = odd0 ? we : wodd;
#pragma unroll
            for (int f = 0; f < 4; f += 2) {
                v2f a0 = to_v(nw[4 * f + qq]), a1 = to_v(nw[4 * (f + 1) + qq]);
                cmul2_v(a0, Wa, a1, Wb);
                nw[4 * f + qq] = to_f2(a0); nw[4 * (f + 1) + qq] = to_f2(a1);
            }
            if (warm) return;
            const float h[16] = {t[0].x, t[0].y, t[0].z, t[0].w, t[1].x, t[1].y, t[1].z, t[1].w, t[2].x, t[2].y, t[2].z, t[2].w, t[3].x, t[3].y, 0.f, 0.f};
            v2f acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
            for (int n = ((B2_ABLATE & 16) ? 0 : P - 1); n >= 0; n--) {
#pragma unroll
                for (int f = 0; f < 4; f++) {
                    const int i = f - n;
                    const float2 s2 = (i >= 0) ? nw[4 * i + qq] : hist[4 * (13 + i) + qq];
                    const v2f sv = {s2.x, s2.y}, hv = {h[n], h[n]};
                    acc[f] = __builtin_elementwise_fma(sv, hv, acc[f]);
                }
            }
#pragma unroll
            for (int f = 0; f < 4; f++) Bf[256 * (4 * f + qq) + col_off] = to_f2(acc[f]);
        };
        branch(tq0, wo0, 0);
        asm volatile("" ::: "memory");
        load_taps(tq2, wo2, 2);
        asm volatile("" ::: "memory");
        branch(tq1, wo1, 1);
        asm volatile("" ::: "memory");
        load_taps(tq3, wo3, 3);
        asm volatile("" ::: "memory");
        branch(tq2, wo2, 2);
        branch(tq3, wo3, 3);
        asm volatile("" ::: "memory");
        if (!(B2_ABLATE & 1) && b + 1 < last) dma_tile(x4 + (size_t)(b + 1) * 2048, goff_t, lds_wave + (unsigned)(par ^ 1) * (B2_BUF * 8u));
        // the window moves on by four frames
        if (!(B2_ABLATE & 128)) {
#pragma unroll
            for (int i = 0; i < 36; i++) hist[i] = hist[i + 16];
#pragma unroll
            for (int i = 0; i < 16; i++) hist[36 + i] = nw[i];
        }
        if (warm) return;
        bar();                                          // B_d: X complete
        // ---- DFT pass 1: wave f, lane b1: radix 16 over a (n = 64 a + b1); Z1 goes back into the frame block, nobody else reads it
        unsigned z1r, z1w, z2w_t, z2r_t;
        unpack(pk1, z1r, z1w);
        unpack(pk2, z2w_t, z2r_t);
        v2f vv[16];
        if (!(B2_ABLATE & 32)) {
#pragma unroll
        for (int a = 0; a < 16; a++) vv[a] = to_v(*reinterpret_cast<const float2 *>(B + fb + 512 * a + (x_a ^ (unsigned)((a & 3) << 5))));
        fft16_v(vv);
#pragma unroll
        for (int i = 1; i < 16; i++)                // rows k1' = 0 mod G only (the spectrum arrives shifted by g: see the tables)
            if (XIDX(i) % G == 0) vv[i] = cmul_v(vv[i], to_v(tw1[64 * (XIDX(i) - 1) + b1]));
#pragma unroll
        for (int i = 0; i < 16; i++)
            if (XIDX(i) % G == 0) *reinterpret_cast<float2 *>(B + fb + 512 * XIDX(i) + (z1w ^ (unsigned)(((2 * XIDX(i)) & 6) << 4))) = to_f2(vv[i]);
        // ---- DFT pass 2: same wave, lane (k1, d): radix 16 over c (b = 4 c + d); a shard's lanes of the rows it did not write sit it out
        if (((l2 >> 2) % G) == 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const v4f v = *reinterpret_cast<const v4f *>(B + (z1r ^ (unsigned)(i << 4)));
            vv[2 * i] = (v2f){v.x, v.y}; vv[2 * i + 1] = (v2f){v.z, v.w};
        }
        fft16_v(vv);                                    // vv[i] = k2 = XIDX(i)
        if (d2) {                                       // W64^(d k2); lanes d = 0 sit this out
#pragma unroll
            for (int i = 1; i < 16; i++) vv[i] = cmul_v(vv[i], to_v(tw1[64 * (4 * d2 - 1) + 4 * XIDX(i)]));
        }
#pragma unroll
        for (int i = 0; i < 16; i++) *reinterpret_cast<float2 *>(B + z2w_t + 512 * XIDX(i)) = to_f2(vv[i]);
        }
        }
        bar();                                          // B_h: Z2 of all four frames complete
        // ---- DFT pass 3 + tail: thread kk = k1 + 16 k2, all four frames; Y[f][k3] = channel kk + 256 k3
        if (((tid & 15) % G) != 0) return;              // threads of the rows the shard does not own are done (no barrier follows on this path)
        v2f y[4][4];
#pragma unroll
        for (int f = 0; f < 4; f++) {
            const v4f v0 = *reinterpret_cast<const v4f *>(B + 8192 * f + z2r_t);
            const v4f v1 = *reinterpret_cast<const v4f *>(B + 8192 * f + z2r_t + 16);
            y[f][0] = (v2f){v0.x, v0.y}; y[f][1] = (v2f){v0.z, v0.w}; y[f][2] = (v2f){v1.x, v1.y}; y[f][3] = (v2f){v1.z, v1.w};
            bfly4_v(y[f][0], y[f][1], y[f][2], y[f][3]);
        }
        // ---- tail: freqdem, then every owned row gets its 16 bytes of this tile
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // next tile image: nothing else is outstanding
        if (FM) {
            char *stp = reinterpret_cast<char *>(ST) + z2r_t;      // my 32 bytes of the stash: [kk][k3]
            const v4f p01 = *reinterpret_cast<const v4f *>(stp), p23 = *reinterpret_cast<const v4f *>(stp + 16);
            const float2 prev[4] = {make_float2(p01.x, p01.y), make_float2(p01.z, p01.w), make_float2(p23.x, p23.y), make_float2(p23.z, p23.w)};
            *reinterpret_cast<v4f *>(stp) = (v4f){y[3][0].x, y[3][0].y, y[3][1].x, y[3][1].y};
            *reinterpret_cast<v4f *>(stp + 16) = (v4f){y[3][2].x, y[3][2].y, y[3][3].x, y[3][3].y};
            if (mute) return;                           // the tile in front of the run: only its last frame was wanted (freqdem history)
            // tile-local copies of the uniform scalings: as loop invariants they end up as VGPRs in the spill area
            FmK2 fkt = fk;
            asm volatile("" : "+s"(fkt.ref), "+s"(fkt.hp), "+s"(fkt.pi), "+s"(fkt.tiny));
#pragma unroll
            for (int k3 = 0; k3 < 4; k3++) {
                // one channel after the other (volatile asms keep their order): four interleaved quads do not fit next to the window
                asm volatile("" : "+v"(y[0][k3]), "+v"(y[1][k3]), "+v"(y[2][k3]), "+v"(y[3][k3]));
                const float2 rp[4] = {prev[k3], to_f2(y[0][k3]), to_f2(y[1][k3]), to_f2(y[2][k3])};
                const float2 rr[4] = {to_f2(y[0][k3]), to_f2(y[1][k3]), to_f2(y[2][k3]), to_f2(y[3][k3])};
                float mq[4];
                if (B2_ABLATE & 4) { mq[0] = rp[0].x + rr[0].y; mq[1] = rp[1].y + rr[1].x; mq[2] = rp[2].x + rr[2].y; mq[3] = rp[3].y + rr[3].x; }
                else if (B2_FM_PACKED) fm_quad(rp, rr, fkt, mq);
                else {
                    const FmK k1s = {fkt.tiny, fkt.ref, fkt.hp, fkt.pi};
#pragma unroll
                    for (int u = 0; u < 4; u++) mq[u] = fm_sample(rp[u], rr[u], k1s);
                }
                const v4f mv = {mq[0], mq[1], mq[2], mq[3]};
                {
                    // row of (k1', k2, k3) in the shard's plane: k1'/G + (16/G) k2 + (256/G) k3; the k3 and frame terms are uniform
                    const unsigned mrow = ((unsigned)(tid & 15) / (unsigned)G + (16u / (unsigned)G) * ((unsigned)tid >> 4)) * A.out_stride * esz;
                    const char *rowg = reinterpret_cast<const char *>(A.out) + (size_t)4 * b * esz + (size_t)k3 * (256 / G) * A.out_stride * esz;
                    if (B2_ABLATE & 2) asm volatile("" :: "v"(mv), "s"(rowg));
                    else asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2" "\n\ts_nop 1" :: "v"(mrow), "v"(mv), "s"(rowg) : "memory");
                }
            }
        }
    };

    // Buffer parity: the main loop starts in buffer 1.  A run >= 1 walks three halo tiles (window and DC state only: buffers 0, 1, 0)
    // and then enters the main loop one tile early, muted: tile first - 1 goes through FIR and DFT for its last frame, the
    // freqdem history of the run's first sample (the window behind that frame is complete: 15 halo frames).
    if (tile_begin < last) dma_tile(x4 + (size_t)tile_begin * 2048, goff, lds_wave + (w == 0 ? (unsigned)(B2_BUF * 8u) : 0u));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned bm = first;
    if (w > 0) {
        tile(tile_begin, 0, true, false);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tile(tile_begin + 1, 1, true, false);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tile(tile_begin + 2, 0, true, false);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bm = first - 1;
    }
    for (unsigned b = bm; b < last; b += 2) {
        tile(b, 1, false, b < first);
        if (b + 1 >= last) break;
        tile(b + 1, 0, false, false);
    }

    // ------------------------------------------------------------------ state after the run
    if (FM) {                                           // a thread reads back what it wrote
#pragma unroll
        for (int k3 = 0; k3 < 4; k3++) {
            if (last == A.nb) A.rp_out[tid + 256 * k3] = ST[4 * tid + k3];
        }
    }
    if (last == A.nb) {
        if (tid == 0) A.vend_out[0] = c;
#pragma unroll
        for (int i = 0; i < 52; i++) A.uhist_out[(i >> 2) * B2_M + 256 * (i & 3) + j] = hist[i];
    }
}

}  // namespace

// one launch per call (called by big_process, kernels_pfb1024.hip); no fix-up kernel: a run computes the frame in front of it itself
int run1024_v2_launch(const Run1024v2Host &h, bool fm, hipStream_t s, KernelTimer *timer)
{
    Run1024v2Args A{};
    A.x = h.x; A.out = h.out; A.taps_q = h.taps_q; A.tw = h.tw;
    A.uhist_in = h.uhist_in; A.uhist_out = h.uhist_out; A.vend_in = h.vend_in; A.vend_out = h.vend_out;
    A.rp_in = h.rp_in; A.rp_out = h.rp_out;
    A.nf = h.nf; A.nb = h.nf / B2_T4; A.nruns = h.nruns; A.parity0 = h.parity0; A.out_stride = h.nf;
    {
        static const double wt = diag_env("CSDR_RUN1024_WEIGHT") ? atof(diag_env("CSDR_RUN1024_WEIGHT")) : 1.2;   // share of the older workgroup of a CU (1 = even)
        A.n0 = (h.nruns >= 2 && !(h.nruns & 1u) && wt > 1.0 && wt < 1.5) ? (uint32_t)std::llround(0.5 * wt * (double)A.nb) : 0u;
    }
    const double beta = h.dc_block ? h.beta : 0.0;
    A.alpha = h.dc_block ? (float)(1.0 - beta) : 0.0f; A.beta = (float)beta; A.l2beta = h.dc_block ? (float)std::log2(beta) : -1000.0f;
    for (int i = 0; i < 16; i++) A.b16[i] = (float)std::pow(beta, 16.0 * i);
    for (int i = 0; i < 17; i++) A.b256[i] = (float)std::pow(beta, 256.0 * i);
    A.fm_ref = h.fm_ref; A.tiny = 1e-37f;
    A.pk = phase_consts(1.0f);                          // unscaled polynomial (fm_quad scales a = min / max by ref)
    A.pk.hp *= h.fm_ref; A.pk.pi *= h.fm_ref; A.pk.ref = h.fm_ref;
    int r;
    if (timer && (r = timer->begin(s))) return r;
    A.g = h.g;
    if (h.G > 1 && !fm) { set_error("k_run1024v2: interleaved shards have F32 output only"); return -1; }
    if (h.G == 2) hipLaunchKernelGGL((k_run1024v2<true, 2>), dim3(h.nruns), dim3(256), 0, s, A);
    else if (h.G == 4) hipLaunchKernelGGL((k_run1024v2<true, 4>), dim3(h.nruns), dim3(256), 0, s, A);
    else if (h.G == 8) hipLaunchKernelGGL((k_run1024v2<true, 8>), dim3(h.nruns), dim3(256), 0, s, A);
    else if (h.G > 1) { set_error("k_run1024v2: interleaved shards of stride %u are not built (2, 4, 8)", h.G); return -1; }
    else { set_error("k_run1024v2: interleaved shards only (whole band: k_run1024v3 / k_run1024)"); return -1; }
    if (timer && (r = timer->end(s))) return r;
    CSDR_HIP(hipGetLastError());
    return 0;
}

uint32_t run1024_v2_runs(uint32_t nf, uint32_t cus)
{
    // two workgroups per CU; a run >= 1 spends 6 read-only + 4 halo tiles on its start state; the shorter (younger) runs get
    // up to 1/4 less than the average: at least 24 tiles per run on average (>= 16 for every one)
    const uint32_t nb = nf / B2_T4;
    uint32_t nruns = 2 * cus;
    if (const char *e = diag_env("CSDR_RUN1024_RUNS")) { const uint32_t v = (uint32_t)atoi(e); if (v >= 1 && v < nruns) nruns = v; }   // experiments
    if (nruns > nb / 24) nruns = nb / 24;
    if (nruns > 2) nruns &= ~1u;
    return nruns;                                       // 0: too short for this kernel
}

}  // namespace csdr
