// Third-generation run kernel of the fused M = 256 chain: ONE 512-thread workgroup per compute unit, split by wave role.
//
//   raw CF32 x --DC blocker--> y --NCO pre-mix, 14-tap polyphase FIR--> X_t[j] --256-point forward DFT (16 x 16)--> Y_t[k]
//              --per-channel freqdem--> out[256][nf]                (8 B read + 4 / 8 B written per sample, Liquid.chs:575-589,
//                                                                     828-862, 324-328)
//
// Why (verdict r03 #1 (i)): k_run256v2 runs two 256-thread workgroups per CU, i.e. 512 runs per launch, and every
// run starts cold: six read-only warm-up tiles + the halo tile = 224 KiB of input that produces no output (134 MB of 671 MB
// read per 67 M-sample launch), all of it requested in one burst while the ALUs idle.  Here a CU holds ONE run of twice the
// length (256 runs per launch: half the cold-start bytes), and its eight waves are a two-stage pipeline instead of two copies
// of the whole tile body:
//   FRONT waves 0-3 (thread = polyphase branch): tile DMA, DC blocker, pre-mix, FIR of tile t   -> X(t) in the tile's buffer
//   BACK  waves 4-7 (thread = (k1, frame)):      DFT passes 1 + 2, freqdem, stores of tile t-1  <- X(t-1) from its buffer
// A SIMD hosts one wave of each role, so the two waves that share its issue port are by construction in DIFFERENT phases of
// the tile (v2's two workgroups drift in and out of phase), the FIR window (front) and the freqdem history + paired F32 rows
// (back) stay in the registers of the waves that use them, and a step of the pipeline needs 2 workgroup barriers (v2: 4 per
// tile and workgroup).  The back waves never wait for memory: the DMA lives in the front waves' vmcnt queue only.
//
// LDS: V3_NBUF tile buffers of 32 KiB, tile t in buffer t % V3_NBUF from its DMA (issued V3_DIST = V3_NBUF - 2 tiles ahead) to
// the back waves' last Z read; layouts inside a buffer exactly as in k_run256v2 (raw image with a 16-byte XOR swizzle -> y' in
// place -> X in place -> Z in place).  Same arithmetic as k_run256v2, instruction for instruction, per tile.
#ifndef V3_NBUF
#define V3_NBUF 4
#endif
#ifndef V3_ABLATE
#define V3_ABLATE 0      // timing experiments only: 1 no input DMA in the loop, 2 no output stores
#endif
#ifndef V3_SNOP
// Wait states in front of the asm stores would cover a scalar base that comes straight out of a spill lane (v_readlane -> VMEM
// hazard, fused_v2_common.h); this kernel has no SGPR spills (the bases are SALU results of the tile index): empty, and
// tests/test_build_invariants.py fails when a change makes hipcc spill SGPRs here.
#define V3_SNOP ""
#endif

#include "fused_v2_common.h"

namespace csdr {
namespace {

constexpr int V3_BUF = 4096;                                 // float2 per tile buffer (32 KiB)
constexpr int V3_DIST = V3_NBUF - 2;                         // the DMA of tile t + V3_DIST is issued at the start of step t
static_assert(V3_NBUF == 3 || V3_NBUF == 4, "three or four tile buffers");
// tile buffers, then: STASH 256, T 16, RED 16, pass-1 twiddles 256
constexpr int V3_STASH = V3_NBUF * V3_BUF;
constexpr int V3_F2 = V3_STASH + 256 + 32 + 256;             // 16 928 float2 = 135 424 B (four buffers): one workgroup per CU

struct V3Args {
    RunArgs r;
};

__device__ __forceinline__ float2 wg_sum8(float2 v, float2 *red, int tid)      // sum over the 512 threads (8 waves)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { v.x += __shfl_xor(v.x, d); v.y += __shfl_xor(v.y, d); }
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float2 r = red[0];
#pragma unroll
    for (int i = 1; i < 8; i++) r = cadd(r, red[i]);
    __syncthreads();
    return r;
}

template <int N> __device__ __forceinline__ void wait_vmcnt()
{
    if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

template <bool FM>
__global__ __launch_bounds__(512) void k_run256v3(V3Args VA)
{
    const RunArgs &RA = VA.r;
    const TileArgs &A = RA.t;
    __shared__ __attribute__((aligned(16))) float2 L[V3_F2];
    float2 *ST = L + V3_STASH, *Tt = ST + 256, *red = Tt + 16;
    float2 *tw_s = red + 16;
    float2 *H = L + (V3_NBUF - 1) * V3_BUF;             // run start only: the halo tile's image, then scratch of the one-frame DFT (the last buffer is free until step 1's DMA)
    float2 *E = ST;                                     // run start only: 256 run carries (the stash area, before the stash is initialised)
    const int tid = threadIdx.x;
    const bool front = tid < 256;                       // wave-uniform role
    const int lt = tid & 255, j = lt;                   // thread index inside the role; front: polyphase branch
    const unsigned w = blockIdx.x;
    unsigned first, last;
    run_range(RA.split, w, first, last);
    if (RA.pair_align) {                                // runs start on even tiles: a tile pair fills whole 128-byte lines of the F32 rows
        first &= ~1u;
        if (last != A.nb) last &= ~1u;
    }
    const unsigned nt = last - first;
    const float4 *x4 = reinterpret_cast<const float4 *>(A.x);
    const int col_off = 16 * (j >> 4) + 2 * (((j & 15) >> 1) ^ (j >> 5)) + (j & 1);
    const unsigned goff = dma_offset(lt);
    const unsigned wave_r = (unsigned)__builtin_amdgcn_readfirstlane(lt >> 6);      // wave index inside the role
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float2 *)L + 1024u * wave_r;

    // Everything the run start waits for is requested up front (one burst of memory traffic, not a chain of round trips):
    // the first V3_DIST tiles and the halo tile by DMA (front waves), the warm-up window as plain loads (all 512 threads).
    // cold run start: DC state from a read-only warm-up window, FIR window and freqdem history from the halo tile.  Run 0
    // instead carries the exact state of the previous call -- unless the call is INDEPENDENT of the previous launch
    // (RA.indep, csdr_chain_submit_device): then run 0 is a cold start like any other, its window being the previous
    // chunk's last WU + 1 tiles, kept in RA.prev_tail (tiles -1 .. -(WU + 1) of this chunk).
    const bool cold = w > 0 || RA.indep;
    const int halo = (int)first - 1;
    auto tile_ptr = [&](int t) -> const float4 * { return t >= 0 ? x4 + (size_t)t * 2048 : RA.prev_tail + (size_t)(WU + 1 + t) * 2048; };
    if (front) {
#pragma unroll
        for (int d = 0; d < V3_DIST; d++)
            if (first + d < last) dma_tile(x4 + (size_t)(first + d) * 2048, goff, lds0 + (unsigned)d * (V3_BUF * 8u));
        if (cold) dma_tile(tile_ptr(halo), goff, lds0 + (unsigned)(V3_NBUF - 1) * (V3_BUF * 8u));
    }
    float h[P];
    float2 Wa = make_float2(0.f, 0.f), Wb = make_float2(0.f, 0.f);
    if (front) {
#pragma unroll
        for (int n = 0; n < P; n++) h[n] = A.taps[(M256 - 1 - j) + n * M256];
        Wa = A.wpre[(A.parity0 & 1) * M256 + j]; Wb = A.wpre[((A.parity0 & 1) ^ 1) * M256 + j];
    } else {
#pragma unroll
        for (int n = 0; n < P; n++) h[n] = 0.f;
        tw_s[lt] = A.tw[lt];
    }
    float2 wa[NB], wb[NB];                              // front: FIR window halves: one holds the previous tile, the other the new one
#pragma unroll
    for (int f = 0; f < NB; f++) { wa[f] = make_float2(0.f, 0.f); wb[f] = make_float2(0.f, 0.f); }
    float2 c = make_float2(0.f, 0.f);                   // front: DC state v before the next tile (same in every lane)
    float2 w2 = make_float2(0.f, 0.f);

    // ------------------------------------------------------------------ run start (arithmetic of k_run256v2; barriers are workgroup-wide)
    if (!cold) {
        if (front) {
            c = A.vend_in[0];
#pragma unroll
            for (int f = 3; f < NB; f++) wa[f] = A.yhist_in[(f - 3) * M256 + j];
        }
    } else {
        const int h0 = RA.indep ? halo - (int)RA.wu : (halo > (int)RA.wu ? halo - (int)RA.wu : 0);
        const unsigned nwu = (unsigned)(halo - h0);
        // read-only warm-up: thread tid takes the 16-byte pieces tid + 512 it (it = 0..3) of a tile, samples n = 2 (tid + 512 it) and n + 1,
        // weight beta^(4095 - n): two registers and a uniform ratio beta^-1024 per step
        const float wt0 = exp2f((float)(4095 - 2 * tid) * RA.l2beta), wt1 = exp2f((float)(4094 - 2 * tid) * RA.l2beta);
        const float wstep = RA.l2beta < -100.0f ? 0.0f : exp2f(-1024.0f * RA.l2beta);   // (dc_block off, l2beta = -1000: no state to warm up, and no inf * 0)
        float2 acc = make_float2(0.f, 0.f);
        auto fold = [&](const float4 (&r)[4], int t) {
            float2 p = make_float2(0.f, 0.f);
            float a0 = wt0, a1 = wt1;
#pragma unroll
            for (int it = 0; it < 4; it++) {
                p = cfma(make_float2(r[it].x, r[it].y), a0, p);
                p = cfma(make_float2(r[it].z, r[it].w), a1, p);
                a0 *= wstep; a1 *= wstep;
            }
            acc = cfma(p, exp2f((float)(4096 * (halo - 1 - t)) * RA.l2beta), acc);
        };
        auto load4 = [&](int t, float4 (&r)[4]) {
            const float4 *src = tile_ptr(t);
#pragma unroll
            for (int it = 0; it < 4; it++) r[it] = src[tid + 512 * it];
        };
        unsigned i = 0;
        if (nwu == 6) {
            // the usual window: all six tiles requested before the first is folded: one memory latency behind the DMA'd tiles
            float4 r0[4], r1[4], r2[4], r3[4], r4[4], r5[4];
            load4(h0, r0); load4(h0 + 1, r1); load4(h0 + 2, r2); load4(h0 + 3, r3); load4(h0 + 4, r4); load4(h0 + 5, r5);
            fold(r0, h0); fold(r1, h0 + 1); fold(r2, h0 + 2); fold(r3, h0 + 3); fold(r4, h0 + 4); fold(r5, h0 + 5);
            i = 6;
        }
        for (; i < nwu; i++) {
            float4 r0[4];
            load4(h0 + (int)i, r0);
            fold(r0, h0 + (int)i);
        }
        float2 ch = wg_sum8(acc, red, tid);
        if (h0 == 0 && !RA.indep) ch = cfma(A.vend_in[0], exp2f((float)(4096 * halo) * RA.l2beta), ch);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the DMA'd tiles (older than every warm-up load)
        __syncthreads();
        if (front) {                                    // zero-state DC scan of the halo image (scan_staged without its barrier)
            float4 *R4 = reinterpret_cast<float4 *>(H);
            const int q = lt, sw = (q >> 1) & 7;
            float2 s = make_float2(0.f, 0.f);
            const float na = -A.alpha, be = A.beta;
#pragma unroll
            for (int i2 = 0; i2 < 8; i2++) {
                float4 v = R4[8 * q + (i2 ^ sw)];
                float2 x0 = make_float2(v.x, v.y), x1 = make_float2(v.z, v.w);
                const float2 z0 = cfma(s, na, x0);
                s = cfma(s, be, x0);
                const float2 z1 = cfma(s, na, x1);
                s = cfma(s, be, x1);
                R4[8 * q + (i2 ^ sw)] = make_float4(z0.x, z0.y, z1.x, z1.y);
            }
            float2 t;
            t = dpp2<0x111>(s); s = cfma(t, A.b16[1], s);
            t = dpp2<0x112>(s); s = cfma(t, A.b16[2], s);
            t = dpp2<0x114>(s); s = cfma(t, A.b16[4], s);
            t = dpp2<0x118>(s); s = cfma(t, A.b16[8], s);
            E[q] = dpp2<0x111>(s);                      // v at my run's start (zero row carry)
            if ((q & 15) == 15) Tt[q >> 4] = s;         // row total
        }
        __syncthreads();
        if (front) {
#pragma unroll
            for (int f = 3; f < NB; f++) wa[f] = H[256 * f + col_off];
            w2 = H[256 * 2 + col_off];                  // FM: the 14th tap of the halo tile's last frame (freqdem history of the run)
            const float br = A.b16[lt & 15], bf = A.b256[lt >> 4];
            float2 vb, ve;
            frame_carries(Tt, A, lt, vb, ve);
            E[lt] = cfma(cfma(ch, bf, vb), br, E[lt]);
            c = cfma(ch, A.b256[16], ve);
        }
        __syncthreads();
        if (front) {
            const float kj = -A.alpha * A.bj[j & 15];
#pragma unroll
            for (int f = 3; f < NB; f++) wa[f] = cfma(E[16 * f + (j >> 4)], kj, wa[f]);
            w2 = cfma(E[16 * 2 + (j >> 4)], kj, w2);
        }
    }
    if (front) {
#pragma unroll
        for (int f = 3; f < NB; f++) wa[f] = cmul(wa[f], (f & 1) ? Wb : Wa);       // the window holds pre-mixed samples
    }
    __syncthreads();                                    // every front thread has read its run carries out of the stash area
    // freqdem history: stash[k1][i] = last Y frame of channel k1 + 16 XIDX(i); back thread lt owns channel lt
    const int st_idx = (lt & 15) * 16 + XIDX(lt >> 4);
    if (!front) ST[st_idx] = (FM && !cold) ? A.rp_in[lt] : make_float2(0.f, 0.f);
    __syncthreads();                                    // H free; stash and twiddles visible
    if (FM && cold) {
        // The run's first freqdem sample needs the frame in front of it: the halo tile's last frame goes through the FIR (front)
        // and a one-frame DFT (back) here, same arithmetic as the tile loop
        if (front) {
            w2 = cmul(w2, Wa);
            v2f acc = {0.f, 0.f};
#pragma unroll
            for (int n = P - 1; n >= 0; n--) {
                const float2 s2 = (n == P - 1) ? w2 : wa[NB - 1 - n];
                acc = __builtin_elementwise_fma((v2f){s2.x, s2.y}, (v2f){h[n], h[n]}, acc);
            }
            H[j] = to_f2(acc);
        }
        __syncthreads();
        v2f vv[16];
        if (!front && lt < 16) {
#pragma unroll
            for (int a = 0; a < 16; a++) vv[a] = to_v(H[16 * a + lt]);
            fft16_v(vv);
#pragma unroll
            for (int i = 1; i < 16; i++) vv[i] = cmul_v(vv[i], to_v(tw_s[16 * XIDX(i) + lt]));
#pragma unroll
            for (int i = 0; i < 16; i++) H[256 + 16 * XIDX(i) + lt] = to_f2(vv[i]);     // Z[k1][b1]
        }
        __syncthreads();
        if (!front && lt < 16) {
#pragma unroll
            for (int b = 0; b < 16; b++) vv[b] = to_v(H[256 + 16 * lt + b]);
            fft16_v(vv);                                // vv[i] = Y[k1 + 16 XIDX(i)]
#pragma unroll
            for (int i = 0; i < 16; i++) ST[lt * 16 + i] = to_f2(vv[i]);
        }
        __syncthreads();
    }
    if (front) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // non-cold run 0: the first tiles were requested at the kernel's entry

    if (front) {
        // ================================================================== FRONT: DMA, DC blocker, pre-mix, FIR
        const v2f Wav = to_v(Wa), Wbv = to_v(Wb);
        const float na = opaque_v(-A.alpha), be = opaque_v(A.beta);
        const float kJ = -A.alpha * exp2f((float)j * RA.l2beta);                    // -alpha beta^j: frame state into column j
        const float b256 = A.b256[1];
        const int q = lt, sw = (q >> 1) & 7;
        const unsigned raw_a = (unsigned)q * 128u + ((unsigned)sw << 4);            // slot i of my run: raw_a ^ (i << 4)
        auto fstep = [&](float2 (&old)[NB], float2 (&nw)[NB], unsigned s_) {
            unsigned s = (unsigned)__builtin_amdgcn_readfirstlane((int)s_);         // step = tile index inside the run, in SGPRs
            asm volatile("" : "+s"(s));
            bar();                                      // A: X(s - 1) complete; tile s landed (every front wave waited for its own DMA); buffer (s + V3_DIST) % V3_NBUF free
            if (s >= nt) { bar(); return; }             // the drain step: the back waves finish the last tile
            const unsigned b = first + s;
            char *B = reinterpret_cast<char *>(L) + (s % V3_NBUF) * (V3_BUF * 8u);  // this tile's buffer
            float2 *Bf = reinterpret_cast<float2 *>(B);
            if (!(V3_ABLATE & 1) && b + V3_DIST < last)
                dma_tile(x4 + (size_t)(b + V3_DIST) * 2048, goff, lds0 + ((s + V3_DIST) % V3_NBUF) * (V3_BUF * 8u));
            // ---- DC blocker inside a frame: thread q owns the run of 16 consecutive samples q (see k_run256v2)
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
            bar();                                      // B: y' (frame carry still missing) and the frame totals are visible
            // ---- column layout: thread j owns branch j; frame state chain V[f] (uniform), y = y' - alpha beta^j V[f], pre-mix
#pragma unroll
            for (int f = 0; f < NB; f++) nw[f] = Bf[256 * f + col_off];
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
            if (b + 1 == A.nb) {                        // the stream's last 13 frames of y
#pragma unroll
                for (int f = 3; f < NB; f++) A.yhist_out[(f - 3) * M256 + j] = nw[f];
            }
#pragma unroll
            for (int f = 0; f < NB; f += 2) {
                v2f a0 = to_v(nw[f]), a1 = to_v(nw[f + 1]);
                cmul2_v(a0, Wav, a1, Wbv);
                nw[f] = to_f2(a0); nw[f + 1] = to_f2(a1);
            }
            // ---- polyphase FIR on the pre-mixed window, four frames at a time; X goes where the thread's column came from
#pragma unroll
            for (int f0 = 0; f0 < NB; f0 += 4) {
                v2f acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
                for (int n = P - 1; n >= 0; n--) {
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int i = f0 + u - n;
                        const float2 s2 = (i >= 0) ? nw[i] : old[NB + i];
                        const v2f sv = {s2.x, s2.y}, hv = {h[n], h[n]};
                        acc[u] = __builtin_elementwise_fma(sv, hv, acc[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) Bf[256 * (f0 + u) + col_off] = to_f2(acc[u]);
            }
            // the next tile's image must have landed before the step's closing barrier; a later tile's DMA (issued above, eight
            // instructions per wave) may stay in flight
            if (V3_DIST >= 2 && b + V3_DIST < last) wait_vmcnt<8>(); else wait_vmcnt<0>();
        };
        for (unsigned s = 0; s <= nt; s += 2) {
            fstep(wa, wb, s);
            if (s + 1 > nt) break;
            fstep(wb, wa, s + 1);
        }
    } else {
        // ================================================================== BACK: DFT passes 1 + 2, freqdem, stores
        const FmK2 fk = {{opaque_v(9.999993443e-01f), opaque_v(-3.332985938e-01f), opaque_v(1.994656026e-01f), opaque_v(-1.390860826e-01f),
                          opaque_v(9.642146528e-02f), opaque_v(-5.591168255e-02f), opaque_v(2.186254039e-02f), opaque_v(-4.054457881e-03f)},
                         opaque_v(1e-37f), opaque_v(A.fm_ref), opaque_v(RA.pk.hp), opaque_v(RA.pk.pi)};
        const int f1 = lt >> 4, b1 = lt & 15;                                       // pass 1: frame, column digit
        const int k1 = lt >> 4, f2 = lt & 15;                                       // pass 2 / tail: channel digit, frame
        const unsigned x_a = (unsigned)f1 * 2048u + (unsigned)b1 * 8u;              // X[f1][16 a + b1]: (x_a ^ ((a >> 1) << 4)) + 128 a
        const unsigned zw_a = (unsigned)(f1 * 256 + ((((b1 >> 1) ^ (f1 & 7)) << 1) | (b1 & 1))) * 8u;   // Z[f1][k1][b1]: + 128 k1
        const unsigned z_a = (unsigned)(f2 * 256 + k1 * 16) * 8u + ((unsigned)(f2 & 7) << 4);           // pair i of Z[f2][k1][.]: z_a ^ (i << 4)
        const uint32_t voff = ((uint32_t)k1 * A.out_stride + A.out_t0 + (uint32_t)f2) * (FM ? 4u : 8u);  // + 16 k2 rows, + 16 b frames
        const size_t row16 = (size_t)16 * A.out_stride * (FM ? 4u : 8u);
        // FM: the even tile of a pair keeps its 16 results per thread and the odd tile stores both: whole 128-byte lines of the F32 rows
        float hold[16];
        auto bstep = [&](unsigned s_, const int par) {                              // tile s - 1 of the run; par = its position in the tile pair
            unsigned s = (unsigned)__builtin_amdgcn_readfirstlane((int)s_);
            asm volatile("" : "+s"(s));
            bar();                                      // A: X(s - 1) complete
            if (s == 0) { bar(); return; }              // the fill step: the front waves work on the first tile
            const unsigned b = first + s - 1;
            char *B = reinterpret_cast<char *>(L) + ((s - 1) % V3_NBUF) * (V3_BUF * 8u);
            // ---- DFT pass 1: thread (f1, b1)
            v2f vv[16];
#pragma unroll
            for (int a = 0; a < 16; a++) vv[a] = to_v(*reinterpret_cast<const float2 *>(B + (x_a ^ (unsigned)((a >> 1) << 4)) + 128 * a));
            fft16_v(vv);
#pragma unroll
            for (int i = 1; i < 16; i++) vv[i] = cmul_v(vv[i], to_v(tw_s[16 * XIDX(i) + b1]));
            // no barrier here: Z[f1] goes into frame f1's own 2 KiB of the buffer, which only the 16 lanes that have just read
            // X[f1] (same wave, program order) ever touched since barrier A
#pragma unroll
            for (int i = 0; i < 16; i++) *reinterpret_cast<float2 *>(B + zw_a + 128 * XIDX(i)) = to_f2(vv[i]);
            bar();                                      // B: Z complete
            // ---- DFT pass 2: thread (k1, f2) reads its 16 consecutive Z values as eight 16-byte pairs
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const v4f v = *reinterpret_cast<const v4f *>(B + (z_a ^ (unsigned)(i << 4)));
                vv[2 * i] = (v2f){v.x, v.y}; vv[2 * i + 1] = (v2f){v.z, v.w};
            }
            fft16_v(vv);                                // vv[i] = Y[k1 + 16 XIDX(i)] of frame f2
            // ---- tail
            char *obase = reinterpret_cast<char *>(A.out) + (size_t)b * RA.tile_step;
            if (FM) {
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
                    float (&mq)[4] = (par == 0) ? *reinterpret_cast<float (*)[4]>(&hold[i]) : mq_;
                    fm_quad(rp, rr, fk, mq);
                    if (par == 0 && b + 1 < last) continue;   // (uniform) the odd tile of the pair stores these
#pragma unroll
                    for (int u = 0; u < 4; u++) {                       // stores go out between the quads
                        const char *rowp = obase + (size_t)XIDX(i + u) * row16;
                        if (V3_ABLATE & 2) asm volatile("" :: "v"(mq[u]), "s"(rowp));
                        else {
                            if (par == 1) asm volatile(V3_SNOP "global_store_dword %0, %1, %2 offset:-64" :: "v"(voff), "v"(hold[i + u]), "s"(rowp) : "memory");
                            asm volatile(V3_SNOP "global_store_dword %0, %1, %2" :: "v"(voff), "v"(mq[u]), "s"(rowp) : "memory");
                        }
                    }
                }
                if (f2 == 15) {
#pragma unroll
                    for (int i = 0; i < 16; i += 2) *reinterpret_cast<v4f *>(ST + k1 * 16 + i) = (v4f){vv[i].x, vv[i].y, vv[i + 1].x, vv[i + 1].y};
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const char *rowp = obase + (size_t)XIDX(i) * row16;
                    if (V3_ABLATE & 2) asm volatile("" :: "v"(vv[i]), "s"(rowp));
                    else asm volatile(V3_SNOP "global_store_dwordx2 %0, %1, %2\n\ts_nop 1" :: "v"(voff), "v"(vv[i]), "s"(rowp) : "memory");
                }
            }
        };
        for (unsigned s = 0; s <= nt; s += 2) {
            bstep(s, 1);
            if (s + 1 > nt) break;
            bstep(s + 1, 0);
        }
    }

    // ------------------------------------------------------------------ state after the run
    bar();                                              // stash of the last tile visible to every back wave
    if (last == A.nb) {
        if (!front && FM) A.rp_out[lt] = ST[st_idx];
        if (tid == 0) A.vend_out[0] = c;
    }
}

}  // namespace

int run256_v3_launch(const void *run_args, bool fm, unsigned nruns, hipStream_t s)
{
    V3Args VA;
    VA.r = *static_cast<const RunArgs *>(run_args);
    if (fm) hipLaunchKernelGGL((k_run256v3<true>), dim3(nruns), dim3(512), 0, s, VA);
    else hipLaunchKernelGGL((k_run256v3<false>), dim3(nruns), dim3(512), 0, s, VA);
    return 0;
}

}  // namespace csdr
