// Single-pass DC blocker + NCO mix for the generic (any M) path: one workgroup per 4096-sample
// tile, zero-state scan in LDS (shared with the fused kernels), tile carries by decoupled
// look-back over the previous 10 published tile aggregates (ticket order => deadlock-free; see
// kernels_fused.hip), one coalesced 16 B/lane read and write per sample.
// Replaces iirfilt_crcf_execute_block + nco_crcf_mix_block_down (Liquid.chs:575-589, 846-847)
// for whole chunks; the 3-kernel scan in kernels_generic.hip remains for the standalone Pipes.
#include "fused_common.h"
#include <cstring>

namespace csdr {

namespace {

struct DcTileArgs {
    const float2 *x; float2 *y;
    uint32_t n;                 // samples
    uint32_t nb;                // tiles
    const float2 *vend_in; float2 *vend_out;
    unsigned *ticket; u64 *agg; unsigned *status;
    uint32_t epoch;
    NcoParams nco; const float2 *nco_tab; int do_mix;
    uint32_t pick; float2 *ypick;   // pick > 0: store only the samples whose stream index is a multiple of `pick`, compacted (branch 0 of every frame)
    float alpha, beta;
    float wtile[LOOKBACK + 2], b16[16], b256[17], bj[16];
};

__global__ __launch_bounds__(256) void k_dc_tile(DcTileArgs D)
{
    __shared__ __attribute__((aligned(16))) float2 R[4096];
    __shared__ float2 Tt[16];
    __shared__ float2 carry_s;
    __shared__ unsigned tile_s;
    const int tid = threadIdx.x;
    if (tid == 0) tile_s = atomicAdd(D.ticket, 1u);
    __syncthreads();
    const unsigned b = tile_s;
    if (b >= D.nb) return;
    const uint32_t n0 = b * 4096u, nleft = D.n - n0;
    const int valid_runs = (int)min(256u, (nleft + 15) / 16);
    // TileArgs view for the shared helpers (alpha/beta/b16/b256 only)
    TileArgs A{};
    A.alpha = D.alpha; A.beta = D.beta;
#pragma unroll
    for (int i = 0; i < 16; i++) { A.b16[i] = D.b16[i]; A.bj[i] = D.bj[i]; }
#pragma unroll
    for (int i = 0; i < 17; i++) A.b256[i] = D.b256[i];

    float4 raw[8];
    if (nleft >= 4096u) tile_load(reinterpret_cast<const float4 *>(D.x) + (size_t)b * 2048, 256, raw, tid);
    else {
        // ragged last tile: per-sample guard (n need not be a multiple of 16)
        const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int slot = 64 * (it * 4 + wave) + lane, q = slot >> 3;
            const int i = (slot & 7) ^ ((q >> 1) & 7);
            const uint32_t s0 = 16u * q + 2u * i;
            const float2 a = s0 < nleft ? D.x[n0 + s0] : make_float2(0.f, 0.f);
            const float2 c = s0 + 1 < nleft ? D.x[n0 + s0 + 1] : make_float2(0.f, 0.f);
            raw[it] = make_float4(a.x, a.y, c.x, c.y);
        }
    }
    (void)valid_runs;
    const float2 e = stage_and_scan(raw, R, nullptr, Tt, A, tid);
    float2 vb, ve;
    frame_carries(Tt, A, tid, vb, ve);

    if (tid < 64) {
        if (tid == 0) {
            __hip_atomic_store(&D.agg[2 * (size_t)b], ((u64)D.epoch << 32) | __float_as_uint(ve.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&D.agg[2 * (size_t)b + 1], ((u64)D.epoch << 32) | __float_as_uint(ve.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const int k = tid;
        float2 cb = make_float2(0.f, 0.f);
        const bool need = (k >= 1 && k <= LOOKBACK && (int)b - k >= 0);
        u64 g0 = 0, g1 = 0;
        unsigned spins = 0;
        bool ok = !need;
        while (true) {
            if (need && !ok) {
                g0 = __hip_atomic_load(&D.agg[2 * (size_t)(b - k)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                g1 = __hip_atomic_load(&D.agg[2 * (size_t)(b - k) + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = (unsigned)(g0 >> 32) == D.epoch && (unsigned)(g1 >> 32) == D.epoch;
            }
            if (__all(ok)) break;
            if (++spins > SPIN_LIMIT) { if (k == 0) atomicOr(D.status, 4u); break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (need) cb = make_float2(__uint_as_float((unsigned)g0) * D.wtile[k - 1], __uint_as_float((unsigned)g1) * D.wtile[k - 1]);
        if (k == 0 && b <= LOOKBACK) { const float2 v = D.vend_in[0]; cb = make_float2(v.x * D.wtile[b], v.y * D.wtile[b]); }
        cb = cadd(cb, dpp2<0x111>(cb)); cb = cadd(cb, dpp2<0x112>(cb));
        cb = cadd(cb, dpp2<0x114>(cb)); cb = cadd(cb, dpp2<0x118>(cb));
        if (k == 15) carry_s = cb;
    }
    __syncthreads();
    const float2 c = carry_s;
    const float br = D.b16[tid & 15], bf = D.b256[tid >> 4];
    const float2 Pq = cfma(cfma(c, bf, vb), br, e);                  // v before my run
    if (b == D.nb - 1) {
        // state after the last sample: v before the first padded sample.  Its run owner rebuilds it.
        const uint32_t last_run = (nleft - 1) / 16, within = (nleft - 1) % 16;
        if ((uint32_t)tid == last_run) {
            float4 *R4 = reinterpret_cast<float4 *>(R);
            const int sw = (tid >> 1) & 7;
            float2 v = Pq, s = make_float2(0.f, 0.f);
            for (uint32_t i = 0; i <= within; i++) {
                const float4 zz = R4[8 * tid + ((int)(i >> 1) ^ sw)];
                const float2 z = (i & 1) ? make_float2(zz.z, zz.w) : make_float2(zz.x, zz.y);
                const float2 x = make_float2(z.x + D.alpha * s.x, z.y + D.alpha * s.y);   // undo z = x - alpha*s_prev
                s = cfma(s, D.beta, x);
                v = cfma(v, D.beta, x);
            }
            D.vend_out[0] = v;
        }
    }
    // ---- finish: y = z - alpha*beta^i*P, NCO mix, coalesced store in the tile-load pattern ----
    {
        float4 *R4 = reinterpret_cast<float4 *>(R);
        const int q = tid, sw = (q >> 1) & 7;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float4 z = R4[8 * q + (i ^ sw)];
            const float k0 = -D.alpha * D.bj[2 * i], k1 = -D.alpha * D.bj[2 * i + 1];
            float2 y0 = make_float2(fmaf(Pq.x, k0, z.x), fmaf(Pq.y, k0, z.y));
            float2 y1 = make_float2(fmaf(Pq.x, k1, z.z), fmaf(Pq.y, k1, z.w));
            R4[8 * q + (i ^ sw)] = make_float4(y0.x, y0.y, y1.x, y1.y);
        }
    }
    __syncthreads();
    const uint32_t tl = D.nco.tab_len;
    if (nleft >= 4096u && D.do_mix && tl && (tl & (tl - 1)) == 0) {
        // whole tile, tabulated NCO with a power-of-two period (the channelizer's pre-mix): no range tests, the
        // period by a mask, and all sixteen table reads issued before the first one is needed.  The general loop below
        // has a lane-varying `continue` in front of its loads (each of them then gets waited for on its own) and two
        // 32-bit remainders per piece.
        const float4 *R4 = reinterpret_cast<const float4 *>(R);
        const int wave = tid >> 6, lane = tid & 63;
        const uint32_t mask = tl - 1, base = D.nco.tab_pos + n0;
        float4 yv[8];
        float2 ca[8], cb[8];
        uint32_t dst[8];
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int slot = 64 * (it * 4 + wave) + lane, q = slot >> 3;
            const int i = (slot & 7) ^ ((q >> 1) & 7);
            const uint32_t p0 = (base + 16u * q + 2u * i) & mask;
            yv[it] = R4[slot];
            ca[it] = D.nco_tab[p0];
            cb[it] = D.nco_tab[(p0 + 1) & mask];
            dst[it] = 8u * q + i;
        }
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const float sa = D.nco.up ? ca[it].y : -ca[it].y, sb = D.nco.up ? cb[it].y : -cb[it].y;
            const float4 v = yv[it];
            const float4 o = make_float4(v.x * ca[it].x - v.y * sa, v.x * sa + v.y * ca[it].x, v.z * cb[it].x - v.w * sb, v.z * sb + v.w * cb[it].x);
            if (D.pick) {
                const uint32_t s0 = n0 + 2u * dst[it];                       // stream index of the pair's first sample
                if (s0 % D.pick == 0) D.ypick[s0 / D.pick] = make_float2(o.x, o.y);
                if ((s0 + 1) % D.pick == 0) D.ypick[(s0 + 1) / D.pick] = make_float2(o.z, o.w);
            } else reinterpret_cast<float4 *>(D.y)[(size_t)b * 2048 + dst[it]] = o;
        }
    } else {
        const float4 *R4 = reinterpret_cast<const float4 *>(R);
        const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int slot = 64 * (it * 4 + wave) + lane, q = slot >> 3;
            const int i = (slot & 7) ^ ((q >> 1) & 7);
            const uint32_t s0 = 16u * q + 2u * i;
            if (s0 >= nleft) continue;
            float4 yv = R4[slot];
            float2 y0 = make_float2(yv.x, yv.y), y1 = make_float2(yv.z, yv.w);
            if (D.do_mix) {
                const uint32_t idx = n0 + s0;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    float c_, s_;
                    if (D.nco.tab_len) { const float2 cs = D.nco_tab[(D.nco.tab_pos + idx + h) % D.nco.tab_len]; c_ = cs.x; s_ = cs.y; }
                    else {
                        const uint32_t theta = D.nco.theta0 + (idx + h) * D.nco.d_theta;
                        const float ph = (float)(6.283185307179586 * (double)(float)theta / 4294967296.0);
                        sincosf(ph, &s_, &c_);
                    }
                    if (!D.nco.up) s_ = -s_;
                    float2 &yy = h ? y1 : y0;
                    yy = make_float2(yy.x * c_ - yy.y * s_, yy.x * s_ + yy.y * c_);
                }
            }
            if (D.pick) {
                if ((n0 + s0) % D.pick == 0) D.ypick[(n0 + s0) / D.pick] = y0;
                if (s0 + 1 < nleft && (n0 + s0 + 1) % D.pick == 0) D.ypick[(n0 + s0 + 1) / D.pick] = y1;
            } else if (s0 + 1 < nleft) reinterpret_cast<float4 *>(D.y)[(size_t)b * 2048 + 8 * q + i] = make_float4(y0.x, y0.y, y1.x, y1.y);
            else D.y[n0 + s0] = y0;
        }
    }
}

// (Round 2 had a first version of this job with decoupled look-back, k_dc_pick_tile: 117 us against 98-107; deleted in round 5 --
// no route reached it without a diagnostics knob.  `git show f9835f7:composable_sdr_amd/csrc/kernels_dc_tile.hip` has it.)
// The same job without any inter-workgroup hand-off (round 2): the DC state in front of a frame is a decayed sum over
// everything before it, so k_dc_fold leaves one zero-state aggregate (and the first sample) per 4096-sample tile -- one
// wave per tile, four 8 KiB quarters with the next one in flight, no LDS, no barrier, no ticket: a plain streaming read --
// and k_mixid_finish forms the state in front of a frame from the ten aggregates before it (beta^(4096 * 10) = 1.3e-9 of
// anything older: the cut the look-back kernels make) and runs the branch-0 FIR.
__global__ __launch_bounds__(256) void k_dc_fold(const float4 *__restrict__ x, float2 *__restrict__ agg, float2 *__restrict__ first, uint32_t nb, float l2beta)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wid = (blockIdx.x * 256u + threadIdx.x) >> 6, nw = (gridDim.x * 256u) >> 6;
    float w0[8], w1[8];
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const int n = 2 * (64 * it + lane);             // first of my two samples of piece `it` inside a 1024-sample quarter
        w0[it] = exp2f((float)(1023 - n) * l2beta); w1[it] = exp2f((float)(1022 - n) * l2beta);
    }
    const float bq = exp2f(1024.0f * l2beta);           // beta^1024: one quarter further
    float4 cur[8], nxt[8];
    uint32_t b = wid;
    if (b < nb) {
#pragma unroll
        for (int it = 0; it < 8; it++) cur[it] = x[(size_t)b * 2048 + 64 * it + lane];
    }
    for (; b < nb; b += nw) {
        float2 a = make_float2(0.f, 0.f);
        float2 x0 = make_float2(cur[0].x, cur[0].y);    // lane 0: the tile's first sample (the picked one)
#pragma unroll 1
        for (int qd = 0; qd < 4; qd++) {
            // next quarter (of this tile, or the first one of my next tile) in flight
            const uint32_t bn = qd < 3 ? b : b + nw;
            const int qn = qd < 3 ? qd + 1 : 0;
            if (bn < nb) {
#pragma unroll
                for (int it = 0; it < 8; it++) nxt[it] = x[(size_t)bn * 2048 + 512 * qn + 64 * it + lane];
            }
            float2 p = make_float2(0.f, 0.f);
#pragma unroll
            for (int it = 0; it < 8; it++) {
                p = cfma(make_float2(cur[it].x, cur[it].y), w0[it], p);
                p = cfma(make_float2(cur[it].z, cur[it].w), w1[it], p);
            }
            a = cfma(a, bq, p);                         // per lane: the lanes are summed once per tile
#pragma unroll
            for (int it = 0; it < 8; it++) cur[it] = nxt[it];
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { a.x += __shfl_xor(a.x, d); a.y += __shfl_xor(a.y, d); }
        if (lane == 0) { agg[b] = a; first[b] = x0; }
    }
}

// The mix identity in two launches: k_dc_fold, then this -- the picked samples of a workgroup's 256 frames (+ p - 1 in front:
// from the tile aggregates, or from the previous call's history) go to LDS, the branch-0 FIR runs out of it
// (out[t] = M sum_n h[(M-1) + n M] u0[t - n]), the call's last p - 1 picked samples and the DC state are left for the next call.
__device__ __forceinline__ float2 dc_state_before(const DcTileArgs &D, const float2 *__restrict__ agg, uint32_t b)
{
    float2 v = make_float2(0.f, 0.f);
#pragma unroll
    for (int k = LOOKBACK; k >= 1; k--) {
        if (b >= (uint32_t)k) { const float2 a = agg[b - k]; v = make_float2(fmaf(a.x, D.wtile[k - 1], v.x), fmaf(a.y, D.wtile[k - 1], v.y)); }
    }
    if (b <= LOOKBACK) { const float2 vi = D.vend_in[0]; v = make_float2(fmaf(vi.x, D.wtile[b], v.x), fmaf(vi.y, D.wtile[b], v.y)); }
    return v;
}
__global__ __launch_bounds__(256) void k_mixid_finish(DcTileArgs D, const float2 *__restrict__ agg, const float2 *__restrict__ first,
                                                     const float *__restrict__ taps, const float2 *__restrict__ hist_in, float2 *__restrict__ hist_out,
                                                     float2 *__restrict__ out, uint32_t M, uint32_t p, uint32_t nf)
{
    __shared__ float2 u[256 + 32];                      // p - 1 <= 32 samples of history in front
    const uint32_t tid = threadIdx.x, f0 = blockIdx.x * 256u, tpf = D.pick / 4096u, H = p - 1;
    auto picked = [&](int64_t f) -> float2 {
        if (f < 0) return hist_in[(int64_t)H + f];
        const uint32_t b = (uint32_t)f * tpf, n0 = b * 4096u;
        const float2 v = dc_state_before(D, agg, b), x0 = first[b];
        float2 y = make_float2(fmaf(-D.alpha, v.x, x0.x), fmaf(-D.alpha, v.y, x0.y));
        if (D.do_mix) {
            float c_, s_;
            if (D.nco.tab_len) { const float2 cs = D.nco_tab[(D.nco.tab_pos + n0) % D.nco.tab_len]; c_ = cs.x; s_ = cs.y; }
            else {
                const uint32_t theta = D.nco.theta0 + n0 * D.nco.d_theta;
                sincosf((float)(6.283185307179586 * (double)(float)theta / 4294967296.0), &s_, &c_);
            }
            if (!D.nco.up) s_ = -s_;
            y = make_float2(y.x * c_ - y.y * s_, y.x * s_ + y.y * c_);
        }
        return y;
    };
    u[H + tid] = (f0 + tid < nf) ? picked((int64_t)(f0 + tid)) : make_float2(0.f, 0.f);
    if (tid < H) u[tid] = picked((int64_t)f0 - (int64_t)H + (int64_t)tid);
    __syncthreads();
    const uint32_t t = f0 + tid;
    if (t < nf) {
        float2 acc = make_float2(0.f, 0.f);
        for (uint32_t n = p; n-- > 0;) {                 // oldest tap first, like the bank's dot product
            const float h = taps[(M - 1) + (size_t)n * M];
            const float2 w = u[H + tid - n];
            acc.x = fmaf(w.x, h, acc.x); acc.y = fmaf(w.y, h, acc.y);
        }
        out[t] = make_float2(acc.x * (float)M, acc.y * (float)M);
    }
    if (f0 + 256u >= nf) {                               // the workgroup with the call's last frame
        if (tid < H) {
            const int64_t f = (int64_t)nf - (int64_t)H + (int64_t)tid;      // >= f0 - H when nf >= ... ; earlier frames come from the old history
            hist_out[tid] = (f >= (int64_t)f0 - (int64_t)H) ? u[(uint32_t)(f - ((int64_t)f0 - (int64_t)H))] : picked(f);
        }
        if (tid == 0) {
            const uint32_t b = D.nb - 1;
            const float2 v = dc_state_before(D, agg, b), a = agg[b];
            D.vend_out[0] = make_float2(fmaf(v.x, D.wtile[1], a.x), fmaf(v.y, D.wtile[1], a.y));
        }
    }
}

// ---- the mix identity of an INTERLEAVED CHANNEL SHARD (round 6; BASELINE configs[4] per rank: 4096 channels, DeNo --mix, rank g of G) ----
// sum_{m < M/G} Y_t[g + G m] = sum_n X_t[n] W_M^(n g) sum_m W_(M/G)^(n m) = (M / G) sum_{n2 < G} X_t[(M / G) n2] W_G^(n2 g):
// of the M polyphase branches only the G branches n = (M / G) n2 survive the shard's channel sum (all ranks together: M X_t[0], the
// whole-band identity above).  So a rank needs the DC blocker on the whole stream, G picked samples per frame (DC state in front of
// each: k_dc_fold8 leaves one zero-state aggregate and the first sample per 512-sample EIGHTH of a tile), their pre-mix, G 2m-tap
// FIRs and a phasor sum -- no bank, no DFT, no channel sum; 8 B read per input sample like the whole-band identity.  The any-M route
// with a pruned DFT that these configurations took before runs 678 us per 67.1 M samples at M = 4096, G = 8.
__global__ __launch_bounds__(256) void k_dc_fold8(const float4 *__restrict__ x, float2 *__restrict__ agg, float2 *__restrict__ agg8, float2 *__restrict__ first8,
                                                  uint32_t nb, float l2beta)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wid = (blockIdx.x * 256u + threadIdx.x) >> 6, nw = (gridDim.x * 256u) >> 6;
    float w0[4], w1[4];
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const int n = 2 * (64 * it + lane);             // first of my two samples of piece `it` inside a 512-sample eighth
        w0[it] = exp2f((float)(511 - n) * l2beta); w1[it] = exp2f((float)(510 - n) * l2beta);
    }
    const float be = exp2f(512.0f * l2beta);            // beta^512: one eighth further
    float4 cur[4], nxt[4];
    uint32_t b = wid;
    if (b < nb) {
#pragma unroll
        for (int it = 0; it < 4; it++) cur[it] = x[(size_t)b * 2048 + 64 * it + lane];
    }
    for (; b < nb; b += nw) {
        float2 pe[8], x0[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            // next eighth (of this tile, or the first one of my next tile) in flight
            const uint32_t bn = e < 7 ? b : b + nw;
            const int en = e < 7 ? e + 1 : 0;
            if (bn < nb) {
#pragma unroll
                for (int it = 0; it < 4; it++) nxt[it] = x[(size_t)bn * 2048 + 256 * en + 64 * it + lane];
            }
            float2 pp = make_float2(0.f, 0.f);
#pragma unroll
            for (int it = 0; it < 4; it++) {
                pp = cfma(make_float2(cur[it].x, cur[it].y), w0[it], pp);
                pp = cfma(make_float2(cur[it].z, cur[it].w), w1[it], pp);
            }
            pe[e] = pp;
            x0[e] = make_float2(cur[0].x, cur[0].y);   // lane 0: the eighth's first sample
#pragma unroll
            for (int it = 0; it < 4; it++) cur[it] = nxt[it];
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
#pragma unroll
            for (int e = 0; e < 8; e++) { pe[e].x += __shfl_xor(pe[e].x, d); pe[e].y += __shfl_xor(pe[e].y, d); }
        }
        if (lane == 0) {
            float2 a = make_float2(0.f, 0.f);
#pragma unroll
            for (int e = 0; e < 8; e++) { a = cfma(a, be, pe[e]); agg8[(size_t)b * 8 + e] = pe[e]; first8[(size_t)b * 8 + e] = x0[e]; }
            agg[b] = a;
        }
    }
}

struct ShardMixArgs { uint32_t G, Mg; float2 ph[8]; };      // W_G^(n2 g), n2 < G <= 8; Mg = M / G (a multiple of 512)

__global__ __launch_bounds__(256) void k_mixid_shard_finish(DcTileArgs D, ShardMixArgs S, const float2 *__restrict__ agg, const float2 *__restrict__ agg8,
                                                           const float2 *__restrict__ first8, const float *__restrict__ taps,
                                                           const float2 *__restrict__ hist_in, float2 *__restrict__ hist_out,
                                                           float2 *__restrict__ out, uint32_t M, uint32_t p, uint32_t nf)
{
    // a workgroup = 256 / G frames x the G surviving branches: thread (n2, fl) picks and filters ONE (branch, frame) pair -- with 256 frames
    // per workgroup and a loop over the branches (first version) a 16 384-frame call was 64 workgroups on 256 CUs, each thread chasing
    // G x (ten tile aggregates + up to seven eighths) through the L2 one after the other: 31 us against the whole-band finish's 8
    __shared__ float2 u[8][128 + 32];                   // per surviving branch: p - 1 <= 32 samples of history in front
    __shared__ float2 part[8][128];
    const uint32_t tid = threadIdx.x, FW = 256u / S.G, n2 = tid / FW, fl = tid % FW, f0 = blockIdx.x * FW, tpf = M / 4096u, H = p - 1;
    const float be = D.b256[2];                         // beta^512
    auto picked = [&](int64_t f, uint32_t nb2) -> float2 {
        if (f < 0) return hist_in[(size_t)nb2 * H + (size_t)((int64_t)H + f)];
        const uint32_t o = S.Mg * nb2, b = (uint32_t)f * tpf + o / 4096u, e = (o % 4096u) / 512u, n0 = b * 4096u + 512u * e;
        float2 v = dc_state_before(D, agg, b);          // in front of the tile; then through the eighths in front of mine
        for (uint32_t k = 0; k < e; k++) { const float2 a = agg8[(size_t)b * 8 + k]; v = make_float2(fmaf(v.x, be, a.x), fmaf(v.y, be, a.y)); }
        const float2 x0 = first8[(size_t)b * 8 + e];
        float2 y = make_float2(fmaf(-D.alpha, v.x, x0.x), fmaf(-D.alpha, v.y, x0.y));
        if (D.do_mix) {
            float c_, s_;
            if (D.nco.tab_len) { const float2 cs = D.nco_tab[(D.nco.tab_pos + n0) % D.nco.tab_len]; c_ = cs.x; s_ = cs.y; }
            else {
                const uint32_t theta = D.nco.theta0 + n0 * D.nco.d_theta;
                sincosf((float)(6.283185307179586 * (double)(float)theta / 4294967296.0), &s_, &c_);
            }
            if (!D.nco.up) s_ = -s_;
            y = make_float2(y.x * c_ - y.y * s_, y.x * s_ + y.y * c_);
        }
        const float2 w = S.ph[nb2];                     // the branch's share of the shard's shift: W_G^(n2 g)
        return make_float2(y.x * w.x - y.y * w.y, y.x * w.y + y.y * w.x);
    };
    u[n2][H + fl] = (f0 + fl < nf) ? picked((int64_t)(f0 + fl), n2) : make_float2(0.f, 0.f);
    if (fl < H) u[n2][fl] = picked((int64_t)f0 - (int64_t)H + (int64_t)fl, n2);
    __syncthreads();
    const uint32_t t = f0 + fl;
    {
        const uint32_t j = S.Mg * n2;                   // my surviving branch: taps h[(M - 1 - j) + n M]
        float2 a2 = make_float2(0.f, 0.f);
        for (uint32_t n = p; n-- > 0;) {                 // oldest tap first, like the bank's dot product
            const float h = taps[(M - 1 - j) + (size_t)n * M];
            const float2 w = u[n2][H + fl - n];
            a2.x = fmaf(w.x, h, a2.x); a2.y = fmaf(w.y, h, a2.y);
        }
        part[n2][fl] = a2;
    }
    __syncthreads();
    if (n2 == 0 && t < nf) {
        float2 acc = part[0][fl];
        for (uint32_t k = 1; k < S.G; k++) { acc.x += part[k][fl].x; acc.y += part[k][fl].y; }     // ascending branch order
        out[t] = make_float2(acc.x * (float)S.Mg, acc.y * (float)S.Mg);
    }
    if (f0 + FW >= nf) {                                 // the workgroup with the call's last frame
        if (fl < H) {
            const int64_t f = (int64_t)nf - (int64_t)H + (int64_t)fl;
            hist_out[(size_t)n2 * H + fl] = (f >= (int64_t)f0 - (int64_t)H) ? u[n2][(uint32_t)(f - ((int64_t)f0 - (int64_t)H))] : picked(f, n2);
        }
        if (tid == 0) {
            const uint32_t b = D.nb - 1;
            const float2 v = dc_state_before(D, agg, b), a = agg[b];
            D.vend_out[0] = make_float2(fmaf(v.x, D.wtile[1], a.x), fmaf(v.y, D.wtile[1], a.y));
        }
    }
}

// DeNo --mix over ALL channels of an M-channel bank (Trans.hs:119-122 after Liquid.chs:843): sum_k Y_t[k] = M X_t[0], because
// sum_k W_M^{jk} = M delta[j]: only polyphase branch 0 of every frame survives the channel sum.  u0[13 + t] = DC-blocked,
// pre-mixed sample t*M of the stream (13 samples of history in front); out[t] = M sum_n h[(M-1) + n M] u0[13 + t - n].
__global__ __launch_bounds__(256) void k_branch0_fir(const float2 *__restrict__ u0, const float *__restrict__ taps, float2 *__restrict__ out,
                                                    float2 *__restrict__ hist_out, uint32_t M, uint32_t p, uint32_t nf)
{
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t < nf) {
        float2 acc = make_float2(0.f, 0.f);
        for (uint32_t n = p; n-- > 0;) {                 // oldest tap first, like the bank's dot product
            const float h = taps[(M - 1) + (size_t)n * M];
            const float2 v = u0[(p - 1) + t - n];
            acc.x = fmaf(v.x, h, acc.x); acc.y = fmaf(v.y, h, acc.y);
        }
        out[t] = make_float2(acc.x * (float)M, acc.y * (float)M);
    }
    if (t < p - 1) hist_out[t] = u0[nf + t];              // the last p - 1 branch-0 samples: history of the next call
}

}  // namespace

int launch_branch0_fir(const float2 *u0, const float *taps, float2 *out, float2 *hist_out, uint32_t M, uint32_t p, uint32_t nf, hipStream_t s)
{
    if (!nf) return 0;
    hipLaunchKernelGGL(k_branch0_fir, dim3((nf + 255) / 256), dim3(256), 0, s, u0, taps, out, hist_out, M, p, nf);
    CSDR_HIP(hipGetLastError());
    return 0;
}

struct DcTilePlan {
    uint32_t max_nb = 0, epoch = 0;
    unsigned *d_ticket = nullptr, *d_status = nullptr;
    u64 *d_agg = nullptr;
    float2 *d_part = nullptr;            // k_dc_fold: one aggregate and the first sample per tile
    float2 *d_part8 = nullptr;           // k_dc_fold8 (shard mix identity; allocated on first use): 8 aggregates + 8 first samples per tile
    uint32_t cus = 256;
    float2 *d_vend[2] = {nullptr, nullptr};
    int cur = 0;
    DcTileArgs proto;
};

int dctile_status(DcTilePlan *p, unsigned *status)
{
    CSDR_HIP(hipMemcpy(status, p->d_status, sizeof(unsigned), hipMemcpyDeviceToHost));
    if (*status) CSDR_HIP(hipMemset(p->d_status, 0, sizeof(unsigned)));
    return 0;
}

void dctile_destroy(DcTilePlan *p)
{
    if (!p) return;
    void *ptrs[] = {p->d_ticket, p->d_status, p->d_agg, p->d_part, p->d_part8, p->d_vend[0], p->d_vend[1]};
    for (void *q : ptrs) if (q) (void)hipFree(q);
    delete p;
}

int dctile_create(const DcParams &dc, uint64_t max_samples, DcTilePlan **out)
{
    DcTilePlan *p = new DcTilePlan();
    p->max_nb = (uint32_t)((max_samples + 4095) / 4096);
    auto fail = [&](int r) { dctile_destroy(p); return r; };
#define ALLOC(ptr, bytes) do { hipError_t e = hipMalloc((void **)&(ptr), (bytes) ? (bytes) : 1); if (e != hipSuccess) return fail(hip_fail(e, "hipMalloc", __FILE__, __LINE__)); } while (0)
    ALLOC(p->d_ticket, sizeof(unsigned)); ALLOC(p->d_status, sizeof(unsigned));
    ALLOC(p->d_agg, sizeof(u64) * 2 * p->max_nb);
    ALLOC(p->d_part, sizeof(float2) * 2 * (size_t)p->max_nb);      // [nb] tile aggregates | [nb] first samples of the tiles
    { int dev = 0, cus = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev); p->cus = (uint32_t)cus; }
    ALLOC(p->d_vend[0], sizeof(float2)); ALLOC(p->d_vend[1], sizeof(float2));
#undef ALLOC
    CSDR_HIP(hipMemset(p->d_status, 0, sizeof(unsigned)));
    CSDR_HIP(hipMemset(p->d_agg, 0, sizeof(u64) * 2 * p->max_nb));
    CSDR_HIP(hipMemset(p->d_vend[0], 0, sizeof(float2)));
    CSDR_HIP(hipMemset(p->d_vend[1], 0, sizeof(float2)));
    DcTileArgs &D = p->proto;
    D = DcTileArgs{};
    const double beta = (double)dc.beta;
    D.alpha = (float)(1.0 - beta); D.beta = (float)beta;
    for (int k = 0; k < LOOKBACK + 2; k++) D.wtile[k] = (float)std::pow(beta, 4096.0 * k);
    for (int k = 0; k < 16; k++) D.b16[k] = (float)std::pow(beta, 16.0 * k);
    for (int k = 0; k < 17; k++) D.b256[k] = (float)std::pow(beta, 256.0 * k);
    for (int k = 0; k < 16; k++) D.bj[k] = (float)std::pow(beta, (double)k);
    D.ticket = p->d_ticket; D.agg = p->d_agg; D.status = p->d_status;
    *out = p;
    return 0;
}

int dctile_reset(DcTilePlan *p, hipStream_t s)
{
    p->cur = 0;
    CSDR_HIP(hipMemsetAsync(p->d_vend[0], 0, sizeof(float2), s));
    CSDR_HIP(hipMemsetAsync(p->d_vend[1], 0, sizeof(float2), s));
    return 0;
}

// DeNo --mix over all M channels, M a multiple of 4096, n a multiple of M: out[n / M] = M x (branch-0 FIR of the DC-blocked,
// pre-mixed stream); hist_in / hist_out: the p - 1 picked samples before / after the call (different buffers)
bool dctile_mix_identity_supported(const DcTilePlan *p, uint32_t M, uint32_t n, uint32_t taps_p)
{
    return M && M % 4096u == 0 && n && n % M == 0 && p->proto.beta > 0.f && taps_p >= 1 && taps_p <= 33;
}

int dctile_mix_identity(DcTilePlan *p, const float2 *x, uint32_t n, const NcoParams &nco, const float2 *nco_tab, const float *taps,
                        uint32_t M, uint32_t taps_p, const float2 *hist_in, float2 *hist_out, float2 *out, hipStream_t s)
{
    DcTileArgs D = p->proto;
    D.x = x; D.y = nullptr; D.n = n; D.nb = n / 4096u;
    D.pick = M; D.ypick = nullptr;
    D.vend_in = p->d_vend[p->cur]; D.vend_out = p->d_vend[p->cur ^ 1];
    D.nco = nco; D.nco_tab = nco_tab; D.do_mix = 1;
    const float l2b = (float)std::log2((double)D.beta);
    uint32_t wgs = p->cus * 4u;                                      // four workgroups of four waves per CU, a tile per wave and round
    if (wgs > (D.nb + 3) / 4) wgs = (D.nb + 3) / 4;
    float2 *d_first = p->d_part + (size_t)p->max_nb;
    const uint32_t nf = n / M;
    hipLaunchKernelGGL(k_dc_fold, dim3(wgs), dim3(256), 0, s, reinterpret_cast<const float4 *>(x), p->d_part, d_first, D.nb, l2b);
    hipLaunchKernelGGL(k_mixid_finish, dim3((nf + 255) / 256), dim3(256), 0, s, D, (const float2 *)p->d_part, (const float2 *)d_first, taps,
                       hist_in, hist_out, out, M, taps_p, nf);
    CSDR_HIP(hipGetLastError());
    p->cur ^= 1;
    return 0;
}

// the same for the interleaved channel shard g of G: out[n / M] = (M / G) sum_{n2 < G} W_G^(n2 g) x (FIR of branch (M / G) n2); G <= 8,
// (M / G) % 512 == 0; hist_in / hist_out: [G][p - 1] picked samples before / after the call (different buffers)
bool dctile_mix_identity_shard_supported(const DcTilePlan *p, uint32_t M, uint32_t n, uint32_t taps_p, uint32_t G)
{
    return dctile_mix_identity_supported(p, M, n, taps_p) && (G == 2 || G == 4 || G == 8) && M % G == 0 && (M / G) % 512u == 0;
}

int dctile_mix_identity_shard(DcTilePlan *p, const float2 *x, uint32_t n, const NcoParams &nco, const float2 *nco_tab, const float *taps,
                              uint32_t M, uint32_t taps_p, uint32_t G, uint32_t g, const float2 *hist_in, float2 *hist_out, float2 *out, hipStream_t s)
{
    if (!p->d_part8) CSDR_HIP(hipMalloc((void **)&p->d_part8, sizeof(float2) * 16 * (size_t)p->max_nb));
    DcTileArgs D = p->proto;
    D.x = x; D.y = nullptr; D.n = n; D.nb = n / 4096u;
    D.pick = M; D.ypick = nullptr;
    D.vend_in = p->d_vend[p->cur]; D.vend_out = p->d_vend[p->cur ^ 1];
    D.nco = nco; D.nco_tab = nco_tab; D.do_mix = 1;
    ShardMixArgs S{};
    S.G = G; S.Mg = M / G;
    for (uint32_t n2 = 0; n2 < G; n2++) {
        const double a = -2.0 * 3.14159265358979323846 * (double)((n2 * g) % G) / (double)G;
        S.ph[n2] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    const float l2b = (float)std::log2((double)D.beta);
    uint32_t wgs = p->cus * 4u;                                      // four workgroups of four waves per CU, a tile per wave and round
    if (wgs > (D.nb + 3) / 4) wgs = (D.nb + 3) / 4;
    float2 *d_agg8 = p->d_part8, *d_first8 = p->d_part8 + (size_t)8 * p->max_nb;
    const uint32_t nf = n / M;
    hipLaunchKernelGGL(k_dc_fold8, dim3(wgs), dim3(256), 0, s, reinterpret_cast<const float4 *>(x), p->d_part, d_agg8, d_first8, D.nb, l2b);
    const uint32_t FW = 256u / G;                                    // frames per workgroup of the finish
    hipLaunchKernelGGL(k_mixid_shard_finish, dim3((nf + FW - 1) / FW), dim3(256), 0, s, D, S, (const float2 *)p->d_part, (const float2 *)d_agg8,
                       (const float2 *)d_first8, taps, hist_in, hist_out, out, M, taps_p, nf);
    CSDR_HIP(hipGetLastError());
    p->cur ^= 1;
    return 0;
}

int dctile_process(DcTilePlan *p, const float2 *x, float2 *y, uint32_t n, bool do_mix, const NcoParams &nco,
                   const float2 *nco_tab, hipStream_t s, uint32_t pick)
{
    if (!n) return 0;
    DcTileArgs D = p->proto;
    D.x = x; D.y = y; D.n = n; D.nb = (n + 4095) / 4096;
    D.pick = pick; D.ypick = y;                          // pick > 0: y receives n / pick samples (calls start on a multiple of pick)
    D.vend_in = p->d_vend[p->cur]; D.vend_out = p->d_vend[p->cur ^ 1];
    if (++p->epoch == 0) p->epoch = 1;
    D.epoch = p->epoch; D.nco = nco; D.nco_tab = nco_tab; D.do_mix = do_mix ? 1 : 0;
    CSDR_HIP(hipMemsetAsync(p->d_ticket, 0, sizeof(unsigned), s));
    hipLaunchKernelGGL(k_dc_tile, dim3(D.nb), dim3(256), 0, s, D);
    CSDR_HIP(hipGetLastError());
    p->cur ^= 1;
    return 0;
}

}  // namespace csdr
