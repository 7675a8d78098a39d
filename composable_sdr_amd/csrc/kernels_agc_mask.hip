// AGC + squelch behind an FM chain as a MASK pass (fused M = 256 chain, demod = FM, -a != 0).
//
// The reference composes fmDemodulator kf . automaticGainControl (SoapySDR.hs:249): y = g x with g > 0, muted to 0 unless the
// squelch is in SIGNALHI (Liquid.chs:703-704), then m = arg(conj(y') y) / (2 pi kf).  A positive gain drops out of the
// argument: wherever this sample and the one before it are un-muted, m is the freqdem of the channelizer output itself, which
// the run kernel has already written.  What the AGC contributes is WHICH samples are muted, and that only takes |Y|^2:
//     e = alpha |Y|^2 ;  y2' <- (1 - alpha) y2' + e g^2 ;  g <- g exp(-alpha/2 ln y2') ;  squelch on g      (agc_common.h)
// So the run kernel (k_run256v2<FM, EN> / k_tile256<FM>) leaves one 4-byte energy word per sample next to the FM samples
// (agc_energy_word: e, sign bit = both components of Y negative), and this file
//   k_agc_mask_spec : one lane per (channel, segment of L samples): W samples of warm-up from the call's start state, then
//                     the segment: two bits per sample into bit planes -- `flag` (this sample or the one before it is
//                     muted: the FM sample is not the plain freqdem) and `pi` (its value is ref*pi instead of 0: a sample
//                     next to a muted one is arg(conj(0) y) or arg(conj(y') 0), which is +pi exactly when both components
//                     of the un-muted one are negative, else +-0; fm_common.h: fm_sample_rn / atan2f_rn);
//   k_agc_mask_fix  : per channel: the boundary check and exact repair of the time-parallel tail (kernels_agc_tail.hip:
//                     recorded start state bitwise equal to the end state in front of it, else recompute), on (g, y2', S)
//                     (its repairs rewrite plane words);
//   k_agc_mask_apply: flagged samples are overwritten with 0 or ref*pi (one lane per four samples).
// Gains, y2' and squelch states are bit-identical to the sequential recurrence (same e, same code), hence so is every mute
// decision; un-muted FM samples are freqdem(Y) instead of freqdem(g Y): equal up to the rounding of an algebraically
// identical expression (<= 3e-7 rad).  2.0 GB per 67 M samples instead of 2.6 GB, and the pass is 4 B/sample instead of 8.
// MEASURED (round 2): 308 us k_run256v2<FM, EN> + 261 us k_agc_mask_spec + 44 us apply = 627 us per step against 601 us for the
// CF32 plane + k_agc_spec route: the step is not traffic-bound (a lone wave per SIMD runs ~40 instructions per sample of the
// recurrence whichever way the samples arrive).  Opt-in with CSDR_AGC_FM_MASK=1; the default stays the exact-FM route.
#include "../../include/csdr.h"
#include "csdr_internal.h"
#include "agc_common.h"
#include <cstdlib>

namespace csdr {

namespace {

struct MaskSeg { float g, y2; uint32_t S, pad; };                   // 16 B: state at a segment boundary

// the squelch state machine of kernels_agc_tail.hip (one integer S: 1 ENABLED, 2 RISE, 3 SIGNALHI, 4 FALL, 8 TIMEOUT,
// 8 + k SIGNALLO with k samples left)
constexpr uint32_t MS_TEX = (2u << 3) | (3u << 6) | (3u << 9) | (3u << 12) | (1u << 24);
constexpr uint32_t MS_TNO = (1u << 3) | (4u << 6) | (4u << 9) | (5u << 12) | (1u << 24);
__device__ __forceinline__ uint32_t ms_encode(int32_t mode, uint32_t timer) { return mode == 6 ? 8u : (mode == 5 ? 8u + timer : (uint32_t)mode); }
__device__ __forceinline__ void ms_decode(uint32_t S, uint32_t timeout, int32_t &mode, uint32_t &timer)
{
    mode = S > 8u ? 5 : (S == 8u ? 6 : (int32_t)S);
    timer = S > 8u ? S - 8u : timeout;
}
__device__ __forceinline__ uint32_t ms_next(uint32_t S, bool ex, const AgcParams &p)
{
    uint32_t t = __builtin_amdgcn_ubfe(ex ? MS_TEX : MS_TNO, 3u * S, 3u);
    t = (t == 5u) ? 8u + p.timeout : t;
    const uint32_t r9 = (ex && S >= 10u) ? 3u : S - 1u;
    return (S >= 9u) ? r9 : t;
}

struct MaskArgs {
    const float *E;             // [C][nf] energy words
    float *out;                 // [C][nf] FM samples (flagged ones are overwritten by k_agc_mask_fix)
    const AgcState *st_in;      // [C] state before the call
    const float2 *rp_in;        // [C] last channelizer frame of the previous call (its signs decide sample 0 after a mute)
    MaskSeg *seg_start, *seg_end;   // [C][nseg]
    uint32_t *flag, *pi;        // [C][nw] bit planes, bit i of word w = sample 32 w + i
    uint32_t C, nf, L, W, nseg, nw;
    AgcParams p;
    float ref;
};

// one sample: the state moves on; returns muted (S != SIGNALHI after the update, Liquid.chs:703-704)
__device__ __forceinline__ uint32_t mask_step(float ew, MaskSeg &q, const AgcParams &p)
{
    agc_gain_update(fabsf(ew), q.g, q.y2, p.alpha);
    q.S = ms_next(q.S, q.g < p.g_thr, p);
    return q.S != 3u;
}

// 32 samples (one word of the planes) from eight 16-byte pieces; mp / np: muted / both-negative of the sample in front
// (carried on).  The gain recurrence does not depend on the squelch state, and most samples leave the squelch where it is
// (SIGNALHI with the threshold exceeded, ENABLED without): a quad runs its four gain steps first and decides ONCE, for
// the wave, whether any lane's state moves; if not, the transition tables are skipped (kernels_agc_tail.hip: agc_quad).
template <bool GUARD>
__device__ __forceinline__ void mask_word(const float4 (&v)[8], uint32_t nv, MaskSeg &q, const AgcParams &p, uint32_t &mp, uint32_t &np,
                                          uint32_t &fw, uint32_t &pw)
{
    fw = 0u; pw = 0u;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const float e[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
        if (GUARD) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t t = 4u * i + k;
                if (t < nv) {
                    const uint32_t m = mask_step(e[k], q, p), n = __float_as_uint(e[k]) >> 31;
                    fw |= (m | mp) << t; pw |= ((m & (mp ^ 1u) & np) | ((m ^ 1u) & mp & n)) << t;
                    mp = m; np = n;
                }
            }
        } else {
            float gs[4], gmx = -INFINITY, gmn = INFINITY;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                agc_gain_update(fabsf(e[k]), q.g, q.y2, p.alpha);
                gs[k] = q.g;
                gmx = __builtin_amdgcn_fmed3f(q.g, gmx, INFINITY);
                gmn = __builtin_amdgcn_fmed3f(q.g, gmn, -INFINITY);
            }
            const uint32_t n3 = __float_as_uint(e[3]) >> 31;
            const bool all_ex = gmx < p.g_thr, none_ex = !(gmn < p.g_thr);
            const bool steady = (q.S == 3u && all_ex) || (q.S == 1u && none_ex);
            if (__builtin_amdgcn_ballot_w64(!steady) == 0ull) {
                // the state holds through the quad: all four muted (S = 1) or all four open (S = 3)
                const uint32_t m = q.S != 3u;
                const uint32_t n0 = __float_as_uint(e[0]) >> 31;
                // open quad: only its first sample can sit behind a muted one; muted quad: all flagged, the first one is
                // ref*pi when the sample in front was open with both components negative
                const uint32_t f4 = ((0u - m) & 15u) | mp, p4 = (m & (mp ^ 1u) & np) | ((m ^ 1u) & mp & n0);     // (no selects: bits)
                fw |= f4 << (4 * i); pw |= p4 << (4 * i);
                mp = m; np = n3;
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t t = 4u * i + k;
                    q.S = ms_next(q.S, gs[k] < p.g_thr, p);
                    const uint32_t m = q.S != 3u, n = __float_as_uint(e[k]) >> 31;
                    fw |= (m | mp) << t; pw |= ((m & (mp ^ 1u) & np) | ((m ^ 1u) & mp & n)) << t;
                    mp = m; np = n;
                }
            }
        }
    }
}

__global__ __launch_bounds__(64) void k_agc_mask_spec(MaskArgs A, uint32_t groups)
{
    const uint32_t lane = threadIdx.x, c = blockIdx.x / groups, sg = (blockIdx.x % groups) * 64u + lane;
    if (sg >= A.nseg) return;
    const size_t row = (size_t)c * A.nf;
    const float *Er = A.E + row;
    const uint32_t t_seg = sg * A.L, t_end = min(A.nf, t_seg + A.L);
    uint32_t t = t_seg > A.W ? t_seg - A.W : 0u;                    // segments that begin <= W samples in start at 0 from the true state
    MaskSeg q;
    {
        const AgcState s0 = A.st_in[c];
        q.g = s0.g; q.y2 = s0.y2; q.S = ms_encode(s0.mode, s0.timer); q.pad = 0;
    }
    // 16-byte loads need row + t to be a multiple of 4 samples: rows start anywhere when nf % 4 != 0 -> scalar loads then
    const bool al = (A.nf & 3u) == 0;
    auto load32 = [&](uint32_t t0, float4 (&v)[8]) {
        if (__builtin_amdgcn_ballot_w64(!(al && t0 + 32u <= A.nf)) == 0ull) {     // the whole wave inside its rows: eight plain 16-byte loads
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = *reinterpret_cast<const float4 *>(Er + t0 + 4u * i);
            return;
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t ti = t0 + 4u * i;
            v[i] = make_float4(ti < A.nf ? Er[ti] : 0.f, ti + 1 < A.nf ? Er[ti + 1] : 0.f, ti + 2 < A.nf ? Er[ti + 2] : 0.f, ti + 3 < A.nf ? Er[ti + 3] : 0.f);
        }
    };
    float4 cur[8], nxt[8];
    load32(t, cur);
    uint32_t mp = 0u, np = 0u, fw, pw;
    // warm-up: only the state matters
    for (; t < t_seg; t += 32u) {
        load32(t + 32u, nxt);
        mask_word<false>(cur, 32u, q, A.p, mp, np, fw, pw);
#pragma unroll
        for (int i = 0; i < 8; i++) cur[i] = nxt[i];
    }
    A.seg_start[(size_t)c * A.nseg + sg] = q;
    // the sample in front of the segment: muted <=> the state it left is not SIGNALHI; its signs from its energy word
    // (sample -1 of the call: from the previous call's last frame)
    mp = q.S != 3u;
    if (t_seg > 0) np = __float_as_uint(Er[t_seg - 1]) >> 31;
    else { const float2 r = A.rp_in[c]; np = (__float_as_uint(r.x) & __float_as_uint(r.y)) >> 31; }
    uint32_t *fl = A.flag + (size_t)c * A.nw, *pl = A.pi + (size_t)c * A.nw;
    for (; t < t_end; t += 32u) {
        if (t + 32u < t_end) load32(t + 32u, nxt);
        // whole words for every lane of the wave: the quad-wise path (its ballot needs uniform control flow)
        const bool whole = t + 32u <= t_end;
        if (__builtin_amdgcn_ballot_w64(!whole) == 0ull) mask_word<false>(cur, 32u, q, A.p, mp, np, fw, pw);
        else mask_word<true>(cur, min(32u, t_end - t), q, A.p, mp, np, fw, pw);
        fl[t >> 5] = fw; pl[t >> 5] = pw;
#pragma unroll
        for (int i = 0; i < 8; i++) cur[i] = nxt[i];
    }
    A.seg_end[(size_t)c * A.nseg + sg] = q;
}

__device__ __forceinline__ bool same_mask_state(const MaskSeg &a, const MaskSeg &b)
{
    return __float_as_uint(a.g) == __float_as_uint(b.g) && __float_as_uint(a.y2) == __float_as_uint(b.y2) && a.S == b.S;
}

// segment s of channel c again, from the state `cur` in front of it
__device__ __forceinline__ void repair_mask_segment(const MaskArgs &A, uint32_t c, uint32_t s, MaskSeg &cur)
{
    const float *Er = A.E + (size_t)c * A.nf;
    const uint32_t t0 = s * A.L, t1 = min(A.nf, t0 + A.L);
    uint32_t mp = cur.S != 3u, np;
    if (t0 > 0) np = __float_as_uint(Er[t0 - 1]) >> 31;
    else { const float2 r = A.rp_in[c]; np = (__float_as_uint(r.x) & __float_as_uint(r.y)) >> 31; }
    uint32_t *fl = A.flag + (size_t)c * A.nw, *pl = A.pi + (size_t)c * A.nw;
    for (uint32_t t = t0; t < t1; t += 32u) {
        float4 v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t ti = t + 4u * i;
            v[i] = make_float4(ti < A.nf ? Er[ti] : 0.f, ti + 1 < A.nf ? Er[ti + 1] : 0.f, ti + 2 < A.nf ? Er[ti + 2] : 0.f, ti + 3 < A.nf ? Er[ti + 3] : 0.f);
        }
        uint32_t fw, pw;
        mask_word<true>(v, min(32u, t1 - t), cur, A.p, mp, np, fw, pw);
        fl[t >> 5] = fw; pl[t >> 5] = pw;
    }
}

__global__ __launch_bounds__(256) void k_agc_mask_fix(MaskArgs A, AgcState *st_out, unsigned *stats)
{
    const uint32_t c = blockIdx.x, tid = threadIdx.x;
    MaskSeg *ss = A.seg_start + (size_t)c * A.nseg, *se = A.seg_end + (size_t)c * A.nseg;
    unsigned redone = 0;
    for (uint32_t round = 0; round < A.nseg; round++) {
        bool changed = false;
        for (uint32_t base = 0; base + 1 < A.nseg; base += 256) {
            const uint32_t s = base + 1 + tid;
            bool need = false;
            MaskSeg e;
            if (s < A.nseg) { e = se[s - 1]; need = !same_mask_state(e, ss[s]); }
            if (!__syncthreads_or(need)) continue;              // (also: everybody has read before anybody writes)
            if (need) {
                MaskSeg cur = e;
                repair_mask_segment(A, c, s, cur);
                ss[s] = e; se[s] = cur;
                changed = true; redone++;
            }
            __threadfence();
            __syncthreads();
            __threadfence();
        }
        if (!__syncthreads_or(changed)) break;
    }
    if (tid == 0) {
        const MaskSeg cur = se[A.nseg - 1];
        AgcState o; o.g = cur.g; o.y2 = cur.y2; ms_decode(cur.S, A.p.timeout, o.mode, o.timer);
        st_out[c] = o;
        if (c == 0) atomicAdd(&stats[0], A.C * (A.nseg - 1));
    }
    if (redone) atomicAdd(&stats[1], redone);
}

// apply the planes: flagged samples are not the plain freqdem of the channelizer output.  One lane per four samples (eight
// lanes per word): a wave instruction covers 1 KiB of a row.
__global__ __launch_bounds__(256) void k_agc_mask_apply(MaskArgs A)
{
    const uint32_t c = blockIdx.y;
    const uint32_t q4 = blockIdx.x * 256u + threadIdx.x;            // group of four samples in the row
    const uint32_t w = q4 >> 3, t0 = 4u * q4;
    if (t0 >= A.nf) return;
    const uint32_t sh = 4u * (q4 & 7u);
    const uint32_t f = (A.flag[(size_t)c * A.nw + w] >> sh) & 15u;
    if (!f) return;
    const uint32_t pb = (A.pi[(size_t)c * A.nw + w] >> sh) & 15u;
    float *o = A.out + (size_t)c * A.nf + t0;
    const float piv = 3.14159265358979324f * A.ref;
    if (f == 15u && pb == 0u && (A.nf & 3u) == 0) { *reinterpret_cast<float4 *>(o) = make_float4(0.f, 0.f, 0.f, 0.f); return; }
    for (uint32_t i = 0; i < 4u && t0 + i < A.nf; i++)
        if ((f >> i) & 1u) o[i] = ((pb >> i) & 1u) ? piv : 0.f;
}

}  // namespace

struct AgcMaskPlan {
    uint32_t C = 0, max_nf = 0, L = 0, Lmin = 384, W = 1024, max_seg = 0, max_nw = 0;
    MaskSeg *d_start = nullptr, *d_end = nullptr;
    uint32_t *d_flag = nullptr, *d_pi = nullptr;
    unsigned *d_stats = nullptr;
    uint32_t wave_slots = 1024;          // waves the device keeps busy at once (one per SIMD: the recurrence is a lone dependent chain per wave)
};

void agc_mask_destroy(AgcMaskPlan *p)
{
    if (!p) return;
    void *ptrs[] = {p->d_start, p->d_end, p->d_flag, p->d_pi, p->d_stats};
    for (void *q : ptrs) if (q) (void)hipFree(q);
    delete p;
}

int agc_mask_create(uint32_t C, uint32_t max_nf, AgcMaskPlan **out)
{
    AgcMaskPlan *p = new AgcMaskPlan();
    p->C = C; p->max_nf = max_nf;
    if (const char *e = getenv("CSDR_AGC_L")) { p->L = (uint32_t)atol(e); p->L = (p->L + 31u) / 32u * 32u; if (p->L < 32) p->L = 32; }
    if (const char *e = getenv("CSDR_AGC_W")) p->W = (uint32_t)atol(e);
    p->W = (p->W + 31u) / 32u * 32u;
    {
        int dev = 0, cus = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        p->wave_slots = (uint32_t)cus * 4u;
        if (const char *e = getenv("CSDR_AGC_WGS")) p->wave_slots = (uint32_t)cus * (uint32_t)(atol(e) > 0 ? atol(e) : 1);
    }
    p->max_seg = (max_nf + 31u) / 32u + 1;
    p->max_nw = (max_nf + 31u) / 32u;
    const size_t n = (size_t)C * p->max_seg;
    if (hipMalloc(&p->d_start, n * sizeof(MaskSeg)) != hipSuccess || hipMalloc(&p->d_end, n * sizeof(MaskSeg)) != hipSuccess ||
        hipMalloc(&p->d_flag, (size_t)C * p->max_nw * 4 + 4) != hipSuccess || hipMalloc(&p->d_pi, (size_t)C * p->max_nw * 4 + 4) != hipSuccess ||
        hipMalloc(&p->d_stats, 2 * sizeof(unsigned)) != hipSuccess) {
        set_error("agc mask: device allocation failed");
        agc_mask_destroy(p);
        return CSDR_ERR_HIP;
    }
    CSDR_HIP(hipMemset(p->d_stats, 0, 2 * sizeof(unsigned)));
    *out = p;
    return 0;
}

int agc_mask_stats(AgcMaskPlan *p, unsigned *checked, unsigned *redone)
{
    unsigned h[2] = {0, 0};
    CSDR_HIP(hipMemcpy(h, p->d_stats, sizeof(h), hipMemcpyDeviceToHost));
    if (checked) *checked = h[0];
    if (redone) *redone = h[1];
    return 0;
}

// E[C][nf] energy words, out[C][nf] FM samples (flagged ones rewritten); st updated in place; rp_prev: the last channelizer
// frame of the call before this one
int agc_mask_process(AgcMaskPlan *p, const float *E, float *out, uint32_t nf, AgcState *st, const AgcParams &prm, float fm_ref,
                     const float2 *rp_prev, hipStream_t s)
{
    if (!nf || !p->C) return 0;
    if ((uint64_t)p->C * nf >= (1ull << 32)) { set_error("agc mask: C*nf = %llu samples exceeds 2^32", (unsigned long long)p->C * nf); return CSDR_ERR_SIZE; }
    // segment length: one 64-stream wave per SIMD over the whole device (more, shorter segments re-read more warm-up), a whole
    // number of 32-sample words, an odd number of them (a power-of-two stride puts all streams on the same HBM channels)
    uint32_t L = p->L;
    if (!L) {
        const uint32_t gmax = p->wave_slots / p->C ? p->wave_slots / p->C : 1u;
        const uint64_t nseg_t = 64ull * gmax;
        L = (uint32_t)((nf + nseg_t - 1) / nseg_t);
        L = (L + 31u) / 32u * 32u;
        if (L < p->Lmin) L = p->Lmin;
        if (((L / 32u) & 1u) == 0) L += 32u;
    }
    const uint32_t nseg = (nf + L - 1) / L;
    if (nseg > p->max_seg) { set_error("agc mask: internal segment bound"); return CSDR_ERR_INVALID; }
    MaskArgs A{};
    A.E = E; A.out = out; A.st_in = st; A.rp_in = rp_prev; A.seg_start = p->d_start; A.seg_end = p->d_end;
    A.flag = p->d_flag; A.pi = p->d_pi;
    A.C = p->C; A.nf = nf; A.L = L; A.W = p->W; A.nseg = nseg; A.nw = (nf + 31u) / 32u; A.p = prm; A.ref = fm_ref;
    const uint32_t groups = (nseg + 63u) / 64u;
    hipLaunchKernelGGL(k_agc_mask_spec, dim3(p->C * groups), dim3(64), 0, s, A, groups);
    hipLaunchKernelGGL(k_agc_mask_fix, dim3(p->C), dim3(256), 0, s, A, st, p->d_stats);
    hipLaunchKernelGGL(k_agc_mask_apply, dim3((nf / 4 + 256) / 256, p->C), dim3(256), 0, s, A);
    CSDR_HIP(hipGetLastError());
    return 0;
}

}  // namespace csdr
