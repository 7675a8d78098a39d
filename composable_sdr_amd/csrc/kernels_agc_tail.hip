// Per-channel AGC + squelch [+ freqdem] tail behind the channelizer, time-parallel and exact.
//
// The reference runs one agc_crcf per channel, one sample at a time (agcExecuteBlock, Liquid.chs:695-705;
// muted unless the squelch is in SIGNALHI, :703-704) and then freqdem on the result (SoapySDR.hs:249).
// The recurrence is non-linear, so one lane per channel is all the parallelism a literal version has
// (256 lanes on a 65 536-lane machine).  This file cuts every channel row into segments of L samples:
//
//   k_agc_spec : one lane per (channel, segment).  The lane starts W samples before its segment from the state
//                the channel had when the call began, runs those W samples without storing anything (the loop
//                gain forgets its start state at |lambda| = 0.95 per sample), records the state it reaches at
//                the segment start, then produces the segment's outputs and records its end state.
//                Segments that begin <= W samples into the call start at sample 0 from the true state instead.
//   k_agc_fix  : one lane per channel walks the segments in order.  A segment whose recorded start state is
//                BITWISE equal to the true end state of the segment before it is, by determinism, exactly what
//                the sequential recurrence produces; any other segment is recomputed sequentially from the
//                true state.  The result is therefore bit-identical to the one-lane-per-channel kernel
//                (k_agc + k_fm in kernels_generic.hip) whatever the signal does; only the speed depends on it.
//
// Memory: a wave owns 64 streams.  Their next 128-byte lines are fetched cooperatively (8 lanes per line, so a
// load instruction covers 8 whole lines), transposed through LDS to one line per lane, and the outputs go back
// the same way (CF32: 8 lanes per 128-byte line, F32: 4 lanes per 64 bytes).
#include "../../include/csdr.h"
#include "csdr_internal.h"
#include "fm_common.h"
#include <cstdlib>

namespace csdr {

namespace {

struct AgcSeg { float g, y2; int32_t mode; uint32_t timer; float rx, ry; uint32_t pad0, pad1; };   // 32 B

// agc_crcf_execute + squelch update + the reference's mute rule; the same code as agc_step in
// kernels_generic.hip (kept textually identical so that both kernels round the same way)
__device__ __forceinline__ float2 agc_tail_step(float2 x, AgcSeg &q, const AgcParams &p)
{
    float2 y = make_float2(x.x * q.g, x.y * q.g);
    const float y2 = fmaf(y.x, y.x, y.y * y.y);          // explicit: must round the same in every kernel
    q.y2 = fmaf(1.0f - p.alpha, q.y2, p.alpha * y2);
    const float upd = __builtin_amdgcn_exp2f((-0.5f * p.alpha) * __builtin_amdgcn_logf(q.y2));
    q.g = (q.y2 > 1e-6f) ? q.g * upd : q.g;
    q.g = fminf(q.g, 1e6f);
    const bool ex = q.g < p.g_thr;                    // rssi > threshold
    int m = q.mode;
    const bool lo_to = (m == 5) && (q.timer == 1u);
    q.timer = (m == 4) ? p.timeout : ((m == 5) ? q.timer - 1u : q.timer);
    const int nxt_ex = (m == 1) ? 2 : ((m == 6) ? 1 : 3);
    const int nxt_no = (m == 1) ? 1 : ((m == 4) ? 5 : ((m == 5) ? 5 : ((m == 6) ? 1 : 4)));
    m = ex ? nxt_ex : nxt_no;
    m = lo_to ? 6 : m;
    q.mode = m;
    if (m != 3) y = make_float2(0.f, 0.f);
    return y;
}

// freqdem of one sample against r' (the same explicitly rounded routine as k_fm)
__device__ __forceinline__ float fm_tail_sample(float2 rp, float2 r, float ref)
{
    return fm_sample_rn(rp, r, ref);
}

__device__ __forceinline__ bool same_state(const AgcSeg &a, const AgcSeg &b, bool fm)
{
    // the timer only lives in SIGNALLO (mode 5): every other mode rewrites it before reading it
    bool ok = __float_as_uint(a.g) == __float_as_uint(b.g) && __float_as_uint(a.y2) == __float_as_uint(b.y2) &&
              a.mode == b.mode && (a.mode != 5 || a.timer == b.timer);
    if (fm) ok = ok && __float_as_uint(a.rx) == __float_as_uint(b.rx) && __float_as_uint(a.ry) == __float_as_uint(b.ry);
    return ok;
}

struct TailArgs {
    const float2 *Z;        // [C][nf] channelizer output
    void *out;              // [C][nf] CF32 or F32
    const AgcState *st_in;  // [C] state before the call
    const float2 *rp_in;    // [C] freqdem r' before the call (FM)
    AgcSeg *seg_start, *seg_end;   // [C][nseg]
    uint32_t C, nf, L, W, nseg;
    AgcParams p;
    float ref;
};

// LDS slot of 16-byte piece `pc` (0..7) of stream `j` (0..63): XOR swizzle, conflict-free for the cooperative
// side (8 streams x 8 pieces per instruction) and for the owner side (64 streams, one piece per instruction)
__device__ __forceinline__ int slot8(int j, int pc) { return 8 * j + (pc ^ ((j ^ (j >> 3)) & 7)); }
__device__ __forceinline__ int slot4(int j, int pc) { return 4 * j + (pc ^ ((j ^ (j >> 2)) & 3)); }

template <bool FM>
__global__ __launch_bounds__(64) void k_agc_spec(TailArgs A)
{
    __shared__ float4 ibuf[64 * 8];
    __shared__ float4 obuf[64 * (FM ? 4 : 8)];
    const int lane = threadIdx.x;
    const uint32_t total = A.C * A.nseg;
    const uint32_t gid0 = blockIdx.x * 64u;
    const uint32_t gid = gid0 + lane;
    const bool mine = gid < total;
    const uint32_t c = mine ? gid / A.nseg : 0, sg = mine ? gid % A.nseg : 0;
    const uint32_t start = sg * A.L, end = min(A.nf, start + A.L);
    const int32_t t00 = (int32_t)start - (int32_t)A.W;          // first sample of block 0 (may be negative)
    const uint32_t nblk = (A.W + A.L) / 16u;

    // helper streams of this lane: instruction m of a cooperative access handles stream 8m + (lane >> 3)
    // (CF32 lines) -- row offset and first sample index of each
    uint32_t hrow[8]; int32_t ht0[8]; uint32_t hend[8];
#pragma unroll
    for (int m = 0; m < 8; m++) {
        const uint32_t g = gid0 + 8 * m + (lane >> 3);
        const bool ok = g < total;
        const uint32_t cc = ok ? g / A.nseg : 0, ss = ok ? g % A.nseg : 0;
        hrow[m] = cc * A.nf;                                   // C*nf < 2^32 samples (checked on the host)
        ht0[m] = (int32_t)(ss * A.L) - (int32_t)A.W;
        hend[m] = ok ? min(A.nf, ss * A.L + A.L) : 0u;        // 0: never loads
    }
    const int pc = lane & 7;

    AgcSeg q;
    {
        const AgcState s0 = A.st_in[c];
        q.g = s0.g; q.y2 = s0.y2; q.mode = s0.mode; q.timer = s0.timer;
        const float2 r0 = FM ? A.rp_in[c] : make_float2(0.f, 0.f);
        q.rx = r0.x; q.ry = r0.y; q.pad0 = q.pad1 = 0;
    }

    float4 ld[8];
    auto coop_load = [&](uint32_t k) {
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const int32_t t = ht0[m] + (int32_t)(16 * k) + 2 * pc;          // first of the two samples of this piece
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t >= 0 && (uint32_t)t + 1 < hend[m]) {
                const float2 *ptr = A.Z + (size_t)hrow[m] + (uint32_t)t;
                if ((((size_t)hrow[m] + (uint32_t)t) & 1) == 0) v = *reinterpret_cast<const float4 *>(ptr);
                else { const float2 a = ptr[0], b = ptr[1]; v = make_float4(a.x, a.y, b.x, b.y); }
            } else if (t >= 0 && (uint32_t)t < hend[m]) {
                const float2 a = A.Z[(size_t)hrow[m] + (uint32_t)t];
                v = make_float4(a.x, a.y, 0.f, 0.f);
            }
            ld[m] = v;
        }
    };

    coop_load(0);
    for (uint32_t k = 0; k < nblk; k++) {
        __syncthreads();                                        // previous block's ibuf / obuf consumed
#pragma unroll
        for (int m = 0; m < 8; m++) ibuf[slot8(8 * m + (lane >> 3), pc)] = ld[m];
        __syncthreads();
        if (k + 1 < nblk) coop_load(k + 1);

        const int32_t t0 = t00 + (int32_t)(16 * k);
        if (mine && k == A.W / 16u) A.seg_start[gid] = q;       // state at the segment start, after the warm-up
        const bool real = k >= A.W / 16u;
        const bool live = mine && t0 >= 0 && (uint32_t)t0 < end;
        float4 o[FM ? 4 : 8];
#pragma unroll
        for (int i = 0; i < (FM ? 4 : 8); i++) o[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) {
            float4 in[8];
#pragma unroll
            for (int i = 0; i < 8; i++) in[i] = ibuf[slot8(lane, i)];
            if ((uint32_t)t0 + 16 <= end) {
                float fo[16];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const float2 a = agc_tail_step(make_float2(in[i].x, in[i].y), q, A.p);
                    if (FM) { fo[2 * i] = fm_tail_sample(make_float2(q.rx, q.ry), a, A.ref); q.rx = a.x; q.ry = a.y; }
                    const float2 b = agc_tail_step(make_float2(in[i].z, in[i].w), q, A.p);
                    if (FM) { fo[2 * i + 1] = fm_tail_sample(make_float2(q.rx, q.ry), b, A.ref); q.rx = b.x; q.ry = b.y; }
                    else o[i] = make_float4(a.x, a.y, b.x, b.y);
                }
                if (FM) {
#pragma unroll
                    for (int i = 0; i < 4; i++) o[i] = make_float4(fo[4 * i], fo[4 * i + 1], fo[4 * i + 2], fo[4 * i + 3]);
                }
            } else {
                // the row's last, partial block
                float fo[16];
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    fo[i] = 0.f;
                    const float4 v = in[i >> 1];
                    float2 a = make_float2(0.f, 0.f);
                    if ((uint32_t)t0 + i < end) {
                        a = agc_tail_step((i & 1) ? make_float2(v.z, v.w) : make_float2(v.x, v.y), q, A.p);
                        if (FM) { fo[i] = fm_tail_sample(make_float2(q.rx, q.ry), a, A.ref); q.rx = a.x; q.ry = a.y; }
                    }
                    if (!FM) { if (i & 1) { o[i >> 1].z = a.x; o[i >> 1].w = a.y; } else { o[i >> 1].x = a.x; o[i >> 1].y = a.y; } }
                }
                if (FM) {
#pragma unroll
                    for (int i = 0; i < 4; i++) o[i] = make_float4(fo[4 * i], fo[4 * i + 1], fo[4 * i + 2], fo[4 * i + 3]);
                }
            }
        }
        if (real) {
            // outputs leave as whole lines: transpose back through LDS
#pragma unroll
            for (int i = 0; i < (FM ? 4 : 8); i++) obuf[FM ? slot4(lane, i) : slot8(lane, i)] = o[i];
            __syncthreads();
            if (FM) {
                float *outp = (float *)A.out;
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const int j = 16 * m + (lane >> 2), p4 = lane & 3;
                    const uint32_t g = gid0 + j;
                    if (g < total) {
                        const uint32_t cc = g / A.nseg, ss = g % A.nseg;
                        const uint32_t t = ss * A.L + 16 * (k - A.W / 16u) + 4 * p4, e = min(A.nf, ss * A.L + A.L);
                        const float4 v = obuf[slot4(j, p4)];
                        float *dst = outp + (size_t)cc * A.nf + t;
                        if (t + 4 <= e && ((((size_t)cc * A.nf + t) & 3) == 0)) *reinterpret_cast<float4 *>(dst) = v;
                        else {
                            if (t < e) dst[0] = v.x;
                            if (t + 1 < e) dst[1] = v.y;
                            if (t + 2 < e) dst[2] = v.z;
                            if (t + 3 < e) dst[3] = v.w;
                        }
                    }
                }
            } else {
                float2 *outp = (float2 *)A.out;
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    const int j = 8 * m + (lane >> 3);
                    const uint32_t g = gid0 + j;
                    if (g < total) {
                        const uint32_t cc = hrow[m] / max(A.nf, 1u), ss = g - cc * A.nseg;
                        const uint32_t t = ss * A.L + 16 * (k - A.W / 16u) + 2 * pc, e = hend[m];
                        const float4 v = obuf[slot8(j, pc)];
                        float2 *dst = outp + (size_t)hrow[m] + t;
                        if (t + 2 <= e && ((((size_t)hrow[m] + t) & 1) == 0)) *reinterpret_cast<float4 *>(dst) = v;
                        else {
                            if (t < e) dst[0] = make_float2(v.x, v.y);
                            if (t + 1 < e) dst[1] = make_float2(v.z, v.w);
                        }
                    }
                }
            }
        }
    }
    if (mine) A.seg_end[gid] = q;
}

// verification + exact fall-back, one wave per channel.  All boundaries are checked in parallel against the
// recorded (speculative) end states; up to the first boundary that fails, those ARE the true states, so the
// common case ends there.  From the first failure on, lane 0 walks the remaining segments in order: a segment
// whose recorded start state equals the true state is taken as recorded, any other is recomputed.
template <bool FM>
__global__ __launch_bounds__(64) void k_agc_fix(TailArgs A, AgcState *st_out, float2 *rp_out, unsigned *stats)
{
    const uint32_t c = blockIdx.x, lane = threadIdx.x;
    const AgcSeg *ss = A.seg_start + (size_t)c * A.nseg, *se = A.seg_end + (size_t)c * A.nseg;
    uint32_t first_bad = A.nseg;
    for (uint32_t s = 1 + lane; s < A.nseg; s += 64) {
        const AgcSeg e = se[s - 1], st = ss[s];
        if (!same_state(e, st, FM)) { first_bad = s; break; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) first_bad = min(first_bad, (uint32_t)__shfl_xor((int)first_bad, d));
    if (lane != 0) return;
    AgcSeg cur = se[first_bad - 1];                             // true state in front of segment first_bad
    unsigned redone = 0;
    for (uint32_t s = first_bad; s < A.nseg; s++) {
        const AgcSeg st = ss[s];
        if (same_state(cur, st, FM)) { cur = se[s]; continue; }
        redone++;
        const uint32_t t0 = s * A.L, t1 = min(A.nf, t0 + A.L);
        const float2 *row = A.Z + (size_t)c * A.nf;
        for (uint32_t t = t0; t < t1; t++) {
            const float2 y = agc_tail_step(row[t], cur, A.p);
            if (FM) {
                ((float *)A.out)[(size_t)c * A.nf + t] = fm_tail_sample(make_float2(cur.rx, cur.ry), y, A.ref);
                cur.rx = y.x; cur.ry = y.y;
            } else ((float2 *)A.out)[(size_t)c * A.nf + t] = y;
        }
    }
    AgcState o; o.g = cur.g; o.y2 = cur.y2; o.mode = cur.mode; o.timer = cur.timer;
    st_out[c] = o;
    if (FM) rp_out[c] = make_float2(cur.rx, cur.ry);
    if (redone) atomicAdd(&stats[1], redone);
    if (c == 0) atomicAdd(&stats[0], A.C * (A.nseg - 1));
}

}  // namespace

struct AgcTailPlan {
    uint32_t C = 0, max_nf = 0, L = 512, W = 1024, max_seg = 0;
    AgcSeg *d_start = nullptr, *d_end = nullptr;
    AgcState *d_st_tmp = nullptr;
    unsigned *d_stats = nullptr;
    uint32_t lanes_target = 131072;      // 2 waves per SIMD
};

int agc_tail_create(uint32_t C, uint32_t max_nf, AgcTailPlan **out)
{
    AgcTailPlan *p = new AgcTailPlan();
    p->C = C; p->max_nf = max_nf;
    if (const char *e = getenv("CSDR_AGC_L")) p->L = (uint32_t)atol(e);
    if (const char *e = getenv("CSDR_AGC_W")) p->W = (uint32_t)atol(e);
    p->L = (p->L + 15u) / 16u * 16u; if (p->L < 16) p->L = 16;
    p->W = (p->W + 15u) / 16u * 16u;
    {
        int dev = 0, cus = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        p->lanes_target = (uint32_t)cus * 4u * 64u * 2u;
    }
    p->max_seg = (max_nf + 15u) / 16u + 1;                      // L >= 16
    const size_t n = (size_t)C * p->max_seg;
    if (hipMalloc(&p->d_start, n * sizeof(AgcSeg)) != hipSuccess || hipMalloc(&p->d_end, n * sizeof(AgcSeg)) != hipSuccess ||
        hipMalloc(&p->d_st_tmp, (size_t)C * sizeof(AgcState)) != hipSuccess || hipMalloc(&p->d_stats, 2 * sizeof(unsigned)) != hipSuccess) {
        set_error("agc tail: device allocation failed");
        agc_tail_destroy(p);
        return CSDR_ERR_HIP;
    }
    CSDR_HIP(hipMemset(p->d_stats, 0, 2 * sizeof(unsigned)));
    *out = p;
    return 0;
}

void agc_tail_destroy(AgcTailPlan *p)
{
    if (!p) return;
    void *ptrs[] = {p->d_start, p->d_end, p->d_st_tmp, p->d_stats};
    for (void *q : ptrs) if (q) (void)hipFree(q);
    delete p;
}

int agc_tail_stats(AgcTailPlan *p, unsigned *checked, unsigned *redone)
{
    unsigned h[2] = {0, 0};
    CSDR_HIP(hipMemcpy(h, p->d_stats, sizeof(h), hipMemcpyDeviceToHost));
    if (checked) *checked = h[0];
    if (redone) *redone = h[1];
    return 0;
}

// Z[C][nf] -> out[C][nf] (CF32, or F32 when fm); st and rp are updated in place.
int agc_tail_process(AgcTailPlan *p, const float2 *Z, void *out, bool fm, uint32_t nf, AgcState *st, const AgcParams &prm,
                     float fm_ref, const float2 *rp_in, float2 *rp_out, hipStream_t s)
{
    if (!nf || !p->C) return 0;
    if ((uint64_t)p->C * nf >= (1ull << 32)) { set_error("agc tail: C*nf = %llu samples exceeds 2^32", (unsigned long long)p->C * nf); return CSDR_ERR_SIZE; }
    // segment length: the configured L, stretched when that would give more lanes than the machine holds at
    // 2 waves per SIMD (longer segments waste less on the warm-up)
    uint32_t L = p->L;
    while ((uint64_t)p->C * ((nf + L - 1) / L) > p->lanes_target && L < (1u << 20)) L *= 2;
    const uint32_t nseg = (nf + L - 1) / L;
    if (nseg > p->max_seg) { set_error("agc tail: internal segment bound"); return CSDR_ERR_INVALID; }
    TailArgs A{};
    A.Z = Z; A.out = out; A.st_in = st; A.rp_in = rp_in; A.seg_start = p->d_start; A.seg_end = p->d_end;
    A.C = p->C; A.nf = nf; A.L = L; A.W = p->W; A.nseg = nseg; A.p = prm; A.ref = fm_ref;
    const uint32_t total = p->C * nseg;
    const dim3 grid((total + 63) / 64), block(64);
    if (fm) hipLaunchKernelGGL(k_agc_spec<true>, grid, block, 0, s, A);
    else hipLaunchKernelGGL(k_agc_spec<false>, grid, block, 0, s, A);
    // the fix-up reads st_in through the segment records only, so st can be overwritten in place
    if (fm) hipLaunchKernelGGL(k_agc_fix<true>, dim3(p->C), dim3(64), 0, s, A, st, rp_out, p->d_stats);
    else hipLaunchKernelGGL(k_agc_fix<false>, dim3(p->C), dim3(64), 0, s, A, st, rp_out, p->d_stats);
    CSDR_HIP(hipGetLastError());
    return 0;
}

}  // namespace csdr
