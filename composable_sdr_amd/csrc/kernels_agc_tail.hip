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
//   k_agc_fix  : one thread per segment boundary.  A segment whose recorded start state is BITWISE equal to the end
//                state of the segment before it is, by determinism (and induction from segment 0, which starts from
//                the true state), exactly what the sequential recurrence produces; segments for which that does not
//                hold are recomputed from the end state in front of them, in parallel, round after round until every
//                boundary holds.  The result is therefore bit-identical to the one-lane-per-channel kernel
//                (k_agc + k_fm in kernels_generic.hip) whatever the signal does; only the speed depends on it.
//
// Memory: a workgroup owns 64 streams and has two waves.  The mover fetches the streams' next 128-byte lines
// cooperatively (8 lanes per line, so a load instruction covers 8 whole lines), transposes them through an LDS ring
// to one line per lane for the worker, and sends the outputs back the same way (CF32: 8 lanes per 128-byte line,
// F32: 4 lanes per 64 bytes); the worker only ever touches LDS.
#include "../../include/csdr.h"
#include "csdr_internal.h"
#include "fm_common.h"
#include "agc_common.h"
#include <cstdlib>
#include <cstdio>
#ifndef CSDR_AGC_ABLATE
#define CSDR_AGC_ABLATE 0      // timing experiments only: 1 no log/exp, 2 every block reloads the same (cached) lines, 4 no stores, 16 no compute, 32 mover skips the ring writes, 64 no barrier
#endif

#ifndef TM_ABLATE
#define TM_ABLATE 0          // timing experiments only: 1 post wave without the freqdem arithmetic, 2 gain wave without the recurrence, 4 no output stores
#endif
#ifndef TM_TRACE
#define TM_TRACE 0           // 1 (variant build, tools/build_variant.sh): where the two waves of three workgroups spend their cycles (device printf)
#endif
namespace csdr {

namespace {

struct AgcSeg { float g, y2; int32_t mode; uint32_t timer; float rx, ry; uint32_t pad0, pad1; };   // 32 B

// agc_crcf_execute + squelch update + the reference's mute rule.  The float path is agc_common.h's, shared with
// agc_step of kernels_generic.hip (both kernels must round the same way); the squelch state machine
// (agc_crcf_squelch_update_mode, modes 1..6 + timer) is folded into ONE integer S so that a step costs ~12
// integer instructions instead of ~27:
//   S = 1 ENABLED, 2 RISE, 3 SIGNALHI, 4 FALL, 8 TIMEOUT, 8+k SIGNALLO with k samples left on the timer
// SIGNALLO counts down by S-1 and runs into TIMEOUT (8) on its own; the other transitions come out of two 3-bit
// tables indexed by S.  The timer of the reference only lives in SIGNALLO (every other mode rewrites it before
// reading it), so nothing is lost.
constexpr uint32_t S_TEX = (2u << 3) | (3u << 6) | (3u << 9) | (3u << 12) | (1u << 24);   // threshold exceeded
constexpr uint32_t S_TNO = (1u << 3) | (4u << 6) | (4u << 9) | (5u << 12) | (1u << 24);   // not exceeded; 5 = "enter SIGNALLO"
__device__ __forceinline__ uint32_t s_encode(int32_t mode, uint32_t timer) { return mode == 6 ? 8u : (mode == 5 ? 8u + timer : (uint32_t)mode); }
__device__ __forceinline__ void s_decode(uint32_t S, uint32_t timeout, int32_t &mode, uint32_t &timer)
{
    mode = S > 8u ? 5 : (S == 8u ? 6 : (int32_t)S);
    timer = S > 8u ? S - 8u : timeout;
}
// the float path of one sample (agc_common.h): updates g and y2', returns the un-muted output
__device__ __forceinline__ float2 agc_gain_step(float2 x, AgcSeg &q, const AgcParams &p)
{
    const float2 y = make_float2(x.x * q.g, x.y * q.g);
    if (CSDR_AGC_ABLATE & 1) { q.y2 = fmaf(1.0f - p.alpha, q.y2, agc_energy(x, p.alpha) * (q.g * q.g)); q.g = __builtin_amdgcn_fmed3f(q.g * (1.0f - 1e-3f * (q.y2 - 1.0f)), 0.0f, 1e6f); }
    else agc_gain_update(agc_energy(x, p.alpha), q.g, q.y2, p.alpha);
    return y;
}
// squelch state after a sample
__device__ __forceinline__ uint32_t squelch_next(uint32_t S, bool ex, const AgcParams &p)
{
    uint32_t t = __builtin_amdgcn_ubfe(ex ? S_TEX : S_TNO, 3u * S, 3u);
    t = (t == 5u) ? 8u + p.timeout : t;
    const uint32_t r9 = (ex && S >= 10u) ? 3u : S - 1u;
    return (S >= 9u) ? r9 : t;
}
__device__ __forceinline__ float2 agc_tail_step(float2 x, AgcSeg &q, const AgcParams &p)
{
    float2 y = agc_gain_step(x, q, p);
    const uint32_t Sn = squelch_next((uint32_t)q.mode, q.g < p.g_thr, p);   // AgcSeg.mode carries S inside this file
    q.mode = (int32_t)Sn;
    if (Sn != 3u) y = make_float2(0.f, 0.f);          // reference mute rule (Liquid.chs:703-704)
    return y;
}

// freqdem of one sample against r' (the same explicitly rounded routine as k_fm)
__device__ __forceinline__ float fm_tail_sample(float2 rp, float2 r, float ref)
{
    return fm_sample_rn(rp, r, ref);
}

__device__ __forceinline__ bool same_state(const AgcSeg &a, const AgcSeg &b, bool fm)
{
    bool ok = __float_as_uint(a.g) == __float_as_uint(b.g) && __float_as_uint(a.y2) == __float_as_uint(b.y2) && a.mode == b.mode;
    if (fm) ok = ok && __float_as_uint(a.rx) == __float_as_uint(b.rx) && __float_as_uint(a.ry) == __float_as_uint(b.ry);
    return ok;
}

struct TailArgs {
    const float2 *Z;        // [C][nf] channelizer output; tm: TILE-MAJOR plane, sample (c, t) at ((t >> 4) C + c) 16 + (t & 15), with
                            // TM_GUARD blocks of readable memory in front of block 0 and behind the last one
    void *out;              // [C][nf] CF32 or F32
    const AgcState *st_in;  // [C] state before the call
    const AgcState *st_spec;// [C] state the speculative warm-ups start from: st_in, except on a stream's first call (k_agc_pilot)
    const float2 *rp_in;    // [C] freqdem r' before the call (FM)
    AgcSeg *seg_start, *seg_end;   // [C][nseg]
    AgcSeg *ckpt;           // tile-major route: [C][nseg][nck] state after every TM_CK samples of a segment (k_agc_fix stops a repair where it meets them)
    uint32_t nck;
    uint32_t C, nf, L, W, nseg;
    AgcParams p;
    float ref;
    uint32_t tm;            // 1: Z is tile-major (what the fused run kernels write for k_agc_spec_tm)
};
constexpr uint32_t TM_CK = 256;         // samples between the checkpoints of a segment
constexpr uint32_t TM_GUARD = 512;      // blocks of 16 samples: >= max(W, L) / 16 of the tile-major route (W, L <= 8192)

// index of sample (c, t) in the channelizer plane
__device__ __forceinline__ size_t z_index(const TailArgs &A, uint32_t c, uint32_t t)
{
    return A.tm ? ((size_t)(t >> 4) * A.C + c) * 16u + (t & 15u) : (size_t)c * A.nf + t;
}

// LDS slot of 16-byte piece `pc` (0..7) of stream `j` (0..63): XOR swizzle, conflict-free for the cooperative
// side (8 streams x 8 pieces per instruction) and for the owner side (64 streams, one piece per instruction)
__device__ __forceinline__ int slot8(int j, int pc) { return 8 * j + (pc ^ ((j ^ (j >> 3)) & 7)); }

// four consecutive samples of one stream: AGC (+ freqdem); GUARD: only samples t < end exist
// The g recurrence does not depend on the squelch state, and most samples leave the squelch where it is (SIGNALHI
// with the threshold exceeded, ENABLED without).  A whole quad therefore runs the four gain steps first and decides
// ONCE, for the wave, whether any lane's state moves: if not, the transition tables are skipped (one branch per
// four samples on the dependent chain instead of four); the freqdem of the four samples follows as independent work.
template <bool FM, bool GUARD>
__device__ __forceinline__ void agc_quad(const float4 &va, const float4 &vb, AgcSeg &q, const AgcParams &p, float ref,
                                         uint32_t t, uint32_t end, float4 &oa, float4 &ob)
{
    float2 y[4];
    float m[4] = {0.f, 0.f, 0.f, 0.f};
    const float2 x[4] = {make_float2(va.x, va.y), make_float2(va.z, va.w), make_float2(vb.x, vb.y), make_float2(vb.z, vb.w)};
    if (GUARD) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            y[i] = make_float2(0.f, 0.f);
            if (t + i < end) {
                y[i] = agc_tail_step(x[i], q, p);
                if (FM) { m[i] = fm_tail_sample(make_float2(q.rx, q.ry), y[i], ref); q.rx = y[i].x; q.ry = y[i].y; }
            }
        }
    } else {
        // "threshold exceeded by all / by none of the four" from the running max / min of g (v_med3 with +-inf): the
        // per-sample compares would each hand a mask to the scalar unit, and the wave waits for every such hand-over
        float gs[4], gmx = -INFINITY, gmn = INFINITY;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            y[i] = agc_gain_step(x[i], q, p);
            gs[i] = q.g;
            gmx = __builtin_amdgcn_fmed3f(q.g, gmx, INFINITY);
            gmn = __builtin_amdgcn_fmed3f(q.g, gmn, -INFINITY);
        }
        uint32_t S = (uint32_t)q.mode;
        const bool all_ex = gmx < p.g_thr, none_ex = !(gmn < p.g_thr);
        const bool steady = (S == 3u && all_ex) || (S == 1u && none_ex);
        if (__builtin_amdgcn_ballot_w64(!steady) == 0ull) {
            const bool open = S == 3u;                          // the state holds through the quad
#pragma unroll
            for (int i = 0; i < 4; i++) y[i] = open ? y[i] : make_float2(0.f, 0.f);
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                S = squelch_next(S, gs[i] < p.g_thr, p);
                if (S != 3u) y[i] = make_float2(0.f, 0.f);      // reference mute rule (Liquid.chs:703-704)
            }
            q.mode = (int32_t)S;
        }
        if (FM) {
#pragma unroll
            for (int i = 0; i < 4; i++) m[i] = fm_tail_sample(i ? y[i - 1] : make_float2(q.rx, q.ry), y[i], ref);
            q.rx = y[3].x; q.ry = y[3].y;
        }
    }
    if (FM) oa = make_float4(m[0], m[1], m[2], m[3]);
    else { oa = make_float4(y[0].x, y[0].y, y[1].x, y[1].y); ob = make_float4(y[2].x, y[2].y, y[3].x, y[3].y); }
}

typedef float tm_v2f __attribute__((ext_vector_type(2)));

// Four consecutive samples of one stream through the gain recurrence + squelch (bit for bit agc_tail_step x 4).  WANT_Y: also the
// (muted) AGC outputs.  What is off the dependent chain runs packed (alpha |x|^2 of two samples in three v_pk, y = x g in one), the
// "does the squelch state move at all" test works on the v_cmp lane masks in SGPRs (scalar unit), and the mute is an AND with a
// per-lane all-ones / zero word: the VCC selects hipcc emits for `cond ? a : b` cost 16 cycles each on gfx950.
template <bool WANT_Y>
__device__ __forceinline__ void agc_gain_quad(const float4 va, const float4 vb, AgcSeg &q, const AgcParams &p, float2 (&y)[4])
{
    const tm_v2f xr0 = {va.x, va.z}, xi0 = {va.y, va.w}, xr1 = {vb.x, vb.z}, xi1 = {vb.y, vb.w};
    const tm_v2f al = {p.alpha, p.alpha};
    // agc_energy: alpha * fmaf(x.x, x.x, x.y * x.y), two samples per instruction
    const tm_v2f e0 = al * __builtin_elementwise_fma(xr0, xr0, xi0 * xi0), e1 = al * __builtin_elementwise_fma(xr1, xr1, xi1 * xi1);
    const float e[4] = {e0.x, e0.y, e1.x, e1.y};
    const float2 x[4] = {make_float2(va.x, va.y), make_float2(va.z, va.w), make_float2(vb.x, vb.y), make_float2(vb.z, vb.w)};
    float gs[4];
    unsigned long long mex[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (WANT_Y) { const tm_v2f yy = (tm_v2f){x[i].x, x[i].y} * (tm_v2f){q.g, q.g}; y[i] = make_float2(yy.x, yy.y); }
        if (TM_ABLATE & 2) { q.y2 += e[i]; q.g = __builtin_amdgcn_fmed3f(q.g + 1e-9f * q.y2, 0.0f, 1e6f); }
        else agc_gain_update(e[i], q.g, q.y2, p.alpha);
        gs[i] = q.g;
        asm("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(mex[i]) : "v"(q.g), "v"(p.g_thr));       // threshold exceeded after this sample
    }
    uint32_t S = (uint32_t)q.mode;
    unsigned long long m3, m1;
    asm("v_cmp_eq_u32_e64 %0, 3, %1" : "=s"(m3) : "v"(S));
    asm("v_cmp_eq_u32_e64 %0, 1, %1" : "=s"(m1) : "v"(S));
    const unsigned long long all_ex = mex[0] & mex[1] & mex[2] & mex[3], any_ex = mex[0] | mex[1] | mex[2] | mex[3];
    const unsigned long long steady = (m3 & all_ex) | (m1 & ~any_ex);
    const unsigned long long act = __builtin_amdgcn_ballot_w64(true);
    if ((act & ~steady) == 0ull) {
        if (WANT_Y) {
            const uint32_t keep = __float_as_uint(fm_sel(m3, __uint_as_float(0xffffffffu), 0.0f));     // SIGNALHI holds through the quad: open
#pragma unroll
            for (int i = 0; i < 4; i++) y[i] = make_float2(__uint_as_float(__float_as_uint(y[i].x) & keep), __uint_as_float(__float_as_uint(y[i].y) & keep));
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            S = squelch_next(S, gs[i] < p.g_thr, p);
            if (WANT_Y && S != 3u) y[i] = make_float2(0.f, 0.f);      // reference mute rule (Liquid.chs:703-704)
        }
        q.mode = (int32_t)S;
    }
}

// One lane per (channel, segment) stream; a workgroup is 64 consecutive segments of ONE channel (grid = channels x
// segment groups, so every address is the uniform row plus a lane-derived segment) -- or, when a row has fewer than 33
// segments (many channels, short calls: 4096 channels x the reference's 4096-frame chunk is 11 segments of 384), spw = 8,
// 16 or 32 segments of 64 / spw neighbouring channels each, so that the lanes are not left idle (round 5: such calls ran
// one 11-lane workgroup per channel, four rounds of workgroups deep) -- served by TWO waves:
//   * the worker (wave 0) runs the recurrence: block k of 16 samples per stream comes out of an LDS ring slot one line
//     per lane, four samples at a time, and the outputs go back over the stream's own consumed input pieces;
//   * the mover (wave 1) does everything that waits for memory: while block k is worked on it writes block k + 1
//     (fetched during the previous iteration, 8 lanes per 128-byte line) into the next ring slot, issues the loads of
//     block k + 2 and stores the outputs of block k - 1 as whole lines.
// One LDS-only barrier per block (s_waitcnt lgkmcnt(0); s_barrier -- __syncthreads() would drain the loads in
// flight).  With a single wave doing load -> LDS -> arithmetic -> LDS -> store one after the other nothing overlapped:
// at the segment lengths that keep the warm-up re-reads low there are too few streams for the other waves of a SIMD
// to cover for it (0.44 ms per 67 M samples however the arithmetic was trimmed; data path alone 0.36, arithmetic
// alone 0.33).
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <bool FM, bool PAIRS>
__global__ __launch_bounds__(128, 2) void k_agc_spec(TailArgs A, uint32_t groups, uint32_t ls)   // <= 256 VGPRs
{
    __shared__ float4 ring[3][64 * 8];
    const int lane = threadIdx.x & 63;
    const bool mover = threadIdx.x >= 64;                       // wave-uniform
    // stream j of the workgroup: channel c + (j >> ls), segment sbase + (j & (spw - 1)); ls = 6: 64 segments of one channel
    const uint32_t spw = 1u << ls, cpw = 64u >> ls;
    const uint32_t c = (blockIdx.x / groups) * cpw, sbase = (blockIdx.x % groups) * spw;   // uniform: first channel, first segment
    const size_t row = (size_t)c * A.nf;
    const uint32_t nblk = (A.W + A.L) / 16u, kreal = A.W / 16u;
    const int pc = lane & 7;
    auto seg_of = [&](uint32_t j) { return sbase + (j & (spw - 1u)); };
    auto chr_of = [&](uint32_t j) { return j >> ls; };          // channel of stream j, relative to c

    // block k: instruction m of the cooperative access handles stream 8m + (lane >> 3).  Branch-free (clamped
    // address) so that the eight loads are in flight together: with a branch per load the compiler waits for each one
    // before the next and the block pays eight memory latencies in a row.  The halves that lie outside the stream are
    // zeroed by mask_block when the block goes into the ring, two iterations later: masking here made the mover wait for
    // the loads it had just issued -- a full memory round trip per block on the groups that start inside the first W
    // samples of a row, which is EVERY block of a quarter of the workgroups at the reference's chunk size (round 5,
    // TM_TRACE: 3200 of the mover's 4800 cycles per block, the worker waiting for it 40 % of the time).
    const size_t total = (size_t)A.C * A.nf;
    auto load_block = [&](uint32_t k, float4 (&ld)[8]) {
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const uint32_t jj = 8 * m + (lane >> 3), ss = seg_of(jj);
            const int32_t t = (int32_t)(ss * A.L) - (int32_t)A.W + (int32_t)(16 * k) + 2 * pc;   // first of the piece's two samples
            const size_t idx = (size_t)min(c + chr_of(jj), A.C - 1u) * A.nf + (uint32_t)max(t, 0);
            if (PAIRS) ld[m] = *reinterpret_cast<const float4 *>(A.Z + min(idx, total - 2));
            else { const float2 a = A.Z[min(idx, total - 1)], b = A.Z[min(idx + 1, total - 1)]; ld[m] = make_float4(a.x, a.y, b.x, b.y); }
        }
    };
    auto mask_block = [&](uint32_t k, float4 (&ld)[8]) {
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const uint32_t jj = 8 * m + (lane >> 3), ss = seg_of(jj);
            const int32_t t = (int32_t)(ss * A.L) - (int32_t)A.W + (int32_t)(16 * k) + 2 * pc;
            const uint32_t e = min(A.nf, ss * A.L + A.L);
            const bool there = ss < A.nseg && c + chr_of(jj) < A.C;
            const bool ok0 = there && t >= 0 && (uint32_t)t < e, ok1 = there && t >= 0 && (uint32_t)t + 1 < e;
            const float4 v = ld[m];
            ld[m] = make_float4(ok0 ? v.x : 0.f, ok0 ? v.y : 0.f, ok1 ? v.z : 0.f, ok1 ? v.w : 0.f);
        }
    };

    // Interior blocks -- all 64 streams of the group exist and the block lies inside every stream's range (all but the
    // first W samples of a row and its ragged end) -- need no range tests at all: one uniform base pointer and a
    // 32-bit lane offset per piece.  The general path's clamps, masks and 64-bit addresses cost as many VALU
    // instructions per block as sixteen warm-up steps.
    const bool group_full = PAIRS && sbase + spw <= A.nseg && c + cpw <= A.C;
    const int64_t t_first = (int64_t)sbase * A.L - (int64_t)A.W;                    // the first segment's sample at k = 0
    const int64_t t_lastend = (int64_t)(sbase + spw - 1u) * A.L - (int64_t)A.W + 15;    // the last segment's last sample at k = 0
    const char *wbase = reinterpret_cast<const char *>(A.Z + row) + t_first * 8;    // uniform; dereferenced on interior blocks only
    // bytes from the workgroup's first sample to my piece of stream 8m + (lane >> 3) (mover only; the launcher keeps cpw rows inside 32 bits)
    uint32_t voff[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int m = 0; m < 8 && mover; m++) {
        const uint32_t jj = 8 * m + (lane >> 3);
        voff[m] = (chr_of(jj) * A.nf + (jj & (spw - 1u)) * A.L + 2u * (uint32_t)pc) * 8u;
    }
    auto is_inner = [&](uint32_t k) { return group_full && t_first + 16 * (int64_t)k >= 0 && t_lastend + 16 * (int64_t)k < (int64_t)A.nf; };
    // block k's eight pieces -> registers (not waited for here on interior blocks: nothing touches the data)
    auto fetch = [&](uint32_t k, float4 (&ld)[8]) {
        if (is_inner(k)) {
#pragma unroll
            for (int m = 0; m < 8; m++) ld[m] = *reinterpret_cast<const float4 *>(wbase + (voff[m] + 128u * ((CSDR_AGC_ABLATE & 2) ? (k >= kreal ? kreal : 0u) : k)));
        } else load_block(k, ld);
    };

    // outputs of block k (>= kreal) leave as whole lines
    auto store_block = [&](uint32_t k, const float4 *buf) {
        if ((CSDR_AGC_ABLATE & 4) && A.nf != 0xffffffffu) return;
        const uint32_t tb = 16 * (k - kreal);
        if (is_inner(k) && (!FM || ((row & 3) == 0 && (cpw == 1u || (A.nf & 3u) == 0)))) {
            // interior block, 16-byte aligned rows: the same uniform-base addressing as the loads
            if (FM) {
                char *obase = reinterpret_cast<char *>((float *)A.out + row + (size_t)sbase * A.L + tb);
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const uint32_t jj = 16 * m + (lane >> 2);
                    const uint32_t off = (chr_of(jj) * A.nf + (jj & (spw - 1u)) * A.L + 4u * (uint32_t)(lane & 3)) * 4u;
                    *reinterpret_cast<float4 *>(obase + off) = buf[slot8((int)jj, lane & 3)];
                }
            } else {
                char *obase = reinterpret_cast<char *>((float2 *)A.out + row + (size_t)sbase * A.L + tb);
#pragma unroll
                for (int m = 0; m < 8; m++)
                    *reinterpret_cast<float4 *>(obase + voff[m]) = buf[slot8(8 * m + (lane >> 3), pc)];
            }
            return;
        }
        if (FM) {
            float *outp = (float *)A.out;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const int j = 16 * m + (lane >> 2), p4 = lane & 3;
                const uint32_t ss = seg_of((uint32_t)j), cc = c + chr_of((uint32_t)j);
                if (ss < A.nseg && cc < A.C) {
                    const uint32_t t = ss * A.L + tb + 4 * p4, e = min(A.nf, ss * A.L + A.L);
                    const float4 v = buf[slot8(j, p4)];
                    const size_t idx = (size_t)cc * A.nf + t;
                    float *dst = outp + idx;
                    if (t + 4 <= e && (idx & 3) == 0) *reinterpret_cast<float4 *>(dst) = v;
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
                const uint32_t jj = 8 * m + (lane >> 3), ss = seg_of(jj), cc = c + chr_of(jj);
                if (ss < A.nseg && cc < A.C) {
                    const uint32_t t = ss * A.L + tb + 2 * pc, e = min(A.nf, ss * A.L + A.L);
                    const float4 v = buf[slot8(8 * m + (lane >> 3), pc)];
                    const size_t idx = (size_t)cc * A.nf + t;
                    float2 *dst = outp + idx;
                    if (t + 2 <= e && (idx & 1) == 0) *reinterpret_cast<float4 *>(dst) = v;
                    else {
                        if (t < e) dst[0] = make_float2(v.x, v.y);
                        if (t + 1 < e) dst[1] = make_float2(v.z, v.w);
                    }
                }
            }
        }
    };

    // the worker's stream
    const uint32_t sg = seg_of((uint32_t)lane), cw = min(c + chr_of((uint32_t)lane), A.C - 1u);     // the worker's segment and channel
    const bool mine = sg < A.nseg && c + chr_of((uint32_t)lane) < A.C;
    const uint32_t endv = mine ? min(A.nf, sg * A.L + A.L) : 0u;
    AgcSeg q;
    {
        // segments that begin <= W samples into the call run from sample 0 and from the TRUE state; the others warm up from st_spec
        const AgcState s0 = ((uint64_t)sg * A.L > A.W) ? A.st_spec[cw] : A.st_in[cw];
        q.g = s0.g; q.y2 = s0.y2; q.mode = (int32_t)s_encode(s0.mode, s0.timer); q.timer = 0;
        const float2 r0 = FM ? A.rp_in[cw] : make_float2(0.f, 0.f);
        q.rx = r0.x; q.ry = r0.y; q.pad0 = q.pad1 = 0;
    }
    auto work_block = [&](uint32_t k, float4 *buf) {
        const int32_t t0 = (int32_t)(sg * A.L) - (int32_t)A.W + (int32_t)(16 * k);
        if (mine && k == kreal) A.seg_start[(size_t)cw * A.nseg + sg] = q;      // state at the segment start, after the warm-up
        const bool inner = is_inner(k);
        const bool live = inner || (t0 >= 0 && (uint32_t)t0 < endv);
        const bool full = inner || !live || (uint32_t)t0 + 16 <= endv;
        if ((CSDR_AGC_ABLATE & 16)) return;
        if (k < kreal) {
            // warm-up block (always whole): only the state matters -- no freqdem, nothing stored
            if (live && k + 1 < kreal) {
                // (round 5) the whole block out of the ring at once and the state-only quad of the tile-major kernel: the four LDS round trips of
                // the loop below and its unpacked arithmetic were a third of a warm-up block, and a call of the reference's chunk size is all warm-up
                float4 v[8];
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = buf[slot8(lane, i)];
#pragma unroll
                for (int h = 0; h < 4; h++) { float2 y[4]; agc_gain_quad<false>(v[2 * h], v[2 * h + 1], q, A.p, y); }
            } else if (live) {
                // the last warm-up block also leaves r', the (possibly muted) AGC output in front of the segment
#pragma unroll 1
                for (int h = 0; h < 4; h++) {
                    const float4 va = buf[slot8(lane, 2 * h)], vb = buf[slot8(lane, 2 * h + 1)];
                    float4 oa, ob;
                    agc_quad<false, false>(va, vb, q, A.p, A.ref, 0u, 0u, oa, ob);
                    if (FM) { q.rx = ob.z; q.ry = ob.w; }       // r' = the last (possibly muted) AGC output
                }
            }
        } else if (__builtin_amdgcn_ballot_w64(!full) == 0ull) {
            if (live) {
#pragma unroll 1
                for (int h = 0; h < 4; h++) {
                    const float4 va = buf[slot8(lane, 2 * h)], vb = buf[slot8(lane, 2 * h + 1)];
                    float4 oa, ob;
                    agc_quad<FM, false>(va, vb, q, A.p, A.ref, (uint32_t)t0 + 4 * h, endv, oa, ob);
                    // in place: output piece h (F32) / pieces 2h, 2h+1 (CF32) over input pieces already consumed
                    if (FM) buf[slot8(lane, h)] = oa;
                    else { buf[slot8(lane, 2 * h)] = oa; buf[slot8(lane, 2 * h + 1)] = ob; }
                }
            }
        } else if (live) {
#pragma unroll 1
            for (int h = 0; h < 4; h++) {
                const float4 va = buf[slot8(lane, 2 * h)], vb = buf[slot8(lane, 2 * h + 1)];
                float4 oa, ob;
                agc_quad<FM, true>(va, vb, q, A.p, A.ref, (uint32_t)t0 + 4 * h, endv, oa, ob);
                if (FM) buf[slot8(lane, h)] = oa;
                else { buf[slot8(lane, 2 * h)] = oa; buf[slot8(lane, 2 * h + 1)] = ob; }
            }
        }
    };

    // iteration `it`: the mover fills slot it % 3 with block it and empties slot (it - 2) % 3, the worker is on
    // block it - 1 in slot (it - 1) % 3.  The mover keeps TWO blocks in flight (it + 1 and it + 2, two register sets
    // used alternately -- hence the loop runs in pairs): with one, an iteration lasted a memory round trip.
    float4 lda[8], ldb[8];
    uint32_t s_in = 0, s_wk = 2, s_out = 1;                     // it % 3, (it - 1) % 3, (it - 2) % 3
    if (mover) { fetch(0, lda); if (1 < nblk) fetch(1, ldb); }
    [[maybe_unused]] unsigned long long r_bar = 0, r_a = 0, r_b = 0, r_c = 0, r_t0 = TM_TRACE ? __builtin_amdgcn_s_memtime() : 0ull, r_t = 0;
#define RM_STAMP(acc) do { if (TM_TRACE) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); acc += n_ - r_t; r_t = n_; } } while (0)
    auto iteration = [&](uint32_t it, float4 (&ld)[8]) {       // ld holds block it (mover)
        if (TM_TRACE) r_t = __builtin_amdgcn_s_memtime();
        if (mover) {
            if (it < nblk) {
                if (CSDR_AGC_ABLATE & 32) {
#pragma unroll
                    for (int m = 0; m < 8; m++) { const float t0 = ld[m].x, t1 = ld[m].w; asm volatile("" :: "v"(t0), "v"(t1)); }
                } else {
                    if (!is_inner(it)) mask_block(it, ld);
#pragma unroll
                    for (int m = 0; m < 8; m++) ring[s_in][slot8(8 * m + (lane >> 3), pc)] = ld[m];
                }
                if (TM_TRACE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                RM_STAMP(r_a);                                  // mover: wait for block it + ring writes
                if (it + 2 < nblk) fetch(it + 2, ld);
                RM_STAMP(r_b);                                  // mover: issue of the loads of block it + 2
            }
            if (it >= 2 && it - 2 >= kreal && it - 2 < nblk) store_block(it - 2, ring[s_out]);
            RM_STAMP(r_c);                                      // mover: stores
        } else if (it >= 1 && it - 1 < nblk) { work_block(it - 1, ring[s_wk]); if (TM_TRACE) asm volatile("" :: "v"(q.g), "v"(q.y2)); RM_STAMP(r_a); }
        if (!(CSDR_AGC_ABLATE & 64)) lds_barrier();
        RM_STAMP(r_bar);
        s_out = s_wk; s_wk = s_in; s_in = s_in == 2 ? 0 : s_in + 1;
    };
    for (uint32_t it = 0; it < nblk + 2; it += 2) {
        iteration(it, lda);
        iteration(it + 1, ldb);                                 // it + 1 may be nblk + 2: nothing left to do but the barrier
    }
#if TM_TRACE
    if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2 || blockIdx.x == gridDim.x - 1))
        printf("rm trace wg %u %s: total %llu cycles over %u blocks (%u warm-up), barrier %llu, %s %llu, load issue %llu, stores %llu\n", blockIdx.x, mover ? "mover" : "worker",
               (unsigned long long)(__builtin_amdgcn_s_memtime() - r_t0), nblk, kreal, r_bar, mover ? "wait for the block + ring writes" : "work", r_a, r_b, r_c);
#endif
    if (!mover && mine) A.seg_end[(size_t)cw * A.nseg + sg] = q;
}


// ---------------------------------------------------------------------------------------------------------------------------
// k_agc_spec_tm (round 4): the same speculation on a TILE-MAJOR channelizer plane.
//
// k_agc_spec's 64 streams are 64 segments of one channel: every block iteration fetches 64 lines that lie a segment (8 KiB)
// apart, one DRAM page each (4.1 TB/s however many were in flight), through registers, and the worker wave pays ~33
// instructions per sample for the freqdem with VCC selects.  Here the fused run kernels write the CF32 plane tile-major -- block
// B of 16 frames holds the 128-byte lines of all C channels back to back, i.e. a tile's whole output is ONE contiguous 32 KiB --
// and a workgroup's 64 streams are 64 CHANNELS at the same segment: every lane is at the same time position, block kk of the
// workgroup is one contiguous 8 KiB, and nothing in the kernel depends on the lane any more except the state.
//   * mover wave: eight global_load_lds_dwordx4 per block (HBM -> LDS ring without registers, XOR swizzle on the source address as
//     in k_run256v2), TM_DEPTH blocks ahead; its vmcnt queue holds the DMA plus a few state records (the seg_start record at the end of the warm-up and a
//     checkpoint every TM_CK samples are global stores inside the loop, and stores count in vmcnt on gfx9), so `s_waitcnt vmcnt(8 n)` can only
//     wait for MORE than the n youngest blocks: always safe, exact in the iterations without such a store;
//   * worker wave: the recurrence out of the ring, freqdem as fm_quad_rn (packed, SGPR-mask selects, bit-identical to
//     fm_sample_rn), and the outputs straight from its registers as 64-byte (F32) / 128-byte (CF32) row pieces -- nothing goes back
//     through the ring;
//   * one LDS-only barrier per block.
// Channels per workgroup Cw = min(C, 64); for C < 64 (interleaved shards of eight) a workgroup takes 64 / C segments.
// Same records, same verification (k_agc_fix), same bits as k_agc_spec.
#ifndef TM_DEPTH_N
#define TM_DEPTH_N 3
#endif
constexpr int TM_DEPTH = TM_DEPTH_N;        // blocks in flight
static_assert(TM_DEPTH >= 1 && TM_DEPTH <= 6, "vmcnt immediates above");
constexpr int TM_SLOTS = TM_DEPTH + 2;      // ring slots of 8 KiB: TM_DEPTH in flight, one with the gain wave, one with the post wave

// Gain wave (wave 0): tile DMA TM_DEPTH blocks ahead, the recurrence out of the ring, y back in place (blocks >= kreal - 1).
// Post wave (wave 1): block it - 1's y -> freqdem (FM: fm_quad_rn) -> row pieces straight to HBM; its vmcnt queue holds stores only.
template <bool FM>
__global__ __launch_bounds__(128, 2) void k_agc_spec_tm(TailArgs A, uint32_t ncg, uint32_t Cw, uint32_t nsub)
{
    __shared__ float4 ring[TM_SLOTS][64 * 8];
    const int lane = threadIdx.x & 63;
    const bool post = threadIdx.x >= 64;                        // wave-uniform
    const uint32_t cg = blockIdx.x % ncg, sgrp = blockIdx.x / ncg;      // neighbouring workgroups: neighbouring 8 KiB of the same blocks
    const uint32_t nblk = (A.W + A.L) / 16u, kreal = A.W / 16u;
    const size_t blk_bytes = (size_t)A.C * 128u;                // one block of the plane
    // stream jj of the workgroup: channel cg Cw + jj % Cw, segment sgrp nsub + jj / Cw
    auto seg_of = [&](uint32_t jj) { return sgrp * nsub + jj / Cw; };
    auto ch_of = [&](uint32_t jj) { return cg * Cw + jj % Cw; };
    const uint32_t sg = seg_of((uint32_t)lane), ch = ch_of((uint32_t)lane);
    const bool mine = sg < A.nseg && ch < A.C;
    const uint32_t endv = mine ? min(A.nf, sg * A.L + A.L) : 0u;
    // block `it` of my stream starts at sample t0(it); whole blocks only (nf, L, W are multiples of 16)
    auto live_at = [&](uint32_t it) { const int32_t t0 = (int32_t)(sg * A.L) - (int32_t)A.W + (int32_t)(16 * it); return mine && t0 >= 0 && (uint32_t)t0 < endv; };

    if (post) {
        const FmRnK fk = fm_rn_consts(A.ref);
        // FM: r' in front of the block in work: the previous call's last output for a stream that starts at sample 0 from the true
        // state, the last warm-up output (read out of block kreal - 1's slot) for the others
        float2 rp = (FM && mine && !((uint64_t)sg * A.L > A.W)) ? A.rp_in[ch] : make_float2(0.f, 0.f);
        // Outputs leave as whole row pieces, cooperatively: the results go back into the block's ring slot (slot8 layout, this wave
        // owns the slot by now) and store instruction mm writes piece lane & 7 of stream 8 mm + (lane >> 3) -- eight lanes per 128-byte
        // line.  F32: a block is only 64 bytes of a row, so an even block's results wait in registers and the odd block stores both
        // (segments start on 128-byte lines: L is a multiple of 32 on this route).
        uint32_t ooff[8];                                       // byte offset of (stream 8 mm + (lane >> 3))'s segment start + my piece
        bool ook[8];
#pragma unroll
        for (int mm = 0; mm < 8; mm++) {
            const uint32_t jj = 8u * mm + ((uint32_t)lane >> 3), s2 = seg_of(jj), c2 = ch_of(jj);
            ook[mm] = s2 < A.nseg && c2 < A.C;
            ooff[mm] = ook[mm] ? (uint32_t)(((size_t)c2 * A.nf + (size_t)s2 * A.L) * (FM ? 4u : 8u)) + 16u * ((uint32_t)lane & 7u) : 0u;
        }
        char *obase = reinterpret_cast<char *>(A.out);
        float4 hold[4];
        bool held = false;                                      // (uniform per sub-segment; nsub > 1: lanes of different segments agree, the segments are equally long except the call's last)
        // store the lines of blocks [k0, k0 + nb) (nb = 1: CF32 block or a lone F32 block; nb = 2: an F32 pair) out of `buf`
        auto coop_store = [&](float4 *buf, uint32_t k0, bool pair) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int mm = 0; mm < 8; mm++) {
                const uint32_t jj = 8u * mm + ((uint32_t)lane >> 3), s2 = seg_of(jj);
                const uint32_t e2 = min(A.nf, s2 * A.L + A.L);
                const int32_t t0 = (int32_t)(s2 * A.L) - (int32_t)A.W + (int32_t)(16 * k0);
                // F32: pieces 0..3 are block k0, pieces 4..7 block k0 + 1 (pair) -- each must lie inside the segment
                const int32_t tp = t0 + (FM ? 16 * (int32_t)(((uint32_t)lane & 7u) >> 2) : 0);
                const bool ok = ook[mm] && t0 >= 0 && (uint32_t)tp < e2 && (FM && !pair ? ((uint32_t)lane & 7u) < 4u : true);
                if (ok && !(TM_ABLATE & 4)) {
                    const float4 v = buf[slot8((int)jj, lane & 7)];
                    *reinterpret_cast<float4 *>(obase + ooff[mm] + (size_t)(16u * (k0 - kreal)) * (FM ? 4u : 8u)) = v;
                }
            }
        };
        [[maybe_unused]] unsigned long long p_bar = 0, p_lds = 0, p_fm = 0, p_st = 0, p_t0 = TM_TRACE ? __builtin_amdgcn_s_memtime() : 0ull;
        for (uint32_t it = 0; it <= nblk; it++) {
            const unsigned long long pb0 = TM_TRACE ? __builtin_amdgcn_s_memtime() : 0ull;
            lds_barrier();                                      // block it - 1's y is in its slot; the gain wave is on block it
            if (TM_TRACE) p_bar += __builtin_amdgcn_s_memtime() - pb0;
            if (it == 0) continue;
            const uint32_t k = it - 1;
            if (k + 1 < kreal) continue;                        // warm-up blocks leave nothing here, except the last one: r' of the segment
            float4 *buf = ring[k % TM_SLOTS];
            const bool lv = live_at(k);
            if (k < kreal) { if (FM && lv) { const float4 v7 = buf[slot8(lane, 7)]; rp = make_float2(v7.z, v7.w); } continue; }
            if (!FM) { coop_store(buf, k, false); continue; }
            float4 mq[4];
            [[maybe_unused]] unsigned long long pq = TM_TRACE ? __builtin_amdgcn_s_memtime() : 0ull;
            if (lv) {
                float4 v[8];
#pragma unroll
                for (int pc = 0; pc < 8; pc++) v[pc] = buf[slot8(lane, pc)];
                if (TM_TRACE) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); p_lds += n_ - pq; pq = n_; }
#pragma unroll
                for (int h = 0; h < 4; h++) {
                    const float2 y[4] = {make_float2(v[2 * h].x, v[2 * h].y), make_float2(v[2 * h].z, v[2 * h].w),
                                         make_float2(v[2 * h + 1].x, v[2 * h + 1].y), make_float2(v[2 * h + 1].z, v[2 * h + 1].w)};
                    const float2 rq[4] = {rp, y[0], y[1], y[2]};
                    float m[4];
                    if (TM_ABLATE & 1) { m[0] = rq[0].x + y[0].y; m[1] = rq[1].x + y[1].y; m[2] = rq[2].x + y[2].y; m[3] = rq[3].x + y[3].y; }
                    else fm_quad_rn(rq, y, fk, m);
                    rp = y[3];
                    mq[h] = make_float4(m[0], m[1], m[2], m[3]);
                }
            } else {
#pragma unroll
                for (int h = 0; h < 4; h++) mq[h] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (TM_TRACE) { asm volatile("" :: "v"(mq[0].x), "v"(mq[3].w)); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); p_fm += n_ - pq; pq = n_; }
            const bool odd = ((k - kreal) & 1u) != 0u;
            if (!odd) {
#pragma unroll
                for (int h = 0; h < 4; h++) hold[h] = mq[h];
                held = true;
                // a segment's (or the call's) last block has no partner when the block count is odd: it leaves alone
                const bool last_any = __builtin_amdgcn_ballot_w64(lv && !live_at(k + 1)) != 0ull || k + 1 == nblk;
                if (last_any) {
#pragma unroll
                    for (int h = 0; h < 4; h++) buf[slot8(lane, h)] = hold[h];
                    coop_store(buf, k, false);
                    held = false;
                }
            } else {
#pragma unroll
                for (int h = 0; h < 4; h++) { buf[slot8(lane, h)] = hold[h]; buf[slot8(lane, 4 + h)] = mq[h]; }
                coop_store(buf, k - 1, true);
                held = false;
            }
            if (TM_TRACE) p_st += __builtin_amdgcn_s_memtime() - pq;
        }
        (void)held;
#if TM_TRACE
        if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2 || blockIdx.x == gridDim.x - 1))
            printf("tm trace wg %u post: total %llu cycles over %u blocks (%u warm-up), at the barrier %llu, ring reads %llu, freqdem %llu, ring writes + row stores %llu\n", blockIdx.x,
                   (unsigned long long)(__builtin_amdgcn_s_memtime() - p_t0), nblk, kreal, p_bar, p_lds, p_fm, p_st);
#endif
        return;
    }

    // ---- gain wave
    // lane l of DMA instruction m fills ring piece 64 m + l = slot8(jj, pc): jj = 8 m + (l >> 3), pc = (l & 7) ^ ((jj ^ (jj >> 3)) & 7)
    uint32_t voff[8];
#pragma unroll
    for (int m = 0; m < 8; m++) {
        const uint32_t jj = 8u * m + ((uint32_t)lane >> 3), pc = ((uint32_t)lane & 7u) ^ ((jj ^ (jj >> 3)) & 7u);
        uint32_t s2 = seg_of(jj);
        if (s2 >= A.nseg) s2 = A.nseg - 1;                      // a stream that does not exist reads what its neighbour reads
        uint32_t c2 = ch_of(jj);
        if (c2 >= A.C) c2 = A.C - 1;
        const int64_t b0 = ((int64_t)s2 * A.L - (int64_t)A.W) / 16 + (int64_t)TM_GUARD;     // >= 0: the plane has TM_GUARD blocks in front
        voff[m] = (uint32_t)((size_t)b0 * blk_bytes + (size_t)c2 * 128u + pc * 16u);
    }
    const char *base = reinterpret_cast<const char *>(A.Z) - (size_t)TM_GUARD * blk_bytes;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float4 *)&ring[0][0];
    auto dma = [&](uint32_t kk_) {
        const uint32_t kk = (uint32_t)__builtin_amdgcn_readfirstlane((int)kk_);     // (uniform anyway: keeps the operands below in SGPRs)
        const char *src = base + (size_t)kk * blk_bytes;
        unsigned dst = lds0 + (kk % TM_SLOTS) * 8192u;
        asm volatile("" : "+s"(dst));
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const unsigned d = dst + 1024u * (unsigned)m;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(voff[m]), "s"(d), "s"(src) : "memory");
        }
    };
    AgcSeg q;
    {
        // segments that begin <= W samples into the call run from sample 0 and from the TRUE state; the others warm up from st_spec
        const uint32_t cc = mine ? ch : 0u;
        const AgcState s0 = ((uint64_t)sg * A.L > A.W) ? A.st_spec[cc] : A.st_in[cc];
        q.g = s0.g; q.y2 = s0.y2; q.mode = (int32_t)s_encode(s0.mode, s0.timer); q.timer = 0;
        const float2 r0 = FM ? A.rp_in[cc] : make_float2(0.f, 0.f);
        q.rx = r0.x; q.ry = r0.y; q.pad0 = q.pad1 = 0;
    }
    for (uint32_t d = 0; d < (uint32_t)TM_DEPTH && d < nblk; d++) dma(d);
    [[maybe_unused]] unsigned long long g_vm = 0, g_bar = 0, g_dma = 0, g_lds = 0, g_cmp = 0, g_cwu = 0, g_wu = 0, g_t0 = TM_TRACE ? __builtin_amdgcn_s_memtime() : 0ull, g_t;
#define TM_STAMP(acc) do { if (TM_TRACE) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); acc += n_ - g_t; g_t = n_; } } while (0)
    for (uint32_t it = 0; it <= nblk; it++) {
        if (TM_TRACE) { g_t = __builtin_amdgcn_s_memtime(); if (it == kreal) g_wu = g_t - g_t0; }
        if (it < nblk) {
            // (Round 5 tried a third wave that owns the DMA and its wait -- they cost this wave 725 of its ~3200 cycles per warm-up block, TM_TRACE --
            // and found that the blocks then simply arrive later: 16 MiB in flight at the ~4 TB/s this read + scattered-write mix reaches IS
            // the ~4 us a block takes to land.  The kernel is bound by its bytes, not by who issues them: 528 against 510 us per step, not kept.)
            // block `it` has landed once at most min(TM_DEPTH - 1, nblk - 1 - it) younger blocks (8 instructions each) are in flight:
            // this wave's vmcnt queue holds the DMA and nothing else (the state records below are stored after the loop)
            const uint32_t younger = min((uint32_t)TM_DEPTH - 1u, nblk - 1u - it);
            if (younger >= 5u) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
            else if (younger == 4u) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            else if (younger == 3u) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else if (younger == 2u) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (younger == 1u) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        TM_STAMP(g_vm);
        lds_barrier();                                          // the post wave has left block it - 2's slot, and takes block it - 1's y
        TM_STAMP(g_bar);
        if (it >= nblk) break;
        if (it + TM_DEPTH < nblk) dma(it + TM_DEPTH);           // into the slot block it - 2 had
        TM_STAMP(g_dma);
        if (it == kreal) {
            // state at the segment start, after the warm-up: kept in registers until the loop is over (a store here would sit in the
            // vmcnt queue in front of the DMA counts above)
        }
        float4 *buf = ring[it % TM_SLOTS];
        if (!live_at(it)) continue;
        float4 v[8];
#pragma unroll
        for (int pc = 0; pc < 8; pc++) v[pc] = buf[slot8(lane, pc)];
        if (TM_TRACE) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); TM_STAMP(g_lds); }
        if (it + 1 < kreal) {
#pragma unroll
            for (int h = 0; h < 4; h++) { float2 y[4]; agc_gain_quad<false>(v[2 * h], v[2 * h + 1], q, A.p, y); }
            if (TM_TRACE) { asm volatile("" :: "v"(q.g), "v"(q.y2)); TM_STAMP(g_cwu); }
        } else {
            if (it == kreal) A.seg_start[(size_t)ch * A.nseg + sg] = q;     // (one 32-byte store per stream and launch)
#pragma unroll
            for (int h = 0; h < 4; h++) {
                float2 y[4];
                agc_gain_quad<true>(v[2 * h], v[2 * h + 1], q, A.p, y);
                buf[slot8(lane, 2 * h)] = make_float4(y[0].x, y[0].y, y[1].x, y[1].y);
                buf[slot8(lane, 2 * h + 1)] = make_float4(y[2].x, y[2].y, y[3].x, y[3].y);
                if (h == 3) { q.rx = y[3].x; q.ry = y[3].y; }
            }
            // a checkpoint every TM_CK samples: where a repair (k_agc_fix) that starts from another state meets this trajectory
            // bit for bit, everything behind it is already what the sequential recurrence produces
            const uint32_t done = 16u * (it - kreal + 1u);
            if (it >= kreal && done % TM_CK == 0u && done < A.L) A.ckpt[((size_t)ch * A.nseg + sg) * A.nck + (done / TM_CK - 1u)] = q;
            if (TM_TRACE) { asm volatile("" :: "v"(q.g), "v"(q.y2)); TM_STAMP(g_cmp); }
        }
    }
#if TM_TRACE
    if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2 || blockIdx.x == gridDim.x - 1))
        printf("tm trace wg %u gain: total %llu cycles over %u blocks (warm-up %u blocks: %llu), vmcnt wait %llu, barrier %llu, dma issue %llu, lds reads %llu, recurrence of the warm-up blocks %llu, recurrence + y + ring writes of the others %llu\n",
               blockIdx.x, (unsigned long long)(__builtin_amdgcn_s_memtime() - g_t0), nblk, kreal, g_wu, g_vm, g_bar, g_dma, g_lds, g_cwu, g_cmp);
#endif
    if (mine) A.seg_end[(size_t)ch * A.nseg + sg] = q;
}

// verification + exact repair, one workgroup per channel, one thread per segment boundary.
// A boundary holds when the state the segment was started from (recorded, S_s) is BITWISE the end state of the
// segment before it (E_{s-1}).  S_0 is the true state, so when every boundary holds every segment is, by
// determinism, exactly what the sequential recurrence produces.  Boundaries that do not hold are repaired in
// rounds: every failing segment is recomputed -- all of them in parallel, one lane each -- from the CURRENT E_{s-1},
// which also becomes its S_s.  After a round the first failing segment of the row is final, so the loop ends after
// at most nseg rounds; in practice failures are isolated and one or two rounds do (a repaired segment almost always
// runs into the end state it had before).  Reads and writes of a round are separated by barriers, so nobody compares
// against a half-written record.
// returns true when the repair met the recorded trajectory at a checkpoint (tile-major route): the segment's recorded end state
// and everything behind the checkpoint stay as they are
template <bool FM>
__device__ __forceinline__ bool repair_segment(const TailArgs &A, uint32_t c, uint32_t s, AgcSeg &cur)
{
    const uint32_t t0 = s * A.L, t1 = min(A.nf, t0 + A.L);
    const size_t rowo = (size_t)c * A.nf;
    auto one = [&](float2 x, uint32_t t) {
        const float2 y = agc_tail_step(x, cur, A.p);
        if (FM) {
            ((float *)A.out)[rowo + t] = fm_tail_sample(make_float2(cur.rx, cur.ry), y, A.ref);
            cur.rx = y.x; cur.ry = y.y;
        } else ((float2 *)A.out)[rowo + t] = y;
    };
    uint32_t t = t0;
    if (A.tm) {                                                 // tile-major: t0, t1 are multiples of 16, a block is one 128-byte line
        AgcSeg *ck = A.ckpt + ((size_t)c * A.nseg + s) * A.nck;
        const FmRnK fk = fm_rn_consts(A.ref);
        for (; t + 16 <= t1; t += 16) {
            // a block at a time with the speculation's own quad routines (same arithmetic, bit for bit): a repair lane is alone on
            // its dependent chain, so what it does NOT issue (scalar freqdem, VCC selects, 4-byte stores) is what shortens it
            const float4 *src = reinterpret_cast<const float4 *>(A.Z + z_index(A, c, t));
            float4 v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = src[i];
#pragma unroll
            for (int h = 0; h < 4; h++) {
                float2 y[4];
                agc_gain_quad<true>(v[2 * h], v[2 * h + 1], cur, A.p, y);
                if (FM) {
                    const float2 rq[4] = {make_float2(cur.rx, cur.ry), y[0], y[1], y[2]};
                    float m[4];
                    fm_quad_rn(rq, y, fk, m);
                    *reinterpret_cast<float4 *>((float *)A.out + rowo + t + 4 * h) = make_float4(m[0], m[1], m[2], m[3]);
                    cur.rx = y[3].x; cur.ry = y[3].y;
                } else {
                    float4 *o = reinterpret_cast<float4 *>((float2 *)A.out + rowo + t + 4 * h);
                    o[0] = make_float4(y[0].x, y[0].y, y[1].x, y[1].y); o[1] = make_float4(y[2].x, y[2].y, y[3].x, y[3].y);
                }
            }
            const uint32_t done = t + 16 - t0;
            if (done % TM_CK == 0u && done < A.L && t + 16 < t1) {
                AgcSeg &rec = ck[done / TM_CK - 1u];
                if (same_state(rec, cur, FM)) return true;      // from here on the outputs and states on file ARE this trajectory
                rec = cur;                                      // the checkpoints follow the trajectory whose outputs are in memory
            }
        }
        return false;
    }
    const float2 *row = A.Z + rowo;
    if (((rowo + t) & 1) && t < t1) { one(row[t], t); t++; }    // up to a 16-byte boundary
    for (; t + 16 <= t1; t += 16) {                             // a line per iteration, all eight loads in flight
        float4 v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = *reinterpret_cast<const float4 *>(row + t + 2 * i);
#pragma unroll
        for (int i = 0; i < 8; i++) { one(make_float2(v[i].x, v[i].y), t + 2 * i); one(make_float2(v[i].z, v[i].w), t + 2 * i + 1); }
    }
    for (; t < t1; t++) one(row[t], t);
    return false;
}

// A stream's first call starts from the reference's create-time state (g = 1000, Liquid.chs:707-717), which 1024 samples of
// warm-up do not forget: every speculative segment would fail its boundary check and the exact repair would recompute the whole
// call in rounds (80 ms at the bench size).  The pilot runs the recurrence over the first `n` samples of every channel once, one
// lane per channel, output discarded: the state it reaches is SETTLED, and that is all a warm-up needs to start from (the
// verification in k_agc_fix keeps the result exact whatever the speculation started from).
__global__ __launch_bounds__(64) void k_agc_pilot(TailArgs A, uint32_t n_, AgcState *st_spec)
{
    const uint32_t c = blockIdx.x * 64u + threadIdx.x;
    uint32_t n = n_;
    if (c >= A.C) return;
    const AgcState s0 = A.st_in[c];
    AgcSeg cur; cur.g = s0.g; cur.y2 = s0.y2; cur.mode = (int32_t)s_encode(s0.mode, s0.timer); cur.timer = 0; cur.rx = cur.ry = 0.f; cur.pad0 = cur.pad1 = 0;
    const size_t rowo = (size_t)c * A.nf;
    const float2 *row = A.Z + rowo;
    uint32_t t = 0;
    if (A.tm) {
        for (; t + 16 <= n; t += 16) {
            const float4 *src = reinterpret_cast<const float4 *>(A.Z + z_index(A, c, t));
            float4 v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = src[i];
#pragma unroll
            for (int i = 0; i < 8; i++) { (void)agc_tail_step(make_float2(v[i].x, v[i].y), cur, A.p); (void)agc_tail_step(make_float2(v[i].z, v[i].w), cur, A.p); }
        }
        n = t;
    }
    if (!A.tm && (rowo & 1) && t < n) { (void)agc_tail_step(row[t], cur, A.p); t++; }
    for (; t + 16 <= n; t += 16) {
        float4 v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = *reinterpret_cast<const float4 *>(row + t + 2 * i);
#pragma unroll
        for (int i = 0; i < 8; i++) { (void)agc_tail_step(make_float2(v[i].x, v[i].y), cur, A.p); (void)agc_tail_step(make_float2(v[i].z, v[i].w), cur, A.p); }
    }
    for (; t < n; t++) (void)agc_tail_step(row[t], cur, A.p);
    AgcState o; o.g = cur.g; o.y2 = cur.y2; s_decode((uint32_t)cur.mode, A.p.timeout, o.mode, o.timer);
    st_spec[c] = o;
}

template <bool FM>
__global__ __launch_bounds__(256) void k_agc_fix(TailArgs A, AgcState *st_out, float2 *rp_out, unsigned *stats)
{
    const uint32_t c = blockIdx.x, tid = threadIdx.x;
    AgcSeg *ss = A.seg_start + (size_t)c * A.nseg, *se = A.seg_end + (size_t)c * A.nseg;
    unsigned redone = 0;
    for (uint32_t round = 0; round < A.nseg; round++) {
        bool changed = false;
        for (uint32_t base = 0; base + 1 < A.nseg; base += 256) {
            const uint32_t s = base + 1 + tid;
            bool need = false;
            AgcSeg e;
            if (s < A.nseg) { e = se[s - 1]; need = !same_state(e, ss[s], FM); }
            if (!__syncthreads_or(need)) continue;              // (also: everybody has read before anybody writes)
            if (need) {
                AgcSeg cur = e;
                const bool met = repair_segment<FM>(A, c, s, cur);
                ss[s] = e;
                if (!met) se[s] = cur;                          // (met: the recorded end state is the end of this very trajectory)
                changed = true; redone++;
            }
            __threadfence();
            __syncthreads();                                    // records complete before the next chunk / round reads them
            __threadfence();
        }
        if (!__syncthreads_or(changed)) break;
    }
    if (tid == 0) {
        const AgcSeg cur = se[A.nseg - 1];
        AgcState o; o.g = cur.g; o.y2 = cur.y2; s_decode((uint32_t)cur.mode, A.p.timeout, o.mode, o.timer);
        st_out[c] = o;
        if (FM) rp_out[c] = make_float2(cur.rx, cur.ry);
        if (c == 0) atomicAdd(&stats[0], A.C * (A.nseg - 1));
    }
    if (redone) atomicAdd(&stats[1], redone);
}

}  // namespace

struct AgcTailPlan {
    uint32_t C = 0, max_nf = 0, L = 0, Lmin = 384, W = 1024, max_seg = 0;   // L > 0: fixed by CSDR_AGC_L
    uint32_t L_tm = 0;               // tile-major route: fixed segment length (CSDR_AGC_L_TM), 0 = chosen per call
    uint32_t tm_calls = 0;           // calls that took k_agc_spec_tm since create
    AgcSeg *d_start = nullptr, *d_end = nullptr, *d_ckpt = nullptr;
    size_t ckpt_cap = 0;
    AgcState *d_st_tmp = nullptr;    // [C] settled state of the pilot (first call of a stream)
    bool fresh = true;               // the AGC state is still inside its create-time transient: the next call runs the pilot
    uint32_t pilot_n = 4096;
    uint64_t seen = 0;               // samples per channel since create / reset
    unsigned *d_stats = nullptr;
    uint32_t wg_slots = 1024;            // workgroups the device holds at once
};

int agc_tail_create(uint32_t C, uint32_t max_nf, AgcTailPlan **out)
{
    AgcTailPlan *p = new AgcTailPlan();
    p->C = C; p->max_nf = max_nf;
    if (TM_ABLATE) fprintf(stderr, "csdr: kernels_agc_tail.hip built with TM_ABLATE=%d: timing only, results are wrong\n", TM_ABLATE);
    if (const char *e = diag_env("CSDR_AGC_L")) { p->L = (uint32_t)atol(e); p->L = (p->L + 15u) / 16u * 16u; if (p->L < 16) p->L = 16; }
    if (const char *e = diag_env("CSDR_AGC_L_TM")) { p->L_tm = ((uint32_t)atol(e) + 15u) / 16u * 16u; if (p->L_tm < 16) p->L_tm = 16; if (p->L_tm > 8176u) p->L_tm = 8176u; }
    if (const char *e = diag_env("CSDR_AGC_W")) p->W = (uint32_t)atol(e);
    p->W = (p->W + 15u) / 16u * 16u;
    {
        int dev = 0, cus = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (const char *e = diag_env("CSDR_CUS")) { if (atoi(e) > 0 && atoi(e) < cus) cus = atoi(e); }      // experiments: a plan sized for a CU-masked stream
        // k_agc_spec: two waves per workgroup at <= 256 VGPRs -> two waves per SIMD -> four workgroups per CU
        p->wg_slots = (uint32_t)cus * 4u;
        if (const char *e = diag_env("CSDR_AGC_WGS")) p->wg_slots = (uint32_t)cus * (uint32_t)(atol(e) > 0 ? atol(e) : 1);
    }
    p->max_seg = (max_nf + 15u) / 16u + 1;                      // L >= 16
    const size_t n = (size_t)C * p->max_seg;
    p->ckpt_cap = (size_t)C * ((size_t)max_nf / 128u + 64u);    // >= nseg x ceil(L / TM_CK) for every L >= 128 of the tile-major route
    if (hipMalloc(&p->d_start, n * sizeof(AgcSeg)) != hipSuccess || hipMalloc(&p->d_end, n * sizeof(AgcSeg)) != hipSuccess ||
        hipMalloc(&p->d_ckpt, p->ckpt_cap * sizeof(AgcSeg)) != hipSuccess ||
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
    void *ptrs[] = {p->d_start, p->d_end, p->d_ckpt, p->d_st_tmp, p->d_stats};
    for (void *q : ptrs) if (q) (void)hipFree(q);
    delete p;
}

void agc_tail_reset(AgcTailPlan *p) { if (p) { p->fresh = true; p->seen = 0; } }
uint32_t agc_tail_tm_calls(const AgcTailPlan *p) { return p ? p->tm_calls : 0u; }

int agc_tail_stats(AgcTailPlan *p, unsigned *checked, unsigned *redone)
{
    unsigned h[2] = {0, 0};
    CSDR_HIP(hipMemcpy(h, p->d_stats, sizeof(h), hipMemcpyDeviceToHost));
    if (checked) *checked = h[0];
    if (redone) *redone = h[1];
    return 0;
}

// Z[C][nf] -> out[C][nf] (CF32, or F32 when fm); st and rp are updated in place.
// the tile-major route (k_agc_spec_tm): whole 16-frame blocks, at least two segments, channel count a multiple or a divisor of 64
bool agc_tail_tm_supported(const AgcTailPlan *p, uint32_t nf)
{
    static const bool off = diag_env("CSDR_AGC_TM") && atoi(diag_env("CSDR_AGC_TM")) == 0;      // A/B: the row-major route for every call
    if (off || !p || !nf || nf % 16u || p->W > 8192u || p->W % 16u) return false;
    if (!(p->C % 64u == 0 || (p->C < 64u && 64u % p->C == 0))) return false;
    if (nf < 4u * p->W) return false;                            // short calls: a handful of segments, the row-major kernel's ground
    if (((uint64_t)nf / 16u + 2ull * TM_GUARD) * p->C * 128ull >= (1ull << 32)) return false;     // 32-bit byte offsets inside the plane
    return true;
}
size_t agc_tail_tm_guard(uint32_t C) { return (size_t)TM_GUARD * 16u * C; }      // float2 elements in front of and behind a tile-major plane

int agc_tail_process(AgcTailPlan *p, const float2 *Z, void *out, bool fm, uint32_t nf, AgcState *st, const AgcParams &prm,
                     float fm_ref, const float2 *rp_in, float2 *rp_out, hipStream_t s, bool tm)
{
    if (!nf || !p->C) return 0;
    if ((uint64_t)p->C * nf >= (1ull << 32)) { set_error("agc tail: C*nf = %llu samples exceeds 2^32", (unsigned long long)p->C * nf); return CSDR_ERR_SIZE; }
    // Segment length: as many 64-segment groups per channel as the device holds workgroups at once (one round, every
    // SIMD busy), not more -- shorter segments re-read more warm-up ((W + L) / L times the data) and a second round of
    // workgroups costs more than it brings.  L / 16 is made odd: with a power-of-two L the 64 streams of a group and
    // the groups of all channels hit the same few HBM channels at every step (L = 1024: 0.37 ms, 1040: 0.34 ms).
    if (tm && !agc_tail_tm_supported(p, nf)) { set_error("agc tail: internal: tile-major plane for a call the tile-major kernel does not take"); return CSDR_ERR_INVALID; }
    const uint32_t Cw = p->C < 64u ? p->C : 64u, nsub = 64u / Cw, ncg = p->C / Cw;     // tile-major: channels per workgroup, segments per workgroup, channel groups
    uint32_t L = tm ? p->L_tm : p->L;
    // tile-major route: the two-thirds rule below gives L >= 384 on its own once the plane passes ~128 MiB; below that the device is
    // under-filled and the re-reads come out of the L2s / the MALL, so shorter segments pay (round 5, profiles/r05_call_size_sweeps.txt:
    // 256 channels x 16 384 frames 153 -> 137 us, 1024 x 4096 151 -> 136 us with 128 against 384; nothing below 128)
    const uint32_t lmin_tm = 128u;
    if (!L && tm) {
        // two thirds of the workgroups the device holds at once (ncg channel groups x nseg / nsub segment groups): measured optimum
        // between the warm-up re-reads ((W + L) / L times the plane, shorter segments) and the blocks a workgroup walks (longer ones);
        // profiles/r04_agc_tm_segment_sweep.txt
        uint64_t nseg_t = (uint64_t)p->wg_slots * nsub * 2u / (3u * ncg);
        if (nseg_t < 2) nseg_t = 2;
        L = (uint32_t)((nf + nseg_t - 1) / nseg_t);
        L = (L + 31u) / 32u * 32u;                               // F32 rows leave as whole 128-byte lines per block pair
        if (L < lmin_tm) L = lmin_tm;
        if (L > 8160u) L = 8160u;
    }
    if (tm && L < lmin_tm) L = lmin_tm;                          // (CSDR_AGC_L_TM below the plan's bounds: the checkpoint table is sized for L >= Lmin)
    if (tm && (L % 32u)) L = (L + 31u) / 32u * 32u;
    if (!L) {
        // as many segments per row as the device has lanes for (the kernel packs the segments of several channels into a workgroup when
        // a row has few)
        const uint64_t nseg_t = 64ull * p->wg_slots / p->C ? 64ull * p->wg_slots / p->C : 1ull;
        L = (uint32_t)((nf + nseg_t - 1) / nseg_t);
        L = (L + 15u) / 16u * 16u;
        // Lmin bounds the warm-up re-reads ((W + L) / L times the plane) of a bandwidth-bound call.  A call whose whole plane sits in the
        // L2s (the reference's own 4096-frame chunk: 8 MiB at 256 channels) is bound by the walk instead -- (W + L) samples of a ~200-cycle
        // recurrence per lane -- and 11 segments of 400 left 53 of a workgroup's 64 lanes idle: such calls take every segment the device
        // has a lane for (round 5: 161 -> ~120 us per reference chunk)
        const uint32_t lmin = ((uint64_t)p->C * nf * sizeof(float2) <= (24ull << 20)) ? 16u : p->Lmin;
        if (L < lmin) L = lmin;
        if (((L / 16u) & 1u) == 0) L += 16u;
    }
    const uint32_t nseg = (nf + L - 1) / L;
    if (nseg > p->max_seg) { set_error("agc tail: internal segment bound"); return CSDR_ERR_INVALID; }
    TailArgs A{};
    A.Z = Z; A.out = out; A.st_in = st; A.rp_in = rp_in; A.seg_start = p->d_start; A.seg_end = p->d_end;
    A.C = p->C; A.nf = nf; A.L = L; A.W = p->W; A.nseg = nseg; A.p = prm; A.ref = fm_ref;
    A.st_spec = st; A.tm = tm ? 1u : 0u;
    A.ckpt = p->d_ckpt; A.nck = (L + TM_CK - 1u) / TM_CK;
    if (tm && (size_t)p->C * nseg * A.nck > p->ckpt_cap) { set_error("agc tail: internal checkpoint bound"); return CSDR_ERR_INVALID; }
    if (p->fresh && nseg > 1) {
        const uint32_t n = nf < p->pilot_n ? nf : p->pilot_n;
        hipLaunchKernelGGL(k_agc_pilot, dim3((p->C + 63u) / 64u), dim3(64), 0, s, A, n, p->d_st_tmp);
        A.st_spec = p->d_st_tmp;
    }
    // the create-time transient is over once the stream has seen pilot_n samples, however many calls that took (round 5: a stream of
    // 1024- or 2048-frame calls used to pilot for ever -- one serial lane per channel over the whole call, 180 / 267 us per call)
    p->seen += nf;
    if (p->seen >= p->pilot_n) p->fresh = false;
    // PAIRS: nf even, so every row starts on a 16-byte boundary and ends on one (t is always even): a piece is one
    // 16-byte load that never reaches past the buffer
    const bool pairs = (nf & 1u) == 0 && (uint64_t)p->C * nf >= 2;
    // row-major kernel: spw segments x cpw channels per workgroup (cpw > 1 needs whole-piece rows and 32-bit offsets inside its rows)
    uint32_t ls = 6;
    if (nseg <= 32u && pairs && (nf & 3u) == 0 && p->C >= 8u && (uint64_t)nf * 64ull < (1ull << 32)) { ls = 3; while ((1u << ls) < nseg) ls++; }
    const uint32_t spw = 1u << ls, cpw = 64u >> ls;
    const uint32_t groups = (nseg + spw - 1u) / spw;
    const dim3 grid(((p->C + cpw - 1u) / cpw) * groups), block(128);
    if (tm) {
        p->tm_calls++;
        const dim3 gtm(ncg * ((nseg + nsub - 1) / nsub)), btm(128);
        if (fm) hipLaunchKernelGGL((k_agc_spec_tm<true>), gtm, btm, 0, s, A, ncg, Cw, nsub);
        else hipLaunchKernelGGL((k_agc_spec_tm<false>), gtm, btm, 0, s, A, ncg, Cw, nsub);
    } else if (pairs) {
        if (fm) hipLaunchKernelGGL((k_agc_spec<true, true>), grid, block, 0, s, A, groups, ls);
        else hipLaunchKernelGGL((k_agc_spec<false, true>), grid, block, 0, s, A, groups, ls);
    } else {
        if (fm) hipLaunchKernelGGL((k_agc_spec<true, false>), grid, block, 0, s, A, groups, ls);
        else hipLaunchKernelGGL((k_agc_spec<false, false>), grid, block, 0, s, A, groups, ls);
    }
    // the fix-up reads st_in through the segment records only, so st can be overwritten in place
    if (fm) hipLaunchKernelGGL(k_agc_fix<true>, dim3(p->C), dim3(256), 0, s, A, st, rp_out, p->d_stats);
    else hipLaunchKernelGGL(k_agc_fix<false>, dim3(p->C), dim3(256), 0, s, A, st, rp_out, p->d_stats);
    CSDR_HIP(hipGetLastError());
    return 0;
}

}  // namespace csdr
