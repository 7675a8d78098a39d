// Single-pass fused channelizer kernel for gfx950 (wave64, 256 CUs, 160 KiB LDS/CU), M = 256.
//
//   raw CF32 x  --DC blocker--> y --14-tap polyphase FIR + NCO pre-mix--> X_t[j]
//               --256-point forward DFT (16 x 16)--> Y_t[k] --transpose--> channel-major
//               --per-channel freqdem--> out[C][nf]          (8 B read + 4/8 B written per sample)
//
// One workgroup (256 threads) produces one TILE of 16 frames (4096 input samples) and never
// writes an intermediate to HBM.  The stream-long recurrences are cut at tile boundaries:
//   * DC blocker (iirfilt_crcf, v[n] = x[n] + beta v[n-1], y = x - alpha v[n-1]) is a linear
//     scan.  Each tile's zero-state aggregate A_b is published as two 8-byte {tag,value}
//     granules as soon as the tile has been read; a tile's carry is the decayed sum of the
//     previous ten aggregates (beta^40960 = 1.3e-9 truncation) -- a decoupled look-back that
//     never chains, because an aggregate does not depend on any other workgroup.
//   * The FIR needs the 13 frames before the tile: the workgroup re-reads the previous tile
//     (an L2 hit: its owner is reading it at the same time) and re-runs the DC blocker on it.
//   * freqdem needs the previous frame's channel sample: each tile publishes its last Y
//     frame (2 KiB) and the successor picks it up in its tail phase.
// Workgroups take their tile index from an atomic ticket, so every tile a workgroup waits on
// belongs to a workgroup that has already started and publishes before it waits (no deadlock
// under any dispatch order; inter-workgroup data uses agent-scope atomics only).
//
// Thread roles inside a tile (tid = 0..255):
//   stage    thread q = run of 16 consecutive samples: serial DC scan, 16-lane DPP row scan
//   FIR      thread j = polyphase branch j           : window of 13 + 16 frames in VGPRs
//   pass 1   thread (f, b)  : Z_f[k1][b] = W256^(b k1) sum_a X_f[16a+b] W16^(a k1)
//   pass 2   thread (f, k1) : Y_f[k1+16 k2] = sum_b Z_f[k1][b] W16^(b k2)
//   tail     thread k       : 16 consecutive time samples of channel k
// MFMA is not used: the path is streaming FIR/FFT bounded by HBM and VALU (north_star).
//
// Replaces per chunk: iirfilt_crcf_execute_block + nco_crcf_mix_block_down + nf x
// firpfbch_crcf_analyzer_execute + the Haskell transpose (Liquid.chs:575-589, 828-862) and
// M x freqdem_demodulate_block (Liquid.chs:324-328).
#include "fused_common.h"
#include <cstring>

namespace csdr {

namespace {

#define STAMP(i) do { if (A.trace && tid == 0) A.trace[(size_t)b * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)

template <bool FM>
__global__ __launch_bounds__(256) void k_tile256(TileArgs A)
{
    __shared__ __attribute__((aligned(16))) float2 R[LDS_F2];
    __shared__ float2 tw_s[M256];
    __shared__ float2 Tt[2][16];        // frame totals: [0] own, [1] halo
    __shared__ float2 carry_s[2];       // c_b, c_{b-1}
    __shared__ unsigned tile_s;

    const int tid = threadIdx.x;
    if (tid == 0) tile_s = atomicAdd(A.ticket, 1u);
    tw_s[tid] = A.tw[tid];
    __syncthreads();
    const unsigned b = tile_s;
    if (b >= A.nb) return;
    STAMP(0);   // ticket known
    const int nvalid = (int)min(16u, A.nf - 16u * b);
    const int j = tid;
    const int col_off = 16 * (j >> 4) + 2 * (((j & 15) >> 1) ^ (j >> 5)) + (j & 1);
    float2 *E = R + E_OFF;

    // ---------------- issue the global loads: own tile and the tile before it ----------------
    float4 raw_o[8], raw_h[8];
    tile_load(reinterpret_cast<const float4 *>(A.x) + (size_t)b * 2048, 16 * nvalid, raw_o, tid);
    if (b > 0) tile_load(reinterpret_cast<const float4 *>(A.x) + (size_t)(b - 1) * 2048, 256, raw_h, tid);

    // ---------------- own tile: stage, scan, publish the aggregate ----------------
    stage_and_scan(raw_o, R, E, Tt[0], A, tid);
    STAMP(1);   // own tile loaded + scanned

    float2 nw[NB], old[NB];
#pragma unroll
    for (int f = 0; f < NB; f++) { nw[f] = R[256 * f + col_off]; old[f] = make_float2(0.f, 0.f); }

    if (tid < 64) {
        // zero-state frame chain of the own tile -> aggregate A_b, published as two granules
        if (tid == 0) {
            float2 v = make_float2(0.f, 0.f);
#pragma unroll
            for (int f = 0; f < 16; f++) v = cfma(v, A.b256[1], Tt[0][f]);
            __hip_atomic_store(&A.agg[2 * (size_t)b], ((u64)A.epoch << 32) | __float_as_uint(v.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&A.agg[2 * (size_t)b + 1], ((u64)A.epoch << 32) | __float_as_uint(v.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // ---------------- look-back: carries c_b (own tile) and c_{b-1} (halo tile) ----------------
        const int lane = tid;
        float2 cb = make_float2(0.f, 0.f), cp = make_float2(0.f, 0.f);
        const int k = lane;                                  // lane k in 1..LOOKBACK looks at tile b-k
        const bool need = (k >= 1 && k <= LOOKBACK && (int)b - k >= 0);
        u64 g0 = 0, g1 = 0;
        unsigned spins = 0;
        bool ok = !need;
        while (true) {
            if (need && !ok) {
                g0 = __hip_atomic_load(&A.agg[2 * (size_t)(b - k)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                g1 = __hip_atomic_load(&A.agg[2 * (size_t)(b - k) + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = (unsigned)(g0 >> 32) == A.epoch && (unsigned)(g1 >> 32) == A.epoch;
            }
            if (__all(ok)) break;
            if (++spins > SPIN_LIMIT) { if (lane == 0) atomicOr(A.status, 1u); break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (need) {
            const float2 Ak = make_float2(__uint_as_float((unsigned)g0), __uint_as_float((unsigned)g1));
            cb = make_float2(Ak.x * A.wtile[k - 1], Ak.y * A.wtile[k - 1]);
            if (k >= 2) cp = make_float2(Ak.x * A.wtile[k - 2], Ak.y * A.wtile[k - 2]);
        }
        if (lane == 0) {                                     // the stream state before this call
            const float2 ve = A.vend_in[0];
            if (b <= LOOKBACK) cb = make_float2(ve.x * A.wtile[b], ve.y * A.wtile[b]);
            if (b >= 1 && b - 1 <= LOOKBACK) cp = make_float2(ve.x * A.wtile[b - 1], ve.y * A.wtile[b - 1]);
        }
        // sum over lanes 0..15 (one DPP row): the inclusive row_shr ladder ends in lane 15
        cb = cadd(cb, dpp2<0x111>(cb)); cp = cadd(cp, dpp2<0x111>(cp));
        cb = cadd(cb, dpp2<0x112>(cb)); cp = cadd(cp, dpp2<0x112>(cp));
        cb = cadd(cb, dpp2<0x114>(cb)); cp = cadd(cp, dpp2<0x114>(cp));
        cb = cadd(cb, dpp2<0x118>(cb)); cp = cadd(cp, dpp2<0x118>(cp));
        if (lane == 15) { carry_s[0] = cb; carry_s[1] = cp; }
    }
    STAMP(2);   // look-back done (wave 0)
    __syncthreads();                                            // own z consumed; carries ready
    STAMP(3);

    // ---------------- halo: the 13 frames before the tile ----------------
    if (b > 0) {
        stage_and_scan(raw_h, R, E + 256, Tt[1], A, tid);
#pragma unroll
        for (int f = 3; f < NB; f++) old[f] = R[256 * f + col_off];
    } else {
#pragma unroll
        for (int f = 3; f < NB; f++) old[f] = A.yhist_in[(f - 3) * M256 + j];
    }

    STAMP(4);   // halo staged + scanned
    // ---------------- carry into every run: P[q] = beta^(16 r) * (v before frame f) + E[q] ----------------
    {
        const int fq = tid >> 4;
        const float br = A.b16[tid & 15], bf = A.b256[fq];
        const float2 v0 = cfma(carry_s[0], bf, frame_carry_zero_state(Tt[0], A, tid));
        E[tid] = cfma(v0, br, E[tid]);
        if (b > 0) {
            const float2 v1 = cfma(carry_s[1], bf, frame_carry_zero_state(Tt[1], A, tid));
            E[256 + tid] = cfma(v1, br, E[256 + tid]);
        }
        if (b == A.nb - 1 && tid == 16 * ((nvalid - 1) & 15) + 15) {
            // DC blocker state after the last valid frame: v before it, advanced over that frame
            // (the thread owning the frame's last run: v0 is v before frame nvalid-1)
            A.vend_out[0] = cfma(v0, A.b256[1], Tt[0][nvalid - 1]);
        }
    }
    __syncthreads();

    // ---------------- finish the DC blocker: y = z - alpha*beta^i * P[run] ----------------
    {
        const float kj = -A.alpha * A.bj[j & 15];
#pragma unroll
        for (int f = 0; f < NB; f++) nw[f] = cfma(E[16 * f + (j >> 4)], kj, nw[f]);
        if (b > 0) {
#pragma unroll
            for (int f = 3; f < NB; f++) old[f] = cfma(E[256 + 16 * f + (j >> 4)], kj, old[f]);
        }
    }
    if (b == A.nb - 1) {
        // stream state for the next call: DC v1 after the last frame, the last 13 frames of y
        const int base = (int)A.nf - 13 - 16 * (int)b;        // tile-relative frame of yhist slot 0
#pragma unroll
        for (int f = 3; f < NB; f++) {
            const int d = (f - 16) - base;
            if (d >= 0 && d < 13) A.yhist_out[d * M256 + j] = old[f];
        }
#pragma unroll
        for (int f = 0; f < NB; f++) {
            const int d = f - base;
            if (d >= 0 && d < 13 && f < nvalid) A.yhist_out[d * M256 + j] = nw[f];
        }
    }
    __syncthreads();                                            // P consumed, R free
    STAMP(5);   // DC blocker finished

    // ---------------- polyphase FIR + NCO pre-mix ----------------
    // u[t][j] = y[t][j] * wpre[parity(t)][j] and X_t[j] = sum_n h[(255-j)+256n] u[t-n][j]:
    // even and odd taps see the two phasors, so two partial sums and two complex products.
    {
        float h[P];
#pragma unroll
        for (int n = 0; n < P; n++) h[n] = A.taps[(M256 - 1 - j) + n * M256];
        const float2 Wa = A.wpre[(A.parity0 & 1) * M256 + j], Wb = A.wpre[((A.parity0 & 1) ^ 1) * M256 + j];
        const v2f Wav = to_v(Wa), Wbv = to_v(Wb);
#pragma unroll
        for (int f = 0; f < NB; f += 2) {
            // complex x real multiply-accumulate as one packed f32 FMA per tap; two frames at a time so that
            // consecutive FMAs are independent
            v2f ev[2] = {{0.f, 0.f}, {0.f, 0.f}}, od[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
            for (int n = P - 1; n >= 0; n--) {
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int i = f + q - n;
                    const float2 s2 = (i >= 0) ? nw[i] : old[NB + i];
                    const v2f sv = {s2.x, s2.y}, hv = {h[n], h[n]};
                    if (n & 1) od[q] = __builtin_elementwise_fma(sv, hv, od[q]); else ev[q] = __builtin_elementwise_fma(sv, hv, ev[q]);
                }
            }
            // frame f is even within the tile: its even taps see Wa, its odd taps Wb; frame f+1 the other way
            cmul2_v(ev[0], Wav, od[0], Wbv);
            cmul2_v(ev[1], Wbv, od[1], Wav);
            R[f * FS_X + j] = to_f2(ev[0] + od[0]);
            R[(f + 1) * FS_X + j] = to_f2(ev[1] + od[1]);
        }
    }
    __syncthreads();                                            // X complete
    STAMP(6);   // FIR done

    // ---------------- DFT pass 1: thread (f, b1) ----------------
        v2f vv[16];
        {
            const int f = tid >> 4, b1 = tid & 15;
#pragma unroll
            for (int a = 0; a < 16; a++) vv[a] = to_v(R[f * FS_X + 16 * a + b1]);
            fft16_v(vv);
#pragma unroll
            for (int i = 1; i < 16; i++) vv[i] = cmul_v(vv[i], to_v(tw_s[16 * XIDX(i) + b1]));
            __syncthreads();                                        // everyone has read X
#pragma unroll
            for (int i = 0; i < 16; i++) R[f * FS_Z + XIDX(i) * RS_Z + b1] = to_f2(vv[i]);
        }
        __syncthreads();                                            // Z complete
    STAMP(7);   // pass 1 done

    // ---------------- DFT pass 2: thread (f, k1) ----------------
        {
            const int f = tid >> 4, k1 = tid & 15;
#pragma unroll
            for (int b1 = 0; b1 < 16; b1++) vv[b1] = to_v(R[f * FS_Z + k1 * RS_Z + b1]);
            fft16_v(vv);
            __syncthreads();                                        // everyone has read Z
#pragma unroll
            for (int i = 0; i < 16; i++) R[(k1 + 16 * XIDX(i)) * RS_Y + f] = to_f2(vv[i]);
        }
        __syncthreads();                                            // Y complete
    STAMP(8);   // pass 2 done

    // ---------------- tail: thread k owns channel k ----------------
    float2 v[16];
#pragma unroll
    for (int f = 0; f < NB; f++) v[f] = R[tid * RS_Y + f];
    const bool owned = (uint32_t)tid >= A.c0 && (uint32_t)tid < A.c0 + A.C;
    const size_t row = (size_t)(owned ? tid - A.c0 : 0) * A.out_stride + A.out_t0 + (size_t)16 * b;

    if (FM) {
        // publish my last valid frame for the next tile, then fetch the previous tile's
        float2 last = v[0];
#pragma unroll
        for (int f = 1; f < NB; f++) if (f < nvalid) last = v[f];
        const u64 bits = ((u64)__float_as_uint(last.y) << 32) | __float_as_uint(last.x);
        __hip_atomic_store(&A.ylast[(size_t)b * M256 + tid], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&A.yflag[b], A.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (b == A.nb - 1 && owned) A.rp_out[tid - A.c0] = last;
        STAMP(9);   // last frame published

        float m[NB];
        if (owned) {
#pragma unroll
            for (int f = 1; f < NB; f++) {
                const float2 rp = v[f - 1], r = v[f];
                // arg(conj(r') r): products rounded separately like the reference's C expression
                const float re = __fadd_rn(__fmul_rn(rp.x, r.x), __fmul_rn(rp.y, r.y));
                const float im = __fsub_rn(__fmul_rn(rp.x, r.y), __fmul_rn(rp.y, r.x));
                m[f] = fast_atan2f(im, re) * A.fm_ref;
            }
        }
        float2 prev;
        if (b == 0) {
            prev = owned ? A.rp_in[tid - A.c0] : make_float2(0.f, 0.f);
        } else {
            unsigned spins = 0;
            while (__hip_atomic_load(&A.yflag[b - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != A.epoch) {
                if (++spins > SPIN_LIMIT) { if (tid == 0) atomicOr(A.status, 2u); break; }
                __builtin_amdgcn_s_sleep(2);
            }
            const u64 pb = __hip_atomic_load(&A.ylast[(size_t)(b - 1) * M256 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            prev = make_float2(__uint_as_float((unsigned)pb), __uint_as_float((unsigned)(pb >> 32)));
        }
        STAMP(10);  // previous tile's last frame received
        if (owned) {
            {
                const float2 r = v[0];
                const float re = __fadd_rn(__fmul_rn(prev.x, r.x), __fmul_rn(prev.y, r.y));
                const float im = __fsub_rn(__fmul_rn(prev.x, r.y), __fmul_rn(prev.y, r.x));
                m[0] = fast_atan2f(im, re) * A.fm_ref;
            }
            float *o = (float *)A.out + row;
            if (((A.out_stride | A.out_t0) % 4u) == 0 && nvalid == NB) {
#pragma unroll
                for (int q = 0; q < 4; q++)
                    *reinterpret_cast<float4 *>(o + 4 * q) = make_float4(m[4 * q], m[4 * q + 1], m[4 * q + 2], m[4 * q + 3]);
            } else {
#pragma unroll
                for (int f = 0; f < NB; f++) if (f < nvalid) o[f] = m[f];
            }
        }
    } else if (owned) {
        float2 *o = (float2 *)A.out + row;
        if (((A.out_stride | A.out_t0) % 2u) == 0 && nvalid == NB) {
#pragma unroll
            for (int q = 0; q < 8; q++)
                *reinterpret_cast<float4 *>(o + 2 * q) = make_float4(v[2 * q].x, v[2 * q].y, v[2 * q + 1].x, v[2 * q + 1].y);
        } else {
#pragma unroll
            for (int f = 0; f < NB; f++) if (f < nvalid) o[f] = v[f];
        }
    }
    STAMP(11);  // stores issued
}


}  // namespace

// Split nb tiles into nruns runs.  With one run per resident workgroup slot (nruns = slots x cus) slot k's runs
// get a share proportional to weight[k]; otherwise the split is uniform.  Every run keeps >= 8 tiles.
static RunSplit make_split(uint32_t nb, uint32_t nruns, uint32_t cus, const float *weight)
{
    RunSplit sp{};
    const uint32_t slots = (cus && nruns % cus == 0) ? nruns / cus : 0;
    if (slots >= 2 && slots <= 8) {
        double tot = 0;
        for (uint32_t k = 0; k < slots; k++) tot += weight[k];
        sp.rps = cus; sp.nslots = slots;
        uint32_t base = 0;
        bool ok = true;
        for (uint32_t k = 0; k < slots; k++) {
            double cum = 0;
            for (uint32_t q = 0; q <= k; q++) cum += weight[q];
            const uint32_t end = (k + 1 == slots) ? nb : (uint32_t)std::llround((double)nb * cum / tot);
            sp.base[k] = base; sp.tiles[k] = end - base;
            if (end < base || sp.tiles[k] < 8 * cus) ok = false;
            base = end;
        }
        if (ok) return sp;
    }
    sp = RunSplit{};
    sp.rps = nruns; sp.nslots = 1; sp.base[0] = 0; sp.tiles[0] = nb;
    return sp;
}

// The chunk's last WU + 1 raw tiles -> the plan's tail slot (a kernel on the launch's own stream rather than a D2D memcpy: no
// copy engine, no blit-path synchronisation in front of the run kernel)
__global__ __launch_bounds__(256) void k_save_tail(const float4 *__restrict__ src, float4 *__restrict__ dst)
{
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    dst[i] = src[i];
}

// ragged end of an interleaved shard's call: rows c0 + G m of the whole-band tile result [256][rem] -> the shard's plane [C][nf] at frame t0
// (w = floats per sample: 1 F32, 2 CF32)
__global__ __launch_bounds__(64) void k_shard_gather(const float *__restrict__ src, float *__restrict__ dst, uint32_t c0, uint32_t G, uint32_t rem,
                                                     uint32_t nf, uint32_t t0, uint32_t w)
{
    const uint32_t m = blockIdx.x;
    for (uint32_t i = threadIdx.x; i < rem * w; i += 64) dst[((size_t)m * nf + t0) * w + i] = src[(size_t)(c0 + G * m) * rem * w + i];
}

struct FusedPlan {
    FusedConfig cfg;
    std::string name;
    uint32_t max_nb = 0, epoch = 0;
    uint64_t frames_done = 0;    // global frame counter (parity selects the premix phasor row)
    float *d_taps = nullptr;
    float2 *d_tw = nullptr, *d_wpre = nullptr;
    float2 *d_yhist[2] = {nullptr, nullptr}, *d_vend[2] = {nullptr, nullptr}, *d_rp[2] = {nullptr, nullptr};
    int cur = 0;
    unsigned *d_ticket = nullptr, *d_yflag = nullptr, *d_status = nullptr;
    u64 *d_agg = nullptr, *d_ylast = nullptr;
    void *d_premix = nullptr;    // per-channel output before mixing
    // whole-band run-kernel launches without warm-up windows (RunArgs::nowu, round 5): DC state in front of every run's halo tile, the
    // uncorrected near-DC channels of every run's first tiles, the chain's response to a unit DC state (k_run256_dcfix)
    float2 *d_cpre = nullptr, *d_side = nullptr, *d_rt = nullptr;
    bool nowu = false;
    u64 *d_trace = nullptr;
    uint32_t run_min_tiles = 1024;   // chunks with at least this many tiles use the run kernel (measured crossover: FM ~900 tiles, CF32 ~700; profiles/r04_call_size_sweeps.txt)
    uint32_t cus = 256;
    float slot_weight[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};   // tile share of the k-th co-resident run of a CU
    uint32_t resident_wgs_v2 = 512;  // workgroups of k_run256v2 the device holds at once
    bool use_v3 = false;             // variant builds only (CSDR_WITH_RUN256_V3 + CSDR_RUN_V3=1): k_run256v3, round 4's one-workgroup-per-CU experiment (tools/variants/)
    // independent launches (csdr_chain_submit_device): the last WU + 1 raw tiles of the previous chunk, three slots (the launch
    // two calls back may still be reading its slot when this call's copy is queued on the other stream)
    float *d_shard_tail = nullptr;   // interleaved shard: whole-band result of a call's ragged end, [256][< 16] CF32 at most
    float4 *d_tail[3] = {nullptr, nullptr, nullptr};
    int tail_w = 0;                  // slot that holds the tail of the most recent chunk
    bool tail_valid = false, keep_tail = false;
    TileArgs proto;
};

bool fused_supported(uint32_t M, uint32_t p) { return M == 256 && p == P; }

void dc_state_response(const FusedConfig &cfg, const float2 *wpre, uint32_t f0, uint32_t nfr, uint32_t k0, float2 *rt)
{
    // the blocker's output carries -alpha beta^n of the state at sample n behind the tile's start; that goes through the pre-mix, the
    // polyphase FIR and the DFT like any input
    const uint32_t M = cfg.M, T = f0 + nfr;
    const double beta = (double)cfg.dc.beta, alpha = 1.0 - beta, tp = -2.0 * 3.14159265358979323846;
    std::vector<double> ur((size_t)T * M), ui((size_t)T * M), xr(M), xi(M);
    for (uint32_t par = 0; par < 2; par++) {
        for (uint32_t f = 0; f < T; f++)
            for (uint32_t j = 0; j < M; j++) {
                const double e = -alpha * std::pow(beta, (double)f * M + j);
                const float2 wv = wpre[((par + f) & 1u) * M + j];
                ur[(size_t)f * M + j] = e * (double)wv.x; ui[(size_t)f * M + j] = e * (double)wv.y;
            }
        for (uint32_t t = 0; t < nfr; t++) {
            const uint32_t f = f0 + t;
            for (uint32_t j = 0; j < M; j++) {
                double ar = 0.0, ai = 0.0;
                for (uint32_t n = 0; n < cfg.p && n <= f; n++) {
                    const double hh = (double)cfg.taps[(M - 1 - j) + n * M];
                    ar += hh * ur[(size_t)(f - n) * M + j]; ai += hh * ui[(size_t)(f - n) * M + j];
                }
                xr[j] = ar; xi[j] = ai;
            }
            for (uint32_t ch = 0; ch < 4; ch++) {
                const uint32_t k = k0 + ch;
                double yr = 0.0, yi = 0.0;
                for (uint32_t j = 0; j < M; j++) {
                    const double a = tp * (double)((j * k) % M) / (double)M, cr = std::cos(a), ci = std::sin(a);
                    yr += xr[j] * cr - xi[j] * ci; yi += xr[j] * ci + xi[j] * cr;
                }
                rt[((size_t)par * nfr + t) * 4 + ch] = make_float2((float)yr, (float)yi);
            }
        }
    }
}

int fused_create(const FusedConfig &cfg, FusedPlan **out)
{
    FusedPlan *p = new FusedPlan();
    p->cfg = cfg;
    p->name = cfg.fm ? "k_tile256<FM>" : "k_tile256<CF32>";
    p->max_nb = (cfg.max_nf + NB - 1) / NB;
    auto fail = [&](int r) { fused_destroy(p); return r; };
#define ALLOC(ptr, bytes) do { hipError_t e = hipMalloc((void **)&(ptr), (bytes) ? (bytes) : 1); if (e != hipSuccess) return fail(hip_fail(e, "hipMalloc", __FILE__, __LINE__)); } while (0)
    ALLOC(p->d_taps, sizeof(float) * cfg.M * cfg.p);
    ALLOC(p->d_tw, sizeof(float2) * cfg.M);
    ALLOC(p->d_wpre, sizeof(float2) * 2 * cfg.M);
    for (int i = 0; i < 2; i++) {
        ALLOC(p->d_yhist[i], sizeof(float2) * 13 * cfg.M);
        ALLOC(p->d_vend[i], sizeof(float2));
        ALLOC(p->d_rp[i], sizeof(float2) * (cfg.G > 1 ? cfg.M : cfg.C));     // interleaved shard: full-band indices (fused_v2: rp_in / rp_out)
    }
    if (cfg.G > 1) ALLOC(p->d_shard_tail, sizeof(float2) * cfg.M * NB);
    for (int i = 0; i < 3; i++) { ALLOC(p->d_tail[i], sizeof(float4) * 2048 * (WU + 1)); CSDR_HIP(hipMemset(p->d_tail[i], 0, sizeof(float4) * 2048 * (WU + 1))); }
    p->tail_valid = true;            // a fresh stream: zero history IS the exact history
    ALLOC(p->d_ticket, sizeof(unsigned));
    ALLOC(p->d_status, sizeof(unsigned));
    ALLOC(p->d_yflag, sizeof(unsigned) * p->max_nb);
    ALLOC(p->d_agg, sizeof(u64) * 2 * p->max_nb);
    ALLOC(p->d_ylast, sizeof(u64) * (size_t)cfg.M * p->max_nb);
    if (const char *e = diag_env("CSDR_RUN_MIN_TILES")) p->run_min_tiles = (uint32_t)atol(e);
    if (cfg.mix) ALLOC(p->d_premix, (size_t)cfg.C * cfg.max_nf * (cfg.fm ? 4 : 8));
    if (diag_env("CSDR_TRACE")) { ALLOC(p->d_trace, sizeof(u64) * 16 * p->max_nb); CSDR_HIP(hipMemset(p->d_trace, 0, sizeof(u64) * 16 * p->max_nb)); }
    p->nowu = cfg.G == 1 && cfg.c0 == 0 && cfg.C == cfg.M && cfg.dc_block && !(diag_env("CSDR_NOWU") && atoi(diag_env("CSDR_NOWU")) == 0);
    if (p->nowu) {
        ALLOC(p->d_cpre, sizeof(float2) * 2050);
        ALLOC(p->d_side, sizeof(float2) * 2048 * 4 * DCFIX_F);
        ALLOC(p->d_rt, sizeof(float2) * 2 * DCFIX_F * 4);
        CSDR_HIP(hipMemset(p->d_cpre, 0, sizeof(float2) * 2050));
    }
#undef ALLOC
    CSDR_HIP(hipMemcpy(p->d_taps, cfg.taps, sizeof(float) * cfg.M * cfg.p, hipMemcpyHostToDevice));
    std::vector<float2> tw(cfg.M), wpre(2 * cfg.M);
    for (uint32_t k1 = 0; k1 < 16; k1++)
        for (uint32_t b1 = 0; b1 < 16; b1++) {
            const double a = -2.0 * 3.14159265358979323846 * (double)(b1 * k1) / (double)cfg.M;
            tw[16 * k1 + b1] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
    // nco_crcf_mix_block_down multiplies by conj(cos + j sin) of theta = n*d_theta; for M = 256
    // the phase sequence has period 2M: row 0 = even frames (n = j), row 1 = odd (n = M + j)
    for (uint32_t i = 0; i < 2 * cfg.M; i++) {
        float c, s;
        nco_phasor(i * cfg.d_theta, &c, &s);
        wpre[i] = make_float2(c, -s);
    }
    CSDR_HIP(hipMemcpy(p->d_tw, tw.data(), sizeof(float2) * cfg.M, hipMemcpyHostToDevice));
    CSDR_HIP(hipMemcpy(p->d_wpre, wpre.data(), sizeof(float2) * 2 * cfg.M, hipMemcpyHostToDevice));
    if (p->nowu) {
        // Response of the chain, at the channels 126..129, to a DC-blocker state of 1 in front of a tile: the blocker's output carries
        // -alpha beta^n of it at sample n, which goes through the pre-mix (the table above), the polyphase FIR and the DFT like any
        // input.  Frames 15 .. 127 behind the tile's start = frame -1 .. 111 of the run that started its halo tile without its state;
        // by then the step's edge (13 frames of FIR, every channel) has left the filter and what remains sits around DC.  Both parities of
        // the tile's first frame; f64 throughout.
        std::vector<float2> rt((size_t)2 * DCFIX_F * 4);
        dc_state_response(cfg, wpre.data(), 15u, (uint32_t)DCFIX_F, 126u, rt.data());
        CSDR_HIP(hipMemcpy(p->d_rt, rt.data(), sizeof(float2) * rt.size(), hipMemcpyHostToDevice));
    }
    CSDR_HIP(hipMemset(p->d_status, 0, sizeof(unsigned)));
    CSDR_HIP(hipMemset(p->d_yflag, 0, sizeof(unsigned) * p->max_nb));
    CSDR_HIP(hipMemset(p->d_agg, 0, sizeof(u64) * 2 * p->max_nb));

    TileArgs &A = p->proto;
    A = TileArgs{};
    A.taps = p->d_taps; A.tw = p->d_tw; A.wpre = p->d_wpre;
    A.ticket = p->d_ticket; A.agg = p->d_agg; A.ylast = p->d_ylast; A.yflag = p->d_yflag; A.status = p->d_status;
    A.c0 = cfg.c0; A.C = cfg.C; A.fm_ref = cfg.fm_ref; A.trace = p->d_trace;
    const double beta = cfg.dc_block ? (double)cfg.dc.beta : 0.0;
    A.alpha = cfg.dc_block ? (float)(1.0 - beta) : 0.0f;    // alpha = 1 - beta with beta = f32(1 - 0.0005)
    A.beta = (float)beta;
    for (int k = 0; k < LOOKBACK + 2; k++) A.wtile[k] = (float)std::pow(beta, 4096.0 * k);
    for (int k = 0; k < 16; k++) A.b16[k] = (float)std::pow(beta, 16.0 * k);
    for (int k = 0; k < 17; k++) A.b256[k] = (float)std::pow(beta, 256.0 * k);
    for (int k = 0; k < 16; k++) A.bj[k] = (float)std::pow(beta, (double)k);
    {
        int dev = 0, cus = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (const char *e = diag_env("CSDR_CUS")) { if (atoi(e) > 0 && atoi(e) < cus) cus = atoi(e); }      // experiments: a plan sized for a CU-masked stream
        p->cus = (uint32_t)cus;
        p->resident_wgs_v2 = (uint32_t)(cus * run256_v2_blocks_per_cu(cfg.fm));
        if (const char *e = diag_env("CSDR_RESIDENT_WGS")) p->resident_wgs_v2 = (uint32_t)atol(e);
#ifdef CSDR_WITH_RUN256_V3          // variant build only (tools/variants/build_run256_v3.sh): round 4's one-workgroup-per-CU experiment
        if (const char *e = diag_env("CSDR_RUN_V3")) p->use_v3 = atoi(e) != 0;
#endif
        if (const char *e = diag_env("CSDR_RUN_WEIGHTS")) {
            int k = 0;
            for (const char *q = e; *q && k < 8; k++) { p->slot_weight[k] = (float)atof(q); q = strchr(q, ','); if (!q) break; q++; }
        }
    }
    *out = p;
    return 0;
}

int fused_reset(FusedPlan *p, hipStream_t s)
{
    p->cur = 0; p->frames_done = 0;
    for (int i = 0; i < 2; i++) {
        CSDR_HIP(hipMemsetAsync(p->d_yhist[i], 0, sizeof(float2) * 13 * p->cfg.M, s));
        CSDR_HIP(hipMemsetAsync(p->d_vend[i], 0, sizeof(float2), s));
        CSDR_HIP(hipMemsetAsync(p->d_rp[i], 0, sizeof(float2) * (p->cfg.G > 1 ? p->cfg.M : p->cfg.C), s));
    }
    CSDR_HIP(hipMemsetAsync(p->d_tail[0], 0, sizeof(float4) * 2048 * (WU + 1), s));
    p->tail_w = 0; p->tail_valid = true;
    return 0;
}

void fused_keep_tail(FusedPlan *p) { p->keep_tail = p->cfg.G == 1; }     // interleaved shards never run as independent launches
bool fused_tail_recorded(const FusedPlan *p) { return p->keep_tail && p->tail_valid; }

static bool fused_v2_call(const FusedPlan *p, uint32_t nf)
{
    const FusedConfig &c = p->cfg;
    return c.G == 1 && c.c0 == 0 && c.C == c.M && nf / NB >= p->run_min_tiles;
}

bool fused_tile_major_ok(const FusedPlan *p, uint32_t nf)
{
    // the calls that go to k_run256v2 / v3 as whole tiles: CF32 output below 4 GiB, no mix inside the plan
    const FusedConfig &c = p->cfg;
    const bool shard = c.G > 1;
    if (c.fm || c.mix || nf % NB || !nf) return false;
    if (shard) return (uint64_t)c.C * nf * 8u < (1ull << 32);
    return c.c0 == 0 && c.C == c.M && nf / NB >= p->run_min_tiles;
}

bool fused_can_overlap(const FusedPlan *p, uint32_t nf)
{
    return p->keep_tail && p->tail_valid && fused_v2_call(p, nf) && nf % NB == 0 && !p->cfg.mix;
}

int fused_process(FusedPlan *p, const FusedCall &call, hipStream_t s, KernelTimer *timer)
{
    const FusedConfig &c = p->cfg;
    const uint32_t nf = call.nf;
    if (!nf) return 0;
    int r;
    TileArgs A = p->proto;
    A.x = call.d_in;
    A.out = c.mix ? p->d_premix : call.d_out;
    A.yhist_in = p->d_yhist[p->cur]; A.yhist_out = p->d_yhist[p->cur ^ 1];
    A.vend_in = p->d_vend[p->cur];   A.vend_out = p->d_vend[p->cur ^ 1];
    A.rp_in = p->d_rp[p->cur];       A.rp_out = p->d_rp[p->cur ^ 1];
    A.out_stride = nf; A.out_t0 = 0;
    if (call.tile_major) {
        if (!fused_tile_major_ok(p, nf)) { set_error("fused: internal: tile-major output requested from a call that cannot write it"); return -1; }
        A.out_stride = NB;                   // a row's 16 frames of a tile are one 128-byte line; rows follow each other inside the tile's block
    }
    A.parity0 = (uint32_t)(p->frames_done & 1);
    const uint32_t nb_full = nf / NB;
    const bool shard = c.G > 1;      // interleaved shard: every whole tile goes through k_run256v2<.., G> (the tile kernel only knows whole bands)
    if (shard && (uint64_t)c.C * nf * (c.fm ? 4u : 8u) >= (1ull << 32)) { set_error("fused: interleaved shard output of %u frames exceeds 4 GiB", nf); return -1; }
    const bool whole_band = c.G == 1 && c.c0 == 0 && c.C == c.M;      // (a contiguous channel shard takes the tile kernel, which masks its stores, at every size)
    if ((nb_full >= p->run_min_tiles && whole_band) || shard) {
        // large chunk: dependency-free runs of full tiles (k_run256v2; its lane offsets are 32-bit over 16 rows and the row groups
        // go through a 64-bit base, so the output may exceed 4 GiB); a ragged tail (< 16 frames) follows as a second launch of the
        // tile kernel on the state the run kernel leaves behind
        const bool v2 = true;
        // third generation: one 512-thread workgroup per CU (half the cold starts), whole band
        const bool v3 = !shard && p->use_v3;
        p->name = v3 ? (c.fm ? "k_run256v3<FM>" : "k_run256v3<CF32>") : (c.fm ? "k_run256v2<FM>" : "k_run256v2<CF32>");
        if (shard) p->name += "/G" + std::to_string(c.G);
        RunArgs RA{};
        A.nf = nb_full * NB; A.nb = nb_full;
        RA.t = A; RA.pk = phase_consts(c.fm_ref);
        // one run per resident workgroup slot (a single round), runs balanced to within one tile,
        // at least 8 tiles per run so that the warm-up reads stay below 7/8 of a run
        uint32_t nruns = v3 ? p->cus : p->resident_wgs_v2;
        if (v3) { if (const char *e = diag_env("CSDR_V3_RUNS")) nruns = (uint32_t)atol(e); }
        if (nruns > A.nb / 8) nruns = A.nb / 8;
        if (nruns < 1) nruns = 1;
        RA.nruns = nruns;
        // k_run256v2: the older of a CU's two workgroups wins the issue arbitration and runs ~1.4x faster than the younger
        // one, so it gets the larger share of the tiles (measured: both end together at about 1.2 : 0.8); rotating the
        // priority per tile instead was no better
        static const float v2_weight_cf[8] = {1.2f, 0.8f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
        static const float v2_weight_fm[8] = {1.24f, 0.76f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};      // with whole-line stores the younger workgroup loses more (r03 trace: 4.8 vs 7.6 us per tile)
        const float *v2_weight = c.fm ? v2_weight_fm : v2_weight_cf;
        static const float equal_weight[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
        static const bool pd_equal = diag_env("CSDR_PD_EQUAL") != nullptr;
        RA.split = make_split(A.nb, nruns, p->cus, (call.indep && pd_equal) ? equal_weight : ((v2 && !diag_env("CSDR_RUN_WEIGHTS")) ? v2_weight : p->slot_weight));
        { const char *e = diag_env("CSDR_PRIO_ROT"); RA.prio_div = e ? (atoi(e) ? p->cus : 0u) : (v2 ? 0u : p->cus); }
        { const char *e = diag_env("CSDR_TRACE"); RA.trace_light = (e && atoi(e) == 2) ? 1u : 0u; }
        { const char *e = diag_env("CSDR_WU"); RA.wu = e ? (uint32_t)atoi(e) : (uint32_t)WU; }       // experiments: fewer tiles = wrong DC state at run starts
        { const char *e = diag_env("CSDR_WU_ROT"); RA.wu_rot = e ? (uint32_t)atoi(e) : 0u; }      // measured: no effect on the run-start burst
        RA.pair_align = (v2 && (c.fm || diag_env("CSDR_PAIR_ALIGN_CF"))) ? 1u : 0u;
        { const char *e = diag_env("CSDR_WU_BATCH6"); RA.wu_batch6 = e ? (uint32_t)atoi(e) : 1u; }
        RA.l2beta = c.dc_block ? (float)std::log2((double)c.dc.beta) : -1000.0f;
        RA.tile_step = call.tile_major ? c.C * 128u : (uint32_t)NB * (c.fm ? 4u : 8u);
        const bool whole = nf == nb_full * NB;
        RA.indep = (call.indep && v2 && whole && !c.mix && p->keep_tail && p->tail_valid) ? 1u : 0u;
        // whole band, dependent launches: no warm-up windows (the DC state a run misses at its start is put back, where it still matters 16 frames
        // later -- the four channels around DC -- by k_run256_dcfix behind the launch); runs of >= 8 tiles, <= 2048 of them
        RA.nowu = (p->nowu && !shard && !v3 && !RA.indep && nruns >= 2 && nruns <= 2048) ? 1u : 0u;
        RA.cpre = p->d_cpre; RA.side = p->d_side;
        RA.prev_tail = p->d_tail[p->tail_w];
        if (call.indep && !RA.indep) { set_error("fused: internal: independent launch requested from a call that cannot run as one"); return -1; }
        if (p->keep_tail && !shard && v2 && whole && nb_full >= WU + 1) {
            // this chunk's last WU + 1 tiles, for run 0 of the next call (queued in front of the launch: the copy only reads the input)
            const int nxt = (p->tail_w + 1) % 3;
            static const bool nocopy = diag_env("CSDR_PD_NOCOPY") != nullptr;      // scheduling experiments only (wrong run-0 starts)
            if (!nocopy) hipLaunchKernelGGL(k_save_tail, dim3(2048 * (WU + 1) / 256), dim3(256), 0, s,
                                            reinterpret_cast<const float4 *>(call.d_in + (size_t)(nb_full - (WU + 1)) * 4096), p->d_tail[nxt]);
            if (call.ev_tail) CSDR_HIP(hipEventRecord(call.ev_tail, s));
            p->tail_w = nxt; p->tail_valid = true;
        } else p->tail_valid = false;
        if (timer && (r = timer->begin(s))) return r;
        if (nb_full == 0) { /* shard, fewer than 16 frames: the tile kernel below does the whole call */ }
#ifdef CSDR_WITH_RUN256_V3
        else if (v3) { if ((r = run256_v3_launch(&RA, c.fm, nruns, s))) return r; }
#endif
        else if ((r = run256_v2_launch(&RA, c.fm, c.G, nruns, s))) return r;
        if (RA.nowu && nb_full && (r = run256_dcfix_launch(&RA, c.fm, nruns, p->d_rt, s))) return r;
        if (timer && (r = timer->end(s))) return r;             // the bracket covers k_run256_dcfix: it is part of every no-warm-up step
        const uint32_t rem = nf - nb_full * NB;
        if (rem) {
            if (nb_full) p->cur ^= 1;                            // the tail starts from the run kernel's state
            TileArgs T = p->proto;
            T.x = call.d_in + (size_t)nb_full * NB * c.M; T.out = A.out;
            if (shard) { T.c0 = 0; T.C = c.M; T.out = p->d_shard_tail; }   // whole band into [256][rem], the owned rows are gathered below
            T.yhist_in = p->d_yhist[p->cur]; T.yhist_out = p->d_yhist[p->cur ^ 1];
            T.vend_in = p->d_vend[p->cur];   T.vend_out = p->d_vend[p->cur ^ 1];
            T.rp_in = p->d_rp[p->cur];       T.rp_out = p->d_rp[p->cur ^ 1];
            if (++p->epoch == 0) p->epoch = 1;
            T.epoch = p->epoch; T.nf = rem; T.nb = 1; T.out_stride = nf; T.out_t0 = nb_full * NB;
            if (shard) { T.out_stride = rem; T.out_t0 = 0; }
            T.parity0 = (uint32_t)((p->frames_done + nb_full * NB) & 1);
            CSDR_HIP(hipMemsetAsync(p->d_ticket, 0, sizeof(unsigned), s));
            if (c.fm) hipLaunchKernelGGL(k_tile256<true>, dim3(1), dim3(256), 0, s, T);
            else hipLaunchKernelGGL(k_tile256<false>, dim3(1), dim3(256), 0, s, T);
            if (shard) hipLaunchKernelGGL(k_shard_gather, dim3(c.C), dim3(64), 0, s, (const float *)p->d_shard_tail, (float *)A.out, c.c0, c.G, rem,
                                          nf, nb_full * NB, c.fm ? 1u : 2u);
        }
    } else {
        p->name = c.fm ? "k_tile256<FM>" : "k_tile256<CF32>";
        p->tail_valid = false;
        if (call.indep) { set_error("fused: internal: independent launch requested from a tile-kernel call"); return -1; }
        if (++p->epoch == 0) p->epoch = 1;
        A.epoch = p->epoch; A.nf = nf; A.nb = (nf + NB - 1) / NB;
        CSDR_HIP(hipMemsetAsync(p->d_ticket, 0, sizeof(unsigned), s));
        if (timer && (r = timer->begin(s))) return r;
        if (c.fm) hipLaunchKernelGGL(k_tile256<true>, dim3(A.nb), dim3(256), 0, s, A);
        else hipLaunchKernelGGL(k_tile256<false>, dim3(A.nb), dim3(256), 0, s, A);
        if (timer && (r = timer->end(s))) return r;
    }
    CSDR_HIP(hipGetLastError());
    p->cur ^= 1;
    p->frames_done += nf;
    if (c.mix) {
        if ((r = launch_mix((const float *)p->d_premix, (float *)call.d_out, c.C, c.fm ? nf : 2 * nf, s))) return r;
    }
    return 0;
}

const char *fused_name(const FusedPlan *p) { return p->name.c_str(); }
void fused_seek(FusedPlan *p, uint64_t frames) { p->frames_done = frames; }

int fused_trace(FusedPlan *p, unsigned long long *out, uint32_t ntiles)
{
    if (!p->d_trace) return 0;
    if (ntiles > p->max_nb) ntiles = p->max_nb;
    CSDR_HIP(hipMemcpy(out, p->d_trace, sizeof(u64) * 16 * ntiles, hipMemcpyDeviceToHost));
    return (int)ntiles;
}

int fused_status(FusedPlan *p, unsigned *status)
{
    CSDR_HIP(hipMemcpy(status, p->d_status, sizeof(unsigned), hipMemcpyDeviceToHost));
    if (*status) CSDR_HIP(hipMemset(p->d_status, 0, sizeof(unsigned)));     // sticky until reported once
    return 0;
}

void fused_destroy(FusedPlan *p)
{
    if (!p) return;
    void *ptrs[] = {p->d_shard_tail, p->d_tail[0], p->d_tail[1], p->d_tail[2], p->d_taps, p->d_tw, p->d_wpre, p->d_yhist[0], p->d_yhist[1], p->d_vend[0], p->d_vend[1], p->d_rp[0],
                    p->d_rp[1], p->d_ticket, p->d_yflag, p->d_status, p->d_agg, p->d_ylast, p->d_premix, p->d_trace, p->d_cpre, p->d_side, p->d_rt};
    for (void *q : ptrs) if (q) (void)hipFree(q);
    delete p;
}

}  // namespace csdr
