// Fused run kernel for M = 64 channels (BASELINE.json configs[1]: 64-ch PFB, DeNo) on gfx950.
//
// Same structure as k_run256 (kernels_fused.hip): one workgroup (256 threads) walks a run of
// consecutive 4096-sample tiles with the DC-blocker state carried in registers and no
// inter-workgroup dependency (read-only DC warm-up at the run start, first freqdem sample of a
// run finished by a fix-up launch).  What changes with M = 64:
//   * a tile is T = 64 frames; the 256 threads are 4 frame groups x 64 polyphase branches, so the
//     14-tap FIR window of a thread crosses into frames that another thread finished: the
//     DC-blocked, pre-mixed samples u are written back to LDS (run mapping, 16-byte swizzled image)
//     and each thread reads its 13+16-frame window from there; the 13 frames before the tile live
//     in a small LDS history that survives from tile to tile;
//   * the 64-point DFT is 16 x 4: pass 1 = radix-16 in VGPRs per (frame, b), in place in LDS;
//     pass 2 = four radix-4 butterflies per thread;
//   * tail: thread (g, k) owns 16 consecutive frames of channel k; the sample before them comes
//     from LDS (previous group) or from the previous tile's saved last frame.
// The kernel handles every chunk of its handle (ragged tails by masking), and keeps its own
// stream state: DC v1, the last 13 frames of u, freqdem r'.
#include "fused_common.h"

namespace csdr {

namespace {

constexpr int LOG2M = 6;
constexpr int MS = 1 << LOG2M;        // 64 channels
constexpr int GS = 256 / MS;          // 4 frame groups
constexpr int TS = 16 * GS;           // 64 frames per tile
constexpr int R2 = MS / 16;           // 4: second DFT radix
constexpr int FSX = MS + R2;          // float2 stride between frames in the X / Z image (68)
constexpr int YS = TS + 1;            // float2 stride between channel rows in the Y image (65)
constexpr int RS_F2 = TS * FSX;       // 4352 float2 = 34 816 B >= 4096 (raw/u image) and MS*YS = 4160
constexpr int WPAD = MS + 8;          // row stride of the pre-mix phasor table in LDS

struct SmallArgs {
    TileArgs t;                 // x, out, taps, tw (16 x R2 pass-1 twiddles), wpre [2][MS], state, nf (frames), nb (tiles)...
    float2 *yfirst;             // [nruns][MS]
    float2 *ylast_run;          // [nruns][MS]
    uint32_t S;                 // tiles per run
    float l2beta;
};

// float2 index of sample n (0..4095) in the 16-byte-swizzled tile image
__device__ __forceinline__ int u_index(int n)
{
    const int q = n >> 4, i = (n & 15) >> 1;
    return 2 * (8 * q + (i ^ ((q >> 1) & 7))) + (n & 1);
}

// FM: two workgroups per CU (212 VGPRs) beat three with 41 spilled dwords (186 -> 209-216 GS/s); DeNo fits three
template <bool FM>
__global__ __launch_bounds__(256, FM ? 2 : 3) void k_run64(SmallArgs SA)
{
    const TileArgs &A = SA.t;
    __shared__ __attribute__((aligned(16))) float2 R[RS_F2];
    __shared__ float2 hist[13 * MS];        // u of the 13 frames before the current tile
    __shared__ float2 wpre_s[2 * WPAD];
    __shared__ float2 tw_s[16 * R2];
    __shared__ float2 yprev[2][MS];         // last Y frame of the previous tile (double buffered)
    __shared__ float2 Tt[16];
    __shared__ float2 red[4];
    __shared__ float taps_s[P * MS];        // taps_s[n][j] = h[(63 - j) + 64 n]

    const int tid = threadIdx.x;
    const unsigned w = blockIdx.x;
    for (int i = tid; i < P * MS; i += 256) taps_s[i] = A.taps[(MS - 1 - (i % MS)) + (i / MS) * MS];
    const unsigned first = w * SA.S, last = min(first + SA.S, A.nb);
    const float4 *x4 = reinterpret_cast<const float4 *>(A.x);
    if (tid < 16 * R2) tw_s[tid] = A.tw[tid];
    if (tid < 2 * MS) wpre_s[(tid >> LOG2M) * WPAD + (tid & (MS - 1))] = A.wpre[((A.parity0 ^ (tid >> LOG2M)) & 1) * MS + (tid & (MS - 1))];
    // (row 0 = phasors of even tile-relative frames, row 1 = odd; T is even, so this holds for every tile)

    float2 c;                                // DC state before the next tile (same in every lane)
    float4 raw[8];
    // one tile: zero-state scan (already staged in `raw`), DC finish + pre-mix written back as u
    auto finish_tile = [&](float2 carry, float2 &after, int tid) {
        const float br = A.b16[tid & 15], bf = A.b256[tid >> 4];
        const float2 e = stage_and_scan(raw, R, nullptr, Tt, A, tid);
        float2 vb, ve;
        frame_carries(Tt, A, tid, vb, ve);
        const float2 Pq = cfma(cfma(carry, bf, vb), br, e);        // v before my run
        after = cfma(carry, A.b256[16], ve);
        float4 *R4 = reinterpret_cast<float4 *>(R);
        const int q = tid, sw = (q >> 1) & 7;
        const int jb = (16 * q) & (MS - 1), prow = ((16 * q) >> LOG2M) & 1;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float4 z = R4[8 * q + (i ^ sw)];
            const v2f W0 = to_v(wpre_s[prow * WPAD + jb + 2 * i]), W1 = to_v(wpre_s[prow * WPAD + jb + 2 * i + 1]);
            const float k0 = -A.alpha * A.bj[2 * i], k1 = -A.alpha * A.bj[2 * i + 1];
            const v2f y0 = {fmaf(Pq.x, k0, z.x), fmaf(Pq.y, k0, z.y)}, y1 = {fmaf(Pq.x, k1, z.z), fmaf(Pq.y, k1, z.w)};
            const v2f u0 = cmul_v(y0, W0), u1 = cmul_v(y1, W1);
            R4[8 * q + (i ^ sw)] = make_float4(u0.x, u0.y, u1.x, u1.y);
        }
        return Pq;
    };

    if (w == 0) {
        c = A.vend_in[0];
        for (int i = tid; i < 13 * MS; i += 256) hist[i] = A.yhist_in[i];
        if (tid < MS) yprev[first & 1][tid] = (tid >= (int)A.c0 && tid < (int)(A.c0 + A.C)) ? A.rp_in[tid - A.c0] : make_float2(0.f, 0.f);
        __syncthreads();
    } else {
        // ---- warm-up: DC state before the halo tile from the WU tiles before it ----
        const unsigned halo = first - 1;
        const unsigned h0 = halo > (unsigned)WU ? halo - WU : 0u;
        float w0[8], w1[8];
        {
            const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
            for (int it = 0; it < 8; it++) {
                const int slot = 64 * (it * 4 + wave) + lane, q = slot >> 3;
                const int i = (slot & 7) ^ ((q >> 1) & 7);
                const int n = 16 * q + 2 * i;
                w0[it] = exp2f((float)(4095 - n) * SA.l2beta);
                w1[it] = exp2f((float)(4094 - n) * SA.l2beta);
            }
        }
        float2 acc = make_float2(0.f, 0.f);
        for (unsigned t = h0; t < halo; t++) {
            tile_load(x4 + (size_t)t * 2048, 256, raw, tid);
            float2 p = make_float2(0.f, 0.f);
#pragma unroll
            for (int it = 0; it < 8; it++) {
                p = cfma(make_float2(raw[it].x, raw[it].y), w0[it], p);
                p = cfma(make_float2(raw[it].z, raw[it].w), w1[it], p);
            }
            acc = cfma(acc, A.b256[16], p);
        }
        float2 ch = wg_sum(acc, red, tid);                          // also orders the LDS tables
        if (h0 == 0) ch = cfma(A.vend_in[0], exp2f((float)(4096u * halo) * SA.l2beta), ch);
        // ---- halo tile: its last 13 frames of u become the history ----
        tile_load(x4 + (size_t)halo * 2048, 256, raw, tid);
        (void)finish_tile(ch, c, tid);
        __syncthreads();
        for (int i = tid; i < 13 * MS; i += 256) hist[i] = R[u_index((TS - 13) * MS + i)];
        __syncthreads();
    }

    const bool vec_out = ((A.out_stride | A.out_t0) % 4u) == 0;
    const PhaseK pk = phase_consts(A.fm_ref);

    tile_load(x4 + (size_t)first * 2048, (int)min((unsigned)TS, A.nf - TS * first) * (MS / 16), raw, tid);
    for (unsigned b = first; b < last; b++) {
        const int nvalid = (int)min((unsigned)TS, A.nf - TS * b);       // frames
        // per-iteration copies of the thread coordinates keep the LDS address arithmetic inside the
        // tile loop (hoisted, it pins > 100 VGPRs)
        int tid_i = tid;
        asm volatile("" : "+v"(tid_i));
        const int g_i = tid_i >> LOG2M, j_i = tid_i & (MS - 1);
        const bool owned_i = (uint32_t)j_i >= A.c0 && (uint32_t)j_i < A.c0 + A.C;
        float2 cn;
        const float2 Pq = finish_tile(c, cn, tid_i);
        if (b + 1 == A.nb) {                                            // DC state after the last valid frame
            if (nvalid == TS) { if (tid_i == 0) A.vend_out[0] = cn; }
            else if (tid_i == nvalid * (MS / 16)) A.vend_out[0] = Pq;     // v before the first padded run
        }
        c = cn;
        if (b + 1 < last) tile_load(x4 + (size_t)(b + 1) * 2048, (int)min((unsigned)TS, A.nf - TS * (b + 1)) * (MS / 16), raw, tid_i);
        __syncthreads();                                                // u complete

        // ---- FIR window: frames 16g-13 .. 16g+15 of branch j_i ----
        float2 win[29];
#pragma unroll
        for (int i = 0; i < 29; i++) {
            const int t = 16 * g_i - 13 + i;                              // tile-relative frame
            win[i] = (t >= 0) ? R[u_index((t << LOG2M) + j_i)] : hist[(13 + t) * MS + j_i];
        }
        float2 hv[4];
        if (nvalid < TS) {
            // ragged last tile: the new history is frames nvalid-13 .. nvalid-1 (reaching back into the old one)
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const int idx = tid_i + 256 * m;
                hv[m] = make_float2(0.f, 0.f);
                if (idx < 13 * MS) {
                    const int t = nvalid - 13 + (idx >> LOG2M), jj = idx & (MS - 1);
                    hv[m] = (t >= 0) ? R[u_index((t << LOG2M) + jj)] : hist[(13 + t) * MS + jj];
                }
            }
        }
        __syncthreads();                                                // everyone holds its window; R, hist free
        // history for the next tile = the last 13 valid frames of u
        if (nvalid == TS) {
            if (g_i == GS - 1) {
#pragma unroll
                for (int i = 0; i < 13; i++) hist[i * MS + j_i] = win[16 + i];
            }
        } else {
#pragma unroll
            for (int m = 0; m < 4; m++) if (tid_i + 256 * m < 13 * MS) hist[tid_i + 256 * m] = hv[m];
        }

        // ---- polyphase FIR, oldest tap first ----
        {
            float h[P];
#pragma unroll
            for (int n = 0; n < P; n++) h[n] = taps_s[n * MS + j_i];
#pragma unroll
            for (int f = 0; f < 16; f++) {
                v2f acc = {0.f, 0.f};
#pragma unroll
                for (int n = P - 1; n >= 0; n--) {
                    const v2f sv = to_v(win[13 + f - n]), hv = {h[n], h[n]};
                    acc = __builtin_elementwise_fma(sv, hv, acc);
                }
                R[(16 * g_i + f) * FSX + j_i] = to_f2(acc);
            }
        }
        __syncthreads();                                                // X complete

        // ---- DFT pass 1 (radix 16 over a, j_i = 4a + b2), in place ----
        v2f vv[16];
        {
            const int f = tid_i >> 2, b2 = tid_i & 3;
#pragma unroll
            for (int a = 0; a < 16; a++) vv[a] = to_v(R[f * FSX + R2 * a + b2]);
            fft16_v(vv);
#pragma unroll
            for (int i = 1; i < 16; i++) vv[i] = cmul_v(vv[i], to_v(tw_s[R2 * XIDX(i) + b2]));
#pragma unroll
            for (int i = 0; i < 16; i++) R[f * FSX + R2 * XIDX(i) + b2] = to_f2(vv[i]);
        }
        __syncthreads();                                                // Z complete
        // ---- DFT pass 2 (radix 4 over b2): items (f, k1) = tid_i + 256 m ----
        {
#pragma unroll
            for (int m = 0; m < GS; m++) {
                const int item = tid_i + 256 * m, f = item >> 4, k1 = item & 15;
                const float4 za = *reinterpret_cast<const float4 *>(&R[f * FSX + R2 * k1]);
                const float4 zb = *reinterpret_cast<const float4 *>(&R[f * FSX + R2 * k1 + 2]);
                vv[4 * m + 0] = (v2f){za.x, za.y}; vv[4 * m + 1] = (v2f){za.z, za.w};
                vv[4 * m + 2] = (v2f){zb.x, zb.y}; vv[4 * m + 3] = (v2f){zb.z, zb.w};
                bfly4_v(vv[4 * m + 0], vv[4 * m + 1], vv[4 * m + 2], vv[4 * m + 3]);
            }
            __syncthreads();                                            // everyone has read Z
#pragma unroll
            for (int m = 0; m < GS; m++) {
                const int item = tid_i + 256 * m, f = item >> 4, k1 = item & 15;
#pragma unroll
                for (int k2 = 0; k2 < R2; k2++) R[(k1 + 16 * k2) * YS + f] = to_f2(vv[4 * m + k2]);
            }
        }
        __syncthreads();                                                // Y complete

        // ---- tail: thread (g_i, k = j_i) owns frames 16g .. 16g+15 of channel k ----
        float2 v[16];
#pragma unroll
        for (int f = 0; f < 16; f++) v[f] = R[j_i * YS + 16 * g_i + f];
        const size_t row = (size_t)(owned_i ? j_i - A.c0 : 0) * A.out_stride + A.out_t0 + (size_t)TS * b + 16 * g_i;
        const int nv = nvalid - 16 * g_i;                                 // valid frames of my group (may be <= 0)
        if (FM) {
            const float2 prev = g_i ? R[j_i * YS + 16 * g_i - 1] : yprev[b & 1][j_i];
            if (g_i == 0 && b == first && w > 0) SA.yfirst[(size_t)w * MS + j_i] = v[0];
            float m[16];
            if (nv > 0) {
#pragma unroll
                for (int f = 0; f < 16; f++) {
                    const float2 rp = f ? v[f - 1] : prev, r = v[f];
                    const float re = fmaf(rp.x, r.x, rp.y * r.y);
                    const float im = fmaf(rp.x, r.y, -(rp.y * r.x));
                    m[f] = scaled_atan2f(im, re, pk);
                }
            }
            if (vec_out && nvalid == TS) {
                // whole tile: transpose the demodulated samples through LDS so that 16 consecutive lanes write the
                // 16 x 16-byte pieces of one channel's 64 frames (a wave instruction = 4 rows x 256 B)
                __syncthreads();                                        // Y consumed by everyone
                float4 *M4 = reinterpret_cast<float4 *>(R);
#pragma unroll
                for (int q = 0; q < 4; q++) M4[j_i * 17 + 4 * g_i + q] = make_float4(m[4 * q], m[4 * q + 1], m[4 * q + 2], m[4 * q + 3]);
                // the next tile's r' is read from Y below: keep it before Y's region is reused
                if (g_i == 3) yprev[(b + 1) & 1][j_i] = v[15];
                __syncthreads();
                float *obase = (float *)A.out + A.out_t0 + (size_t)TS * b;
#pragma unroll
                for (int it = 0; it < 4; it++) {
                    const int item = tid_i + 256 * it, rowk = item >> 4, piece = item & 15;
                    if ((uint32_t)rowk >= A.c0 && (uint32_t)rowk < A.c0 + A.C)
                        *reinterpret_cast<float4 *>(obase + (size_t)(rowk - A.c0) * A.out_stride + 4 * piece) = M4[rowk * 17 + piece];
                }
            } else {
                if (owned_i && nv > 0) {
                    float *o = (float *)A.out + row;
#pragma unroll
                    for (int f = 0; f < 16; f++) if (f < nv) o[f] = m[f];
                }
                // last valid frame of the tile -> next tile's r'
                if (nv >= 1 && nv <= 16) {
                    float2 lv = v[0];
#pragma unroll
                    for (int f = 1; f < 16; f++) if (f < nv) lv = v[f];
                    yprev[(b + 1) & 1][j_i] = lv;
                }
            }
        } else if (vec_out && nvalid == TS) {
            // CF32 rows leave as whole lines: 32 consecutive lanes write the 32 x 16-byte pieces of one channel's
            // 64 frames (a wave instruction = 2 rows x 512 B) instead of 64 lanes x 16 B of scattered rows
            float2 *obase = (float2 *)A.out + A.out_t0 + (size_t)TS * b;
#pragma unroll
            for (int it = 0; it < 8; it++) {
                const int item = tid_i + 256 * it, rowk = item >> 5, piece = item & 31;
                const float2 a0 = R[rowk * YS + 2 * piece], a1 = R[rowk * YS + 2 * piece + 1];
                if ((uint32_t)rowk >= A.c0 && (uint32_t)rowk < A.c0 + A.C)
                    *reinterpret_cast<float4 *>(obase + (size_t)(rowk - A.c0) * A.out_stride + 2 * piece) = make_float4(a0.x, a0.y, a1.x, a1.y);
            }
        } else if (owned_i && nv > 0) {
            float2 *o = (float2 *)A.out + row;
#pragma unroll
            for (int f = 0; f < 16; f++) if (f < nv) o[f] = v[f];
        }
        __syncthreads();                                                // Y consumed, R free
    }
    // ---- what the run leaves behind ----
    if (FM && tid < MS) SA.ylast_run[(size_t)w * MS + tid] = yprev[last & 1][tid];
    if (last == A.nb) {
        for (int i = tid; i < 13 * MS; i += 256) A.yhist_out[i] = hist[i];
        if (FM && tid < MS && (uint32_t)tid >= A.c0 && (uint32_t)tid < A.c0 + A.C) A.rp_out[tid - A.c0] = yprev[last & 1][tid];
    }
}

__global__ __launch_bounds__(64) void k_run64_fixup(const float2 *__restrict__ yfirst, const float2 *__restrict__ ylast,
                                                    float *__restrict__ out, uint32_t out_stride, uint32_t frames_per_run,
                                                    uint32_t c0, uint32_t C, float ref)
{
    const uint32_t k = threadIdx.x, w = blockIdx.x + 1;
    if (k < c0 || k >= c0 + C) return;
    const float2 r = yfirst[(size_t)w * MS + k], rp = ylast[(size_t)(w - 1) * MS + k];
    const float re = fmaf(rp.x, r.x, rp.y * r.y);
    const float im = fmaf(rp.x, r.y, -(rp.y * r.x));
    out[(size_t)(k - c0) * out_stride + (size_t)frames_per_run * w] = fast_atan2f(im, re) * ref;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct SmallPlan {
    FusedConfig cfg;
    uint32_t max_nb = 0, resident_wgs = 768, cus = 256;
    bool v2_ok = false, v2_last = false;     // k_run64v2 usable (CF32, whole band); used by the last call
    uint64_t frames_done = 0;
    float *d_taps = nullptr;
    float2 *d_tw = nullptr, *d_wpre = nullptr;
    float2 *d_yhist[2] = {nullptr, nullptr}, *d_vend[2] = {nullptr, nullptr}, *d_rp[2] = {nullptr, nullptr};
    float2 *d_yfirst = nullptr, *d_ylast = nullptr;
    float2 *d_cpre = nullptr, *d_rt = nullptr;      // k_run64v2 without warm-up windows (Run64v2Host::cpre / rt)
    void *d_premix = nullptr;
    int cur = 0;
    TileArgs proto;
};

bool small_supported(uint32_t M, uint32_t p) { return M == (uint32_t)MS && p == (uint32_t)P; }

void small_destroy(SmallPlan *p)
{
    if (!p) return;
    void *ptrs[] = {p->d_taps, p->d_tw, p->d_wpre, p->d_yhist[0], p->d_yhist[1], p->d_vend[0], p->d_vend[1], p->d_rp[0],
                    p->d_rp[1], p->d_yfirst, p->d_ylast, p->d_premix, p->d_cpre, p->d_rt};
    for (void *q : ptrs) if (q) (void)hipFree(q);
    delete p;
}

int small_create(const FusedConfig &cfg, SmallPlan **out)
{
    SmallPlan *p = new SmallPlan();
    p->cfg = cfg;
    p->max_nb = (cfg.max_nf + TS - 1) / TS;
    auto fail = [&](int r) { small_destroy(p); return r; };
#define ALLOC(ptr, bytes) do { hipError_t e = hipMalloc((void **)&(ptr), (bytes) ? (bytes) : 1); if (e != hipSuccess) return fail(hip_fail(e, "hipMalloc", __FILE__, __LINE__)); } while (0)
    ALLOC(p->d_taps, sizeof(float) * cfg.M * cfg.p);
    ALLOC(p->d_tw, sizeof(float2) * 16 * R2);
    ALLOC(p->d_wpre, sizeof(float2) * 2 * cfg.M);
    for (int i = 0; i < 2; i++) {
        ALLOC(p->d_yhist[i], sizeof(float2) * 13 * cfg.M);
        ALLOC(p->d_vend[i], sizeof(float2));
        ALLOC(p->d_rp[i], sizeof(float2) * cfg.C);
    }
    ALLOC(p->d_yfirst, sizeof(float2) * (size_t)cfg.M * (p->max_nb + 2));
    ALLOC(p->d_ylast, sizeof(float2) * (size_t)cfg.M * (p->max_nb + 2));
    if (cfg.mix) ALLOC(p->d_premix, (size_t)cfg.C * cfg.max_nf * (cfg.fm ? 4 : 8));
#undef ALLOC
    CSDR_HIP(hipMemcpy(p->d_taps, cfg.taps, sizeof(float) * cfg.M * cfg.p, hipMemcpyHostToDevice));
    std::vector<float2> tw(16 * R2), wpre(2 * cfg.M);
    for (int k1 = 0; k1 < 16; k1++)
        for (int b2 = 0; b2 < R2; b2++) {
            const double a = -2.0 * 3.14159265358979323846 * (double)(b2 * k1) / (double)cfg.M;
            tw[R2 * k1 + b2] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
    for (uint32_t i = 0; i < 2 * cfg.M; i++) {       // nco phase sequence has period 2M for a power-of-two M
        float c, s;
        nco_phasor(i * cfg.d_theta, &c, &s);
        wpre[i] = make_float2(c, -s);
    }
    CSDR_HIP(hipMemcpy(p->d_tw, tw.data(), sizeof(float2) * tw.size(), hipMemcpyHostToDevice));
    CSDR_HIP(hipMemcpy(p->d_wpre, wpre.data(), sizeof(float2) * wpre.size(), hipMemcpyHostToDevice));
    TileArgs &A = p->proto;
    A = TileArgs{};
    A.taps = p->d_taps; A.tw = p->d_tw; A.wpre = p->d_wpre;
    A.c0 = cfg.c0; A.C = cfg.C; A.fm_ref = cfg.fm_ref;
    const double beta = cfg.dc_block ? (double)cfg.dc.beta : 0.0;
    A.alpha = cfg.dc_block ? (float)(1.0 - beta) : 0.0f;
    A.beta = (float)beta;
    for (int k = 0; k < 16; k++) A.b16[k] = (float)std::pow(beta, 16.0 * k);
    for (int k = 0; k < 17; k++) A.b256[k] = (float)std::pow(beta, 256.0 * k);
    for (int k = 0; k < 16; k++) A.bj[k] = (float)std::pow(beta, (double)k);
    {
        int dev = 0, cus = 256, occ = 3;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (cfg.fm) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_run64<true>, 256, 0);
        else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_run64<false>, 256, 0);
        if (occ < 1) occ = 1;
        p->resident_wgs = (uint32_t)(cus * occ);
        p->cus = (uint32_t)cus;
        p->v2_ok = !cfg.fm && cfg.c0 == 0 && cfg.C == (uint32_t)MS && !diag_env("CSDR_RUN64_V1");
    }
    if (p->v2_ok && cfg.dc_block && !(diag_env("CSDR_NOWU") && atoi(diag_env("CSDR_NOWU")) == 0)) {
        // k_run64v2 without warm-up windows: the state hand-over array and the chain's response to a unit DC state at the channels 30..33,
        // frames 64 .. 64 + RUN64_DCFIX_F - 1 behind a halo tile's start (= the run's first output frames)
        hipError_t e1 = hipMalloc((void **)&p->d_cpre, sizeof(float2) * 1026), e2 = hipMalloc((void **)&p->d_rt, sizeof(float2) * 2 * RUN64_DCFIX_F * 4);
        if (e1 != hipSuccess || e2 != hipSuccess) return fail(hip_fail(e1 != hipSuccess ? e1 : e2, "hipMalloc", __FILE__, __LINE__));
        CSDR_HIP(hipMemset(p->d_cpre, 0, sizeof(float2) * 1026));
        std::vector<float2> rt((size_t)2 * RUN64_DCFIX_F * 4);
        dc_state_response(cfg, wpre.data(), 64u, (uint32_t)RUN64_DCFIX_F, 30u, rt.data());
        CSDR_HIP(hipMemcpy(p->d_rt, rt.data(), sizeof(float2) * rt.size(), hipMemcpyHostToDevice));
    }
    *out = p;
    return 0;
}

int small_reset(SmallPlan *p, hipStream_t s)
{
    p->cur = 0; p->frames_done = 0;
    for (int i = 0; i < 2; i++) {
        CSDR_HIP(hipMemsetAsync(p->d_yhist[i], 0, sizeof(float2) * 13 * p->cfg.M, s));
        CSDR_HIP(hipMemsetAsync(p->d_vend[i], 0, sizeof(float2), s));
        CSDR_HIP(hipMemsetAsync(p->d_rp[i], 0, sizeof(float2) * p->cfg.C, s));
    }
    return 0;
}

void small_seek(SmallPlan *p, uint64_t frames) { p->frames_done = frames; }

bool small_tile_major_ok(const SmallPlan *p, uint32_t nf)
{
    return p && p->v2_ok && !p->cfg.mix && !p->cfg.fm && (uint64_t)p->cfg.C * nf * 8u < (1ull << 32) && run64_v2_runs(nf, p->cus) != 0;
}

int small_process(SmallPlan *p, const FusedCall &call, hipStream_t s, KernelTimer *timer)
{
    const FusedConfig &c = p->cfg;
    const uint32_t nf = call.nf;
    if (!nf) return 0;
    int r;
    const uint32_t v2runs = (p->v2_ok && (uint64_t)c.C * nf * 8u < (1ull << 32)) ? run64_v2_runs(nf, p->cus) : 0;
    p->v2_last = v2runs != 0;
    if (call.tile_major && !v2runs) { set_error("small_process: tile-major output asked for a call k_run64v2 does not take"); return -1; }
    if (v2runs) {
        Run64v2Host H{};
        H.x = call.d_in; H.out = (float2 *)(c.mix ? p->d_premix : call.d_out);
        H.taps = p->d_taps; H.tw = p->d_tw; H.wpre = p->d_wpre;
        H.uhist_in = p->d_yhist[p->cur]; H.uhist_out = p->d_yhist[p->cur ^ 1];
        H.vend_in = p->d_vend[p->cur]; H.vend_out = p->d_vend[p->cur ^ 1];
        H.nf = nf; H.nruns = v2runs; H.parity0 = (uint32_t)(p->frames_done & 1);
        H.tile_major = call.tile_major;
        H.cpre = v2runs <= 1024u ? p->d_cpre : nullptr; H.rt = p->d_rt;
        H.dc_block = c.dc_block; H.beta = c.dc_block ? (double)c.dc.beta : 0.0;
        if ((r = run64_v2_launch(H, s, timer))) return r;
        p->cur ^= 1;
        p->frames_done += nf;
        if (c.mix) {
            if ((r = launch_mix((const float *)p->d_premix, (float *)call.d_out, c.C, 2 * nf, s))) return r;
        }
        return 0;
    }
    SmallArgs SA{};
    TileArgs &A = SA.t;
    A = p->proto;
    A.x = call.d_in;
    A.out = c.mix ? p->d_premix : call.d_out;
    A.yhist_in = p->d_yhist[p->cur]; A.yhist_out = p->d_yhist[p->cur ^ 1];
    A.vend_in = p->d_vend[p->cur];   A.vend_out = p->d_vend[p->cur ^ 1];
    A.rp_in = p->d_rp[p->cur];       A.rp_out = p->d_rp[p->cur ^ 1];
    A.nf = nf; A.nb = (nf + TS - 1) / TS; A.out_stride = nf; A.out_t0 = 0;
    A.parity0 = (uint32_t)(p->frames_done & 1);
    SA.yfirst = p->d_yfirst; SA.ylast_run = p->d_ylast;
    // one run per resident workgroup slot, at least 8 tiles per run once the chunk is large enough
    SA.S = (A.nb + p->resident_wgs - 1) / p->resident_wgs;
    if (SA.S < 8) SA.S = A.nb >= 8 * 64 ? 8 : (A.nb + 63) / 64;
    if (SA.S < 1) SA.S = 1;
    SA.l2beta = c.dc_block ? (float)std::log2((double)c.dc.beta) : -1000.0f;
    const uint32_t nruns = (A.nb + SA.S - 1) / SA.S;
    if (timer && (r = timer->begin(s))) return r;
    if (c.fm) hipLaunchKernelGGL(k_run64<true>, dim3(nruns), dim3(256), 0, s, SA);
    else hipLaunchKernelGGL(k_run64<false>, dim3(nruns), dim3(256), 0, s, SA);
    if (timer && (r = timer->end(s))) return r;
    if (c.fm && nruns > 1)
        hipLaunchKernelGGL(k_run64_fixup, dim3(nruns - 1), dim3(64), 0, s, p->d_yfirst, p->d_ylast, (float *)A.out, nf,
                           SA.S * TS, c.c0, c.C, c.fm_ref);
    CSDR_HIP(hipGetLastError());
    p->cur ^= 1;
    p->frames_done += nf;
    if (c.mix) {
        if ((r = launch_mix((const float *)p->d_premix, (float *)call.d_out, c.C, c.fm ? nf : 2 * nf, s))) return r;
    }
    return 0;
}

const char *small_name(const SmallPlan *p) { return p->cfg.fm ? "k_run64<FM>" : (p->v2_last ? "k_run64v2" : "k_run64<CF32>"); }

}  // namespace csdr
