// Fused polyphase-channelizer kernel for gfx950 (wave64, 256 CUs, 160 KiB LDS/CU).
//
//   premixed input u  ->  14-tap polyphase branch filters (sliding window in VGPRs)
//                     ->  256-point forward DFT as 16 x 16 (two radix-16 passes in VGPRs,
//                         exchanged through LDS)
//                     ->  LDS transpose to channel-major
//                     ->  per-channel tail (freqdem) -> coalesced row stores [C][nf]
//
// One workgroup = 256 threads = one thread per polyphase branch j; it walks a run of
// frames in batches of 16.  Thread roles per batch:
//   FIR      thread j            : X_f[j] for f = 0..15 from a 13+16 deep register window
//   pass 1   thread (f, b)       : Z_f[k1][b] = W256^(b k1) * sum_a X_f[16a+b] W16^(a k1)
//   pass 2   thread (f, k1)      : Y_f[k1+16 k2] = sum_b Z_f[k1][b] W16^(b k2)
//   tail     thread k            : 16 consecutive time samples of channel k
// MFMA is not used: the path is streaming FIR/FFT bounded by HBM and VALU (north_star).
//
// Replaces per chunk: nf x firpfbch_crcf_analyzer_execute + the Haskell transpose
// (Liquid.chs:840-849) + M x freqdem_demodulate_block (Liquid.chs:324-328).
#include "fused.h"

#include <cmath>
#include <vector>

namespace csdr {

namespace {

constexpr int M256 = 256;
constexpr int P = 14;            // taps per branch (2m, m = 7)
constexpr int NB = 16;           // frames per batch
constexpr int FS_X = 272;        // float2 stride between frames, FIR -> pass-1 layout
constexpr int FS_Z = 289;        // float2 stride between frames, pass-1 -> pass-2 layout
constexpr int RS_Z = 18;         // float2 stride between k1 rows inside a frame (16 + 2 pad)
constexpr int RS_Y = 17;         // float2 stride between channel rows, pass-2 -> tail layout
constexpr int LDS_F2 = 16 * FS_Z;   // 4624 float2 = 36992 B (largest of the three layouts)

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// multiply by -j
__device__ __forceinline__ float2 mulmj(float2 a) { return make_float2(a.y, -a.x); }

// forward radix-4 butterfly (W4 = -j)
__device__ __forceinline__ void bfly4(float2 &x0, float2 &x1, float2 &x2, float2 &x3)
{
    const float2 s02 = cadd(x0, x2), d02 = csub(x0, x2);
    const float2 s13 = cadd(x1, x3), d13 = mulmj(csub(x1, x3));
    x0 = cadd(s02, s13);
    x1 = cadd(d02, d13);
    x2 = csub(s02, s13);
    x3 = csub(d02, d13);
}

// In-register forward 16-point DFT, natural order in and out.
__device__ __forceinline__ void fft16(float2 (&v)[16])
{
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
    // 4 butterflies over c (stride-4 subsequences): t[a][q] in v[a + 4q]
#pragma unroll
    for (int a = 0; a < 4; a++) bfly4(v[a], v[a + 4], v[a + 8], v[a + 12]);
    // twiddle W16^(a q) on element (a, q) = v[a + 4q]
    v[1 + 4] = cmul(v[1 + 4], make_float2(C1, -S1));            // W^1
    v[1 + 8] = cmul(v[1 + 8], make_float2(R2, -R2));            // W^2
    v[1 + 12] = cmul(v[1 + 12], make_float2(S1, -C1));          // W^3
    v[2 + 4] = cmul(v[2 + 4], make_float2(R2, -R2));            // W^2
    v[2 + 8] = mulmj(v[2 + 8]);                                 // W^4 = -j
    v[2 + 12] = cmul(v[2 + 12], make_float2(-R2, -R2));         // W^6
    v[3 + 4] = cmul(v[3 + 4], make_float2(S1, -C1));            // W^3
    v[3 + 8] = cmul(v[3 + 8], make_float2(-R2, -R2));           // W^6
    v[3 + 12] = cmul(v[3 + 12], make_float2(-C1, S1));          // W^9
    // 4 butterflies over a for each q: X[q + 4r]
#pragma unroll
    for (int q = 0; q < 4; q++) bfly4(v[4 * q + 0], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    // now v[4q + r] = X[q + 4r]: transpose the 4x4 index to natural order
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int r = q + 1; r < 4; r++) {
            const float2 t = v[4 * q + r];
            v[4 * q + r] = v[4 * r + q];
            v[4 * r + q] = t;
        }
}

struct FusedArgs {
    const float2 *u;          // first NEW premixed sample; (P-1)*M samples of history before it
    const float *taps;        // [P][M] prototype taps
    const float2 *tw;         // W256^i, i < 256
    void *out;                // [C][nf]
    float2 *bound_first;      // [nwg][M]  Y of each run's first frame
    float2 *bound_last;       // [nwg][M]  Y of each run's last frame
    uint32_t nf, run;         // frames in this call, frames per workgroup (multiple of 16)
    uint32_t c0, C;
    float fm_ref;
};

template <bool FM>
__global__ __launch_bounds__(256) void k_fused256(FusedArgs A)
{
    __shared__ __attribute__((aligned(16))) float2 lds[LDS_F2];
    __shared__ float2 tw_s[M256];

    const int tid = threadIdx.x;
    const int64_t t0 = (int64_t)blockIdx.x * A.run;
    const int64_t t1 = min((int64_t)A.nf, t0 + (int64_t)A.run);

    tw_s[tid] = A.tw[tid];

    // branch taps: X[j] uses h[(M-1-j) + n*M]
    float h[P];
#pragma unroll
    for (int n = 0; n < P; n++) h[n] = A.taps[(M256 - 1 - tid) + n * M256];

    // sliding window: old[3..15] = the 13 frames before the run
    float2 old[NB], nw[NB];
    const float2 *ucol = A.u + tid;
#pragma unroll
    for (int i = 3; i < NB; i++) old[i] = ucol[(t0 - NB + i) * M256];
    old[0] = old[1] = old[2] = make_float2(0.f, 0.f);

    float2 prev = make_float2(0.f, 0.f);     // channel tid's previous sample (tail role)
    const bool owned = (uint32_t)tid >= A.c0 && (uint32_t)tid < A.c0 + A.C;
    const bool vec_ok = (A.nf % 4u) == 0;

    for (int64_t tb = t0; tb < t1; tb += NB) {
        const int nvalid = (int)min((int64_t)NB, t1 - tb);
        // ---- load 16 new frames of my branch ----
#pragma unroll
        for (int f = 0; f < NB; f++)
            nw[f] = (f < nvalid) ? ucol[(tb + f) * M256] : make_float2(0.f, 0.f);

        // ---- polyphase FIR, oldest tap first (dotprod_crcf order) ----
#pragma unroll
        for (int f = 0; f < NB; f++) {
            float2 acc = make_float2(0.f, 0.f);
#pragma unroll
            for (int n = P - 1; n >= 0; n--) {
                const int i = f - n;
                const float2 s = (i >= 0) ? nw[i] : old[NB + i];
                acc.x = fmaf(h[n], s.x, acc.x);
                acc.y = fmaf(h[n], s.y, acc.y);
            }
            lds[f * FS_X + tid] = acc;
        }
#pragma unroll
        for (int f = 0; f < NB; f++) old[f] = nw[f];
        __syncthreads();                                        // B1: X complete

        // ---- pass 1: thread (f, b) ----
        float2 v[16];
        {
            const int f = tid >> 4, b = tid & 15;
#pragma unroll
            for (int a = 0; a < 16; a++) v[a] = lds[f * FS_X + 16 * a + b];
            fft16(v);
#pragma unroll
            for (int k1 = 1; k1 < 16; k1++) v[k1] = cmul(v[k1], tw_s[b * k1]);
            __syncthreads();                                    // B2: everyone has read X
#pragma unroll
            for (int k1 = 0; k1 < 16; k1++) lds[f * FS_Z + k1 * RS_Z + b] = v[k1];
        }
        __syncthreads();                                        // B3: Z complete

        // ---- pass 2: thread (f, k1) ----
        {
            const int f = tid >> 4, k1 = tid & 15;
#pragma unroll
            for (int b = 0; b < 16; b++) v[b] = lds[f * FS_Z + k1 * RS_Z + b];
            fft16(v);
            __syncthreads();                                    // B4: everyone has read Z
#pragma unroll
            for (int k2 = 0; k2 < 16; k2++) lds[(k1 + 16 * k2) * RS_Y + f] = v[k2];
        }
        __syncthreads();                                        // B5: Y complete

        // ---- tail: thread k = tid owns channel k, 16 consecutive samples ----
#pragma unroll
        for (int f = 0; f < NB; f++) v[f] = lds[tid * RS_Y + f];
        if (tb == t0) A.bound_first[(size_t)blockIdx.x * M256 + tid] = v[0];
        if (tb + NB >= t1) {
            float2 last = v[0];
#pragma unroll
            for (int f = 1; f < NB; f++) if (f < nvalid) last = v[f];
            A.bound_last[(size_t)blockIdx.x * M256 + tid] = last;
        }
        if (owned) {
            const size_t row = (size_t)(tid - A.c0) * A.nf + (size_t)tb;
            if (FM) {
                float m[NB];
#pragma unroll
                for (int f = 0; f < NB; f++) {
                    const float2 r = v[f];
                    const float re = __fadd_rn(__fmul_rn(prev.x, r.x), __fmul_rn(prev.y, r.y));
                    const float im = __fsub_rn(__fmul_rn(prev.x, r.y), __fmul_rn(prev.y, r.x));
                    m[f] = atan2f(im, re) * A.fm_ref;
                    prev = r;
                }
                float *o = (float *)A.out + row;
                if (vec_ok && nvalid == NB) {
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        *reinterpret_cast<float4 *>(o + 4 * q) = make_float4(m[4 * q], m[4 * q + 1], m[4 * q + 2], m[4 * q + 3]);
                } else {
#pragma unroll
                    for (int f = 0; f < NB; f++) if (f < nvalid) o[f] = m[f];
                }
            } else {
                float2 *o = (float2 *)A.out + row;
                if ((A.nf % 2u) == 0 && nvalid == NB) {
#pragma unroll
                    for (int q = 0; q < 8; q++)
                        *reinterpret_cast<float4 *>(o + 2 * q) = make_float4(v[2 * q].x, v[2 * q].y, v[2 * q + 1].x, v[2 * q + 1].y);
                } else {
#pragma unroll
                    for (int f = 0; f < NB; f++) if (f < nvalid) o[f] = v[f];
                }
            }
        }
        __syncthreads();                                        // B6: Y consumed
    }
}

// out[c][t0(w)] for every run start: freqdem against the previous run's last sample
__global__ __launch_bounds__(256) void k_fm_fixup(const float2 *__restrict__ bound_first,
                                                  const float2 *__restrict__ bound_last, const float2 *__restrict__ rp_in,
                                                  float2 *__restrict__ rp_out, float *__restrict__ out, uint32_t nf,
                                                  uint32_t run, uint32_t nwg, uint32_t c0, uint32_t C, float ref)
{
    const uint32_t k = threadIdx.x, w = blockIdx.x;
    if (w == nwg) {                                  // extra block: save r' for the next call
        if (k >= c0 && k < c0 + C) rp_out[k - c0] = bound_last[(size_t)(nwg - 1) * M256 + k];
        return;
    }
    if (k < c0 || k >= c0 + C) return;
    const float2 r = bound_first[(size_t)w * M256 + k];
    const float2 rp = w ? bound_last[(size_t)(w - 1) * M256 + k] : rp_in[k - c0];
    const float re = __fadd_rn(__fmul_rn(rp.x, r.x), __fmul_rn(rp.y, r.y));
    const float im = __fsub_rn(__fmul_rn(rp.x, r.y), __fmul_rn(rp.y, r.x));
    out[(size_t)(k - c0) * nf + (size_t)w * run] = atan2f(im, re) * ref;
}

}  // namespace

struct FusedPlan {
    FusedConfig cfg;
    std::string name;
    size_t hist;                 // (p-1)*M
    float *d_taps = nullptr;
    float2 *d_tw = nullptr, *d_u = nullptr, *d_hist_tmp = nullptr, *d_dcstate = nullptr, *d_scratch = nullptr;
    float2 *d_nco_tab = nullptr; uint32_t tab_len = 0, tab_pos = 0;
    float2 *d_bfirst = nullptr, *d_blast = nullptr; uint32_t max_wg = 0;
    float2 *d_rp[2] = {nullptr, nullptr}; int rp_cur = 0;
    void *d_premix = nullptr;    // per-channel output before mixing
};

bool fused_supported(uint32_t M, uint32_t p) { return M == 256 && p == P; }

static uint32_t pick_run(uint32_t nf)
{
    // frames per workgroup: multiple of 16, aiming at >= ~2048 workgroups on large chunks
    uint32_t run = (nf + 2047) / 2048;
    run = (run + NB - 1) / NB * NB;
    if (run < NB) run = NB;
    return run;
}

int fused_create(const FusedConfig &cfg, FusedPlan **out)
{
    FusedPlan *p = new FusedPlan();
    p->cfg = cfg;
    p->name = cfg.fm ? "k_fused256<FM>" : "k_fused256<CF32>";
    p->hist = (size_t)(cfg.p - 1) * cfg.M;
    const uint64_t max_nx = (uint64_t)cfg.max_nf * cfg.M;
    auto fail = [&](int r) { fused_destroy(p); return r; };
#define ALLOC(ptr, count) do { hipError_t e = hipMalloc((void **)&(ptr), (count) ? (count) : 1); if (e != hipSuccess) return fail(hip_fail(e, "hipMalloc", __FILE__, __LINE__)); } while (0)
    ALLOC(p->d_taps, sizeof(float) * cfg.M * cfg.p);
    ALLOC(p->d_tw, sizeof(float2) * cfg.M);
    ALLOC(p->d_u, sizeof(float2) * (p->hist + max_nx));
    ALLOC(p->d_hist_tmp, sizeof(float2) * p->hist);
    ALLOC(p->d_dcstate, sizeof(float2));
    ALLOC(p->d_scratch, sizeof(float2) * 2 * (max_nx / DC_BLOCK + 2));
    p->max_wg = (cfg.max_nf + NB - 1) / NB;
    ALLOC(p->d_bfirst, sizeof(float2) * (size_t)p->max_wg * cfg.M);
    ALLOC(p->d_blast, sizeof(float2) * (size_t)p->max_wg * cfg.M);
    if (cfg.fm) { ALLOC(p->d_rp[0], sizeof(float2) * cfg.C); ALLOC(p->d_rp[1], sizeof(float2) * cfg.C); }
    if (cfg.mix) ALLOC(p->d_premix, (size_t)cfg.C * cfg.max_nf * (cfg.fm ? 4 : 8));
#undef ALLOC
    CSDR_HIP(hipMemcpy(p->d_taps, cfg.taps, sizeof(float) * cfg.M * cfg.p, hipMemcpyHostToDevice));
    std::vector<float2> tw(cfg.M);
    for (uint32_t i = 0; i < cfg.M; i++) {
        const double a = -2.0 * 3.14159265358979323846 * (double)i / (double)cfg.M;
        tw[i] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    CSDR_HIP(hipMemcpy(p->d_tw, tw.data(), sizeof(float2) * cfg.M, hipMemcpyHostToDevice));
    p->tab_len = nco_period(cfg.d_theta, 1u << 17);
    if (p->tab_len) {
        std::vector<float2> tab(p->tab_len);
        for (uint32_t i = 0; i < p->tab_len; i++) { float c, s; nco_phasor(i * cfg.d_theta, &c, &s); tab[i] = make_float2(c, s); }
        hipError_t e = hipMalloc((void **)&p->d_nco_tab, sizeof(float2) * p->tab_len);
        if (e != hipSuccess) return fail(hip_fail(e, "hipMalloc", __FILE__, __LINE__));
        CSDR_HIP(hipMemcpy(p->d_nco_tab, tab.data(), sizeof(float2) * p->tab_len, hipMemcpyHostToDevice));
    }
    *out = p;
    return 0;
}

int fused_reset(FusedPlan *p, hipStream_t s)
{
    p->tab_pos = 0; p->rp_cur = 0;
    CSDR_HIP(hipMemsetAsync(p->d_u, 0, sizeof(float2) * p->hist, s));
    CSDR_HIP(hipMemsetAsync(p->d_dcstate, 0, sizeof(float2), s));
    if (p->d_rp[0]) {
        CSDR_HIP(hipMemsetAsync(p->d_rp[0], 0, sizeof(float2) * p->cfg.C, s));
        CSDR_HIP(hipMemsetAsync(p->d_rp[1], 0, sizeof(float2) * p->cfg.C, s));
    }
    return 0;
}

int fused_process(FusedPlan *p, const FusedCall &call, hipStream_t s)
{
    const FusedConfig &c = p->cfg;
    const uint32_t nf = call.nf, nx = nf * c.M;
    if (!nf) return 0;
    int r;
    // stage 0 (separate pass for now): DC blocker + NCO pre-mix into u (history in front)
    NcoParams nco{};
    nco.theta0 = call.theta0; nco.d_theta = c.d_theta; nco.tab_len = p->tab_len; nco.tab_pos = p->tab_pos; nco.up = 0;
    float2 *u_new = p->d_u + p->hist;
    if ((r = launch_dc_mix(call.d_in, u_new, nx, c.dc_block, c.dc, p->d_dcstate, p->d_scratch, true, nco, p->d_nco_tab, s))) return r;
    if (p->tab_len) p->tab_pos = (uint32_t)(((uint64_t)p->tab_pos + nx) % p->tab_len);

    FusedArgs A{};
    A.u = u_new; A.taps = p->d_taps; A.tw = p->d_tw;
    A.out = c.mix ? p->d_premix : call.d_out;
    A.bound_first = p->d_bfirst; A.bound_last = p->d_blast;
    A.nf = nf; A.run = pick_run(nf); A.c0 = c.c0; A.C = c.C; A.fm_ref = c.fm_ref;
    const uint32_t nwg = (nf + A.run - 1) / A.run;
    if (c.fm) {
        hipLaunchKernelGGL(k_fused256<true>, dim3(nwg), dim3(256), 0, s, A);
        hipLaunchKernelGGL(k_fm_fixup, dim3(nwg + 1), dim3(256), 0, s, p->d_bfirst, p->d_blast, p->d_rp[p->rp_cur],
                           p->d_rp[p->rp_cur ^ 1], (float *)A.out, nf, A.run, nwg, c.c0, c.C, c.fm_ref);
        p->rp_cur ^= 1;
    } else {
        hipLaunchKernelGGL(k_fused256<false>, dim3(nwg), dim3(256), 0, s, A);
    }
    CSDR_HIP(hipGetLastError());
    // keep the last (p-1) frames of premixed input as the next call's history
    CSDR_HIP(hipMemcpyAsync(p->d_hist_tmp, p->d_u + nx, sizeof(float2) * p->hist, hipMemcpyDeviceToDevice, s));
    CSDR_HIP(hipMemcpyAsync(p->d_u, p->d_hist_tmp, sizeof(float2) * p->hist, hipMemcpyDeviceToDevice, s));
    if (c.mix) {
        if ((r = launch_mix((const float *)p->d_premix, (float *)call.d_out, c.C, c.fm ? nf : 2 * nf, s))) return r;
    }
    return 0;
}

const char *fused_name(const FusedPlan *p) { return p->name.c_str(); }

void fused_destroy(FusedPlan *p)
{
    if (!p) return;
    void *ptrs[] = {p->d_taps, p->d_tw, p->d_u, p->d_hist_tmp, p->d_dcstate, p->d_scratch, p->d_nco_tab, p->d_bfirst,
                    p->d_blast, p->d_rp[0], p->d_rp[1], p->d_premix};
    for (void *q : ptrs) if (q) (void)hipFree(q);
    delete p;
}

}  // namespace csdr
