// Internal declarations shared by the HIP translation units of libcsdr_hip.so.
// Product code: must not include or link anything under oracle/.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string>
#include <vector>

namespace csdr {

// ---- error plumbing -------------------------------------------------------
void set_error(const char *fmt, ...);
int  hip_fail(hipError_t e, const char *what, const char *file, int line);
#define CSDR_HIP(call)                                                          \
    do {                                                                        \
        hipError_t e__ = (call);                                                \
        if (e__ != hipSuccess) return ::csdr::hip_fail(e__, #call, __FILE__, __LINE__); \
    } while (0)

// same, for create functions: run `cleanup` (the object's destroy) before returning the error
#define CSDR_HIP_CLEAN(call, cleanup)                                           \
    do {                                                                        \
        hipError_t e__ = (call);                                                \
        if (e__ != hipSuccess) {                                                \
            const int rc__ = ::csdr::hip_fail(e__, #call, __FILE__, __LINE__); \
            cleanup;                                                            \
            return rc__;                                                        \
        }                                                                       \
    } while (0)

// ---- diagnostics knobs ------------------------------------------------------
// The CSDR_* variables that change a launch plan (CSDR_RUN_MIN_TILES, CSDR_RUN_WEIGHTS, CSDR_AGC_L / _W, CSDR_WU, CSDR_RUN1024_V3, ...:
// DESIGN.md section 6.1; several of them change RESULTS) exist for A/B measurements and for the tests that force a kernel onto a small
// input.  They are read only when CSDR_DIAG=1 is set next to them: a production host does not inherit a knob from its environment by
// accident.  (Read when a handle is created or a plan is made, never inside a launch loop.)  Not gated: CSDR_QUIET (a print),
// CSDR_RCCL_LIB (where librccl lives), CSDR_LIB (which build of this library the Python binding loads).
inline const char *diag_env(const char *name)
{
    const char *d = getenv("CSDR_DIAG");
    return (d && d[0] == '1' && d[1] == 0) ? getenv(name) : nullptr;
}

// ---- host-side design (design.cpp) -----------------------------------------
// Kaiser prototype of firpfbch_crcf_create_kaiser(ANALYZER, M, m, As)
// (reference call: Liquid.chs:813).  Returns the M*2m taps the bank uses.
std::vector<float> design_pfb_taps(uint32_t M, uint32_t m, float As);
// msresamp_crcf(r, As) decomposition and filters (reference call: Liquid.chs:104, rate = bw/fs, As = 60)
struct ResampDesign {
    float rate = 0.f; double rho = 0.0; uint32_t K = 0;
    std::vector<uint32_t> m_hb; std::vector<std::vector<float>> h_hb;   // half-band stage s: 4 m + 1 taps
    uint32_t npfb = 256, m_arb = 7; float fc = 0.f; std::vector<float> pfb;   // [npfb][2 m_arb]
    uint64_t delta = 0;          // input samples per output at the arbitrary stage, Q32.32
};
ResampDesign design_msresamp(float rate, float As);
// nco_crcf_set_frequency's float -> uint32 phase-step conversion.
uint32_t nco_freq_word(float freq);
// Haskell-Float value of  -0.5*(M-1)/M*2*pi  (Liquid.chs:817).
float pfb_premix_freq(uint32_t M);
// cos/sin of one uint32 phase, evaluated like nco_crcf (VCO) does.
void nco_phasor(uint32_t theta, float *c, float *s);
// Period (in samples) of the phase sequence k*d_theta mod 2^32; 0 if > limit.
uint32_t nco_period(uint32_t d_theta, uint32_t limit);
// Smallest gain g for which rssi(g) = (float)(-20*log10(g)) is NOT above thr:
// squelch "threshold exceeded"  <=>  g < returned value.
float agc_gain_threshold(float threshold_db);

// ---- device-side parameter blocks ------------------------------------------
constexpr int DC_THREADS = 256;
constexpr int DC_PER_THREAD = 8;
constexpr int DC_BLOCK = DC_THREADS * DC_PER_THREAD;   // samples per scan block

struct DcParams {
    float a1;              // -(1-alpha)        (v0 = x - a1*v1)
    float beta;            // 1-alpha as f32 = -a1
    float beta_pow_thr[9]; // beta^(DC_PER_THREAD * 2^i), i=0..8
    double beta_blk;       // beta^DC_BLOCK (f64)
    float log2_beta;
};

struct NcoParams {
    uint32_t theta0;       // phase of the first sample of this call
    uint32_t d_theta;
    uint32_t tab_len;      // 0: evaluate sincos on device; else period of the table
    uint32_t tab_pos;      // index of the first sample of this call in the table
    int      up;           // 1: multiply by v, 0: by conj(v)
};

struct AgcParams {
    float alpha;           // 0.1
    float g_thr;           // exceeded <=> g < g_thr
    uint32_t timeout;      // 1000
};

// per-channel AGC state as kept on the device between chunks
struct AgcState { float g, y2; int32_t mode; uint32_t timer; };

// ---- generic kernels (kernels_generic.hip) ----------------------------------
// DC blocker [+ NCO mix] of n samples: y[i] = mix(dcblock(x[i])).  `state` is the
// device-resident v1 (float2), updated in place.  scratch: >= 2*ceil(n/DC_BLOCK)+2 float2.
int launch_dc_mix(const float2 *x, float2 *y, uint32_t n, bool do_dc, const DcParams &dc,
                  float2 *state, float2 *scratch, bool do_mix, const NcoParams &nco,
                  const float2 *nco_tab, hipStream_t s);
// X[t][j] = sum_n h[(M-1-j)+n*M] * u[(t-n)*M + j], u points at the first NEW sample and
// has (p-1)*M samples of history in front of it.
int launch_pfb_fir(const float2 *u, const float *taps, float2 *X, uint32_t M, uint32_t p,
                   uint32_t nf, hipStream_t s);
// forward M-point DFT of every frame: Y[t][k].  tw: e^{-j 2 pi i/M}, i<M.
int launch_dft(const float2 *X, float2 *Y, const float2 *tw, uint32_t M, uint32_t nf, hipStream_t s);
// interleaved channel shard g of G: Z[t][j1] = W_M^{j1 g} sum_{j2 < G} X[t][j1 + (M/G) j2] W_G^{j2 g}; the (M/G)-point DFT of Z[t]
// is Y[t][g + G m].  ph: G phasors W_G^{j2 g} followed by M/G phasors W_M^{j1 g}
int launch_fold(const float2 *X, float2 *Z, const float2 *ph, uint32_t M, uint32_t G, uint32_t nf, hipStream_t s);
// out[t] = sum_k DFT(X[t])[k] in k_mix_frames' summation order, without materialising Y (M = 1024, 4096, all channels)
bool dft_mix_supported(uint32_t M);
int launch_dft_mix(const float2 *X, float2 *out, const float2 *tw, uint32_t M, uint32_t nf, hipStream_t s);
// fused FIR + DFT + transpose [+ freqdem] for M = 1024 (kernels_pfb1024.hip): u_new as for launch_pfb_fir; out = channel-major
// [C][nf] CF32, or F32 with fm; scratch: 2 * min(max_runs, nf/32) * 1024 float2
bool pfb1024_supported(uint32_t M, uint32_t p);
int launch_pfb1024(const float2 *u_new, const float *taps, const float2 *tw, void *out, bool fm, uint32_t nf, uint32_t c0, uint32_t C,
                   float ref, const float2 *rp_in, float2 *rp_out, float2 *scratch, uint32_t max_runs, hipStream_t s);
// Z[c][t] = Y[t][c0 + c]  for c < C
int launch_transpose(const float2 *Y, float2 *Z, uint32_t M, uint32_t nf, uint32_t c0, uint32_t C,
                     hipStream_t s);
// per-channel AGC + squelch mute, in place on Z[C][nf]
int launch_agc(float2 *Z, uint32_t C, uint32_t nf, AgcState *st, const AgcParams &p, hipStream_t s);
// F[c][t] = arg(conj(prev)*Z[c][t]) * ref ; rp_in/rp_out: per-channel r'
int launch_fm(const float2 *Z, float *F, uint32_t C, uint32_t nf, float ref, const float2 *rp_in,
              float2 *rp_out, hipStream_t s);
// out[i] = ((in[0][i] + in[1][i]) + ...) + in[C-1][i], row length E floats
int launch_mix(const float *in, float *out, uint32_t C, uint32_t E, hipStream_t s);
int launch_agc_init(AgcState *st, uint32_t C, hipStream_t s);
// out row k = in row (C - k) mod C (CSDR_FLAG_DFT_BACKWARD: the analyzer's transform taken as e^{+j}); in != out
int launch_rows_reversed(const void *in, void *out, uint32_t C, size_t row_bytes, hipStream_t s);
// frame-major tails: freqdem straight from Y[nf][M] into channel-major F, or mixed over channels per frame
int launch_transpose_fm(const float2 *Y, float *F, uint32_t M, uint32_t nf, uint32_t c0, uint32_t C, float ref,
                        const float2 *rp_in, float2 *rp_out, hipStream_t s);
int launch_mix_frames(const float2 *Y, void *out, bool fm, uint32_t M, uint32_t nf, uint32_t c0, uint32_t C, float ref,
                      const float2 *rp_in, float2 *rp_out, hipStream_t s);

// ---- ampmodem DSB peak detector (kernels_am.hip): F[c][t] = 2 (|Z| - q_hat), q_hat a one-pole smoother per channel;
// q_in / q_out must be different arrays (ping-pong)
int launch_am(const float2 *Z, float *F, uint32_t C, uint32_t nf, const float *q_in, float *q_out, float alpha, hipStream_t s);

// ---- multi-stage resampler kernels (kernels_resamp.hip) ----
// y[j] = sum_i h[i] w[base0 + 2 j - i], i <= 4m (half-band decimator over a history-prefixed buffer)
int launch_hb_decim(const float2 *w, const float *h, float2 *y, uint32_t ny, uint32_t base0, uint32_t m, hipStream_t s);
// y[k] = (1-mu) F_b(n) + mu F_{b+1}(n) at t = t_first + k delta (Q32.32 over buffer positions)
int launch_resamp_arb(const float2 *w, const float *pfb, float2 *y, uint32_t ny, uint64_t t_first, uint64_t delta, uint32_t npfb,
                      uint32_t P, hipStream_t s);
// w[0..H) <- w[n..n+H)  (keep the last H samples of a history-prefixed buffer; H <= 1024)
int launch_keep_tail(float2 *w, uint32_t H, uint32_t n, hipStream_t s);

// ---- WBFM audio tail (kernels_wbfm.hip) ----
// one direct-form-II section y = b0 v0 + b1 v1 + b2 v2, v0 = x - a1 v1 - a2 v2; pw[k] = A^(16 * 2^k), A = [[-a1,-a2],[1,0]] row-major
struct BiquadParams { float b0, b1, b2, a1, a2; double pw[8][4]; };
BiquadParams design_butter2_lowpass(float fc);                   // iirFilter 2 fc 0 10 10 (Liquid.chs:636-638)
std::vector<float> design_firdecim_kaiser(uint32_t M, uint32_t m, float As);   // firdecim_rrrf_create_kaiser (Liquid.chs:487)
// rows X[C][nf] -> Y[C][nf] (may alias); per-channel state (v1, v2): st_in != st_out
int launch_biquad(const float *X, float *Y, uint32_t C, uint32_t nf, const BiquadParams &p, const float2 *st_in, float2 *st_out,
                  hipStream_t s);
// rows X[C][nf] (nf % M == 0) -> out[C][nf/M]; per-channel history of h_len-1 samples: hist_in != hist_out
int launch_firdecim(const float *X, float *out, uint32_t C, uint32_t nf, uint32_t M, const float *h, uint32_t h_len,
                    const float *hist_in, float *hist_out, hipStream_t s);

// ---- time-parallel exact AGC [+ freqdem] tail (kernels_agc_tail.hip) ----
struct AgcTailPlan;
int agc_tail_create(uint32_t C, uint32_t max_nf, AgcTailPlan **out);
void agc_tail_destroy(AgcTailPlan *p);
// Z[C][nf] -> out[C][nf] (CF32, or F32 when fm); st (and rp_in -> rp_out when fm) carry the per-channel state
// tm: Z is a TILE-MAJOR plane -- sample (c, t) at ((t >> 4) C + c) 16 + (t & 15), agc_tail_tm_guard(C) readable elements in front of
// Z and behind the plane -- which the fused M = 256 run kernels write for calls agc_tail_tm_supported() accepts (k_agc_spec_tm)
int agc_tail_process(AgcTailPlan *p, const float2 *Z, void *out, bool fm, uint32_t nf, AgcState *st, const AgcParams &prm,
                     float fm_ref, const float2 *rp_in, float2 *rp_out, hipStream_t s, bool tm = false);
bool agc_tail_tm_supported(const AgcTailPlan *p, uint32_t nf);
size_t agc_tail_tm_guard(uint32_t C);
uint32_t agc_tail_tm_calls(const AgcTailPlan *p);
int agc_tail_stats(AgcTailPlan *p, unsigned *checked, unsigned *redone);
void agc_tail_reset(AgcTailPlan *p);     // the AGC state was re-initialised: the next call pilots (kernels_agc_tail.hip)

// ---- hipEvent bracket around the dominant kernel (CSDR_FLAG_TIME_KERNELS) ----
struct KernelTimer {
    std::vector<hipEvent_t> ev;          // pairs
    size_t used = 0;
    double acc_ms = 0.0; uint32_t launches = 0;
    bool enabled = false;
    // region mode (CSDR_FLAG_TIME_REGION): ONE event in front of the first timed launch and one behind the last, recorded when the
    // total is read; total / launches = the launch cadence of back-to-back calls (kernel + the gap to the next one).  A pair of
    // events around every launch costs the stream 10-20 us per launch on this runtime, which the per-launch mode pays and reports.
    bool region = false, open = false;
    hipEvent_t r0 = nullptr, r1 = nullptr; hipStream_t last = nullptr; uint32_t pending = 0;
    int begin(hipStream_t s) {
        if (!enabled) return 0;
        if (region) {
            if (!open) {
                if (!r0) { CSDR_HIP(hipEventCreate(&r0)); CSDR_HIP(hipEventCreate(&r1)); }
                CSDR_HIP(hipEventRecord(r0, s));
                open = true; pending = 0;
            }
            return 0;
        }
        if (used + 2 > ev.size()) {
            if (ev.size() >= 4096) { int r = drain(); if (r) return r; }
            else for (int i = 0; i < 2; i++) { hipEvent_t e; CSDR_HIP(hipEventCreate(&e)); ev.push_back(e); }
        }
        CSDR_HIP(hipEventRecord(ev[used], s));
        return 0;
    }
    int end(hipStream_t s) {
        if (!enabled) return 0;
        if (region) { last = s; pending++; return 0; }
        CSDR_HIP(hipEventRecord(ev[used + 1], s));
        used += 2;
        return 0;
    }
    int drain() {
        if (region) {
            if (open && pending) {
                CSDR_HIP(hipEventRecord(r1, last));
                CSDR_HIP(hipEventSynchronize(r1));
                float ms = 0.f;
                CSDR_HIP(hipEventElapsedTime(&ms, r0, r1));
                acc_ms += ms; launches += pending;
            }
            open = false; pending = 0;
            return 0;
        }
        for (size_t i = 0; i + 1 < used; i += 2) {
            CSDR_HIP(hipEventSynchronize(ev[i + 1]));
            float ms = 0.f;
            CSDR_HIP(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
            acc_ms += ms; launches++;
        }
        used = 0;
        return 0;
    }
    void destroy() { for (auto e : ev) (void)hipEventDestroy(e); ev.clear(); used = 0; if (r0) { (void)hipEventDestroy(r0); (void)hipEventDestroy(r1); r0 = r1 = nullptr; } open = false; }
};

// ---- fused kernels (kernels_fused.hip) --------------------------------------
struct FusedPlan;   // opaque per-handle plan
bool fused_supported(uint32_t M, uint32_t p);

}  // namespace csdr
