// C ABI of libcsdr_hip.so (see include/csdr.h).  Handle management, device buffers,
// stream state, and the per-chunk launch sequences.  Product code: no CPU fallback,
// nothing from oracle/.
#include "../../include/csdr.h"
#include "csdr_internal.h"
#include "fused.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>

namespace csdr {
const char *last_error();
}
using namespace csdr;

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------
namespace {

struct DevGuard {
    int prev = -1; bool ok = true;
    explicit DevGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (dev >= 0 && dev != prev && hipSetDevice(dev) != hipSuccess) ok = false;
    }
    ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

int check_device(int32_t want, int *out_dev)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device visible (hipGetDeviceCount: %s); libcsdr_hip has no CPU fallback",
                  e == hipSuccess ? "0 devices" : hipGetErrorString(e));
        return CSDR_ERR_NODEV;
    }
    int dev = want;
    if (dev < 0) { CSDR_HIP(hipGetDevice(&dev)); }
    if (dev >= n) { set_error("device %d out of range (%d visible)", dev, n); return CSDR_ERR_INVALID; }
    *out_dev = dev;
    return 0;
}

template <class T> int dev_alloc(T **p, size_t count)
{
    *p = nullptr;
    if (!count) count = 1;
    CSDR_HIP(hipMalloc((void **)p, count * sizeof(T)));
    return 0;
}

DcParams make_dc(float alpha)
{
    DcParams d;
    d.a1 = -1.0f + alpha;                 // iirfilt_crcf_create_dc_blocker
    d.beta = -d.a1;
    for (int i = 0; i < 9; i++) d.beta_pow_thr[i] = (float)std::pow((double)d.beta, (double)(DC_PER_THREAD << i));
    d.beta_blk = std::pow((double)d.beta, (double)DC_BLOCK);
    d.log2_beta = (float)std::log2((double)d.beta);
    return d;
}

}  // namespace

// ---------------------------------------------------------------------------
// handles
// ---------------------------------------------------------------------------
struct csdr_dcblock {
    int device; uint32_t max_n; DcParams dc;
    float2 *d_state = nullptr, *d_scratch = nullptr, *d_x = nullptr, *d_y = nullptr;
};
struct csdr_nco {
    int device; uint32_t max_n; uint32_t theta, d_theta;
    float2 *d_x = nullptr, *d_y = nullptr;
};
struct csdr_agc {
    int device; uint32_t C, max_n; AgcParams p; AgcState *d_st = nullptr; float2 *d_z = nullptr;
};
struct csdr_iirfilt {
    int device; uint32_t C, max_n; BiquadParams p; float2 *d_st[2] = {nullptr, nullptr}; int cur = 0; float *d_x = nullptr;
};
struct csdr_firdecim {
    int device; uint32_t C, max_n, M, h_len; float *d_h = nullptr, *d_hist[2] = {nullptr, nullptr}; int cur = 0;
    float *d_x = nullptr, *d_y = nullptr;
};
struct csdr_resamp {
    int device; uint32_t max_in; ResampDesign d;
    // stage s (s < K: half-band decimators; s == K: the arbitrary stage): history-prefixed input buffer
    std::vector<float2 *> d_buf; std::vector<float *> d_h; std::vector<uint32_t> H; std::vector<uint64_t> n_seen;
    float *d_pfb = nullptr; float2 *d_out = nullptr;
    uint64_t t_next = 0;         // Q32.32 absolute time (arbitrary-stage input samples) of the next output
    bool passthrough = false;    // rate 0: the reference's "no resampler" (Liquid.chs:100-103)
};
struct csdr_ampdem {
    int device; uint32_t C, max_n; float *d_q[2] = {nullptr, nullptr}; int cur = 0;
    float2 *d_z = nullptr; float *d_f = nullptr;
};
struct csdr_freqdem {
    int device; uint32_t C, max_n; float ref; float2 *d_rp[2] = {nullptr, nullptr}; int cur = 0;
    float2 *d_z = nullptr; float *d_f = nullptr;
};

struct csdr_chain {
    csdr_chain_cfg cfg;
    int device;
    uint32_t M, p, C, c0, max_nf; uint64_t max_nx;
    bool use_fused = false;
    std::string path;
    // design
    std::vector<float> taps;
    uint32_t theta = 0, d_theta = 0, tab_len = 0, tab_pos = 0;
    DcParams dc; AgcParams agc; float fm_ref = 0.f;
    // device memory
    float *d_taps = nullptr;
    float2 *d_tw = nullptr, *d_nco_tab = nullptr, *d_dcstate = nullptr, *d_scratch = nullptr;
    uint32_t G = 1;                  // chan_stride: interleaved shard g = c0 of G (generic route, pruned DFT)
    float2 *d_tw_g = nullptr, *d_fold_ph = nullptr, *d_fold = nullptr;   // (M/G)-point twiddles, fold phasors, folded frames
    bool dft_backward = false;       // CSDR_FLAG_DFT_BACKWARD: rows leave through d_perm and a row permutation k -> (M - k) mod M
    void *d_perm = nullptr;
    bool mix_identity = false;       // DeNo --mix over all channels: M * (branch-0 FIR) instead of bank + DFT + sum
    bool mix_identity_shard = false; // the same for an interleaved shard g of G: (M / G) * sum of the G surviving branches' FIRs (round 6)
    float2 *d_u0 = nullptr, *d_u0hist = nullptr;     // branch-0 samples of the call behind p - 1 of history; history between calls (two copies, ping-pong)
    int u0_cur = 0;
    float2 *d_u = nullptr, *d_hist_tmp = nullptr, *d_A = nullptr, *d_B = nullptr;
    size_t a_guard = 0;              // fused M = 256 chain with the AGC tail: d_A has this many readable elements in front of and behind the plane (tile-major route)
    AgcState *d_agc = nullptr;
    float2 *d_rp[2] = {nullptr, nullptr}; int rp_cur = 0;
    // host-API staging: CSDR_CHAIN_INFLIGHT slots (device in / out, page-locked host in / out), three streams
    struct HostSlot {
        float2 *d_in = nullptr; void *d_out = nullptr; void *h_in = nullptr, *h_out = nullptr;
        hipEvent_t e_in = nullptr, e_k = nullptr, e_out = nullptr;
        void *user_out = nullptr; uint32_t n_out = 0; size_t out_bytes = 0; bool staged_out = false;
    };
    HostSlot slot[CSDR_CHAIN_INFLIGHT];
    hipStream_t s_in = nullptr, s_k = nullptr, s_out = nullptr;
    uint32_t q_head = 0, q_count = 0;      // oldest pending slot, number of pending chunks
    FusedPlan *fused = nullptr;
    SmallPlan *small = nullptr;
    BigPlan *big = nullptr;          // M = 1024 run kernel
    HugePlan *huge = nullptr;        // M = 4096: branch-tiled front kernel + 1024-point back kernel (kernels_pfb4096.hip)
    DcTilePlan *dctile = nullptr;   // generic path with the DC blocker: single-pass scan kernel
    uint32_t n_cus = 256;            // compute units of the device (run count of the fused M = 1024 kernel)
    AgcTailPlan *agc_tail = nullptr; // AGC on: time-parallel verified tail (unless CSDR_FLAG_AGC_SEQUENTIAL)
    // DeAM: the chain runs as DeNo into d_amz, then the ampmodem peak detector (kernels_am.hip) [+ mix]
    bool tail_only = false;          // CSDR_FLAG_TAIL_ONLY: AGC [+ freqdem] [+ mix] on a channel-major CF32 plane
    bool am = false, am_mix = false;
    // DeWBFM: the chain runs as DeNBFM 0.6 into d_wbf, then de-emphasis + decimator (kernels_wbfm.hip) [+ mix]
    bool wbfm = false, wbfm_mix = false; uint32_t wb_decim = 4, wb_hlen = 0; BiquadParams wb_bq{};
    float *d_wbf = nullptr, *d_wbo = nullptr, *d_wbh = nullptr, *d_wbhist[2] = {nullptr, nullptr}; float2 *d_wbst[2] = {nullptr, nullptr}; int wb_cur = 0;
    float2 *d_amz = nullptr; float *d_amf = nullptr; float *d_amq[2] = {nullptr, nullptr}; int amq_cur = 0;
    KernelTimer timer;
    std::string timed_kernel;
    // pipelined device entry point (csdr_chain_submit_device): two handle-owned streams used alternately, so that consecutive
    // chunks' launches may overlap (the next launch's cold run starts fill the CUs the previous one has already left)
    hipStream_t s_pd[2] = {nullptr, nullptr};
    hipEvent_t e_pd_done[2] = {nullptr, nullptr}, e_pd_tail[3] = {nullptr, nullptr, nullptr}, e_serial = nullptr;
    bool pd_used[2] = {false, false}, pd_tail_rec = false, serial_pending = false, in_submit = false;
    uint32_t pd_count = 0, pd_indep_calls = 0;
    bool call_indep = false; hipEvent_t call_ev_tail = nullptr;     // handed to the fused plan by the call in progress
    // entry points share the handle's state: a call on a caller's stream is remembered so that the host-buffer entry point
    // (own stream s_k) can order itself behind it, and the other way round
    // (a handle-owned event recorded on the caller's stream behind the call's work: the caller may destroy the stream afterwards)
    hipEvent_t e_user = nullptr; bool user_stream_dirty = false;
    bool pd_last_indep = false;      // the previous csdr_chain_submit_device call ran as an independent launch
};

extern "C" {

const char *csdr_last_error(void) { return csdr::last_error(); }
const char *csdr_version(void) { return "csdr-hip gfx950 0.1"; }

int csdr_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---------------------------------------------------------------------------
// dcBlocker
// ---------------------------------------------------------------------------
int csdr_dcblock_create(float alpha, uint32_t max_samples, csdr_dcblock **out)
{
    if (!out || !(alpha > 0.f && alpha < 1.f)) { set_error("dcblock: bad arguments"); return CSDR_ERR_INVALID; }
    int dev; int r = check_device(-1, &dev); if (r) return r;
    csdr_dcblock *h = new (std::nothrow) csdr_dcblock();
    if (!h) return CSDR_ERR_NOMEM;
    h->device = dev; h->max_n = max_samples ? max_samples : 1u << 20; h->dc = make_dc(alpha);
    if ((r = dev_alloc(&h->d_state, 1)) || (r = dev_alloc(&h->d_scratch, 2 * (size_t)(h->max_n / DC_BLOCK + 2))) ||
        (r = dev_alloc(&h->d_x, h->max_n)) || (r = dev_alloc(&h->d_y, h->max_n))) { csdr_dcblock_destroy(h); return r; }
    CSDR_HIP_CLEAN(hipMemset(h->d_state, 0, sizeof(float2)), csdr_dcblock_destroy(h));
    *out = h;
    return CSDR_OK;
}

int csdr_dcblock_process_device(csdr_dcblock *h, const void *d_x, uint32_t n, void *d_y, void *stream)
{
    if (!h) { set_error("dcblock: null handle"); return CSDR_ERR_INVALID; }
    if (n > h->max_n) { set_error("dcblock: %u samples > max %u", n, h->max_n); return CSDR_ERR_SIZE; }
    NcoParams nco{};
    return launch_dc_mix((const float2 *)d_x, (float2 *)d_y, n, true, h->dc, h->d_state, h->d_scratch, false, nco,
                         nullptr, (hipStream_t)stream);
}

int csdr_dcblock_process(csdr_dcblock *h, const float *x, uint32_t n, float *y)
{
    if (!h || (n && (!x || !y))) { set_error("dcblock: null argument"); return CSDR_ERR_INVALID; }
    if (n > h->max_n) { set_error("dcblock: %u samples > max %u", n, h->max_n); return CSDR_ERR_SIZE; }
    if (!n) return CSDR_OK;
    DevGuard guard(h->device);
    if (!guard.ok) { set_error("dcblock: cannot select device %d", h->device); return CSDR_ERR_HIP; }
    CSDR_HIP(hipMemcpy(h->d_x, x, sizeof(float2) * n, hipMemcpyHostToDevice));
    int r = csdr_dcblock_process_device(h, h->d_x, n, h->d_y, nullptr);
    if (r) return r;
    CSDR_HIP(hipMemcpy(y, h->d_y, sizeof(float2) * n, hipMemcpyDeviceToHost));
    return CSDR_OK;
}

int csdr_dcblock_destroy(csdr_dcblock *h)
{
    if (!h) return CSDR_OK;
    (void)hipFree(h->d_state); (void)hipFree(h->d_scratch); (void)hipFree(h->d_x); (void)hipFree(h->d_y);
    delete h;
    return CSDR_OK;
}

// ---------------------------------------------------------------------------
// mixDown / mixUp
// ---------------------------------------------------------------------------
int csdr_nco_create(float freq, uint32_t max_samples, csdr_nco **out)
{
    if (!out || !std::isfinite(freq)) { set_error("nco: bad arguments"); return CSDR_ERR_INVALID; }
    int dev; int r = check_device(-1, &dev); if (r) return r;
    csdr_nco *h = new (std::nothrow) csdr_nco();
    if (!h) return CSDR_ERR_NOMEM;
    h->device = dev; h->max_n = max_samples ? max_samples : 1u << 20;
    h->theta = 0; h->d_theta = nco_freq_word(freq);
    if ((r = dev_alloc(&h->d_x, h->max_n)) || (r = dev_alloc(&h->d_y, h->max_n))) { csdr_nco_destroy(h); return r; }
    *out = h;
    return CSDR_OK;
}

static int nco_mix(csdr_nco *h, const float *x, uint32_t n, float *y, int up)
{
    if (!h || (n && (!x || !y))) { set_error("nco: null argument"); return CSDR_ERR_INVALID; }
    if (n > h->max_n) { set_error("nco: %u samples > max %u", n, h->max_n); return CSDR_ERR_SIZE; }
    if (!n) return CSDR_OK;
    DevGuard guard(h->device);
    if (!guard.ok) { set_error("nco: cannot select device %d", h->device); return CSDR_ERR_HIP; }
    CSDR_HIP(hipMemcpy(h->d_x, x, sizeof(float2) * n, hipMemcpyHostToDevice));
    NcoParams nco{}; nco.theta0 = h->theta; nco.d_theta = h->d_theta; nco.up = up;
    DcParams dc{};
    int r = launch_dc_mix(h->d_x, h->d_y, n, false, dc, nullptr, nullptr, true, nco, nullptr, nullptr);
    if (r) return r;
    CSDR_HIP(hipMemcpy(y, h->d_y, sizeof(float2) * n, hipMemcpyDeviceToHost));
    h->theta += n * h->d_theta;
    return CSDR_OK;
}
int csdr_nco_mix_down(csdr_nco *h, const float *x, uint32_t n, float *y) { return nco_mix(h, x, n, y, 0); }
int csdr_nco_mix_up(csdr_nco *h, const float *x, uint32_t n, float *y) { return nco_mix(h, x, n, y, 1); }
int csdr_nco_get_words(const csdr_nco *h, uint32_t *theta, uint32_t *d_theta)
{
    if (!h) return CSDR_ERR_INVALID;
    if (theta) *theta = h->theta;
    if (d_theta) *d_theta = h->d_theta;
    return CSDR_OK;
}
int csdr_nco_destroy(csdr_nco *h)
{
    if (!h) return CSDR_OK;
    (void)hipFree(h->d_x); (void)hipFree(h->d_y);
    delete h;
    return CSDR_OK;
}

// ---------------------------------------------------------------------------
// automaticGainControl (nchan instances)
// ---------------------------------------------------------------------------
static AgcParams make_agc(float thr_db)
{
    AgcParams p; p.alpha = 0.1f; p.g_thr = agc_gain_threshold(thr_db); p.timeout = 1000u;
    return p;
}

int csdr_agc_create(float threshold_db, uint32_t nchan, uint32_t max_samples, csdr_agc **out)
{
    if (!out || !nchan || !std::isfinite(threshold_db)) { set_error("agc: bad arguments"); return CSDR_ERR_INVALID; }
    int dev; int r = check_device(-1, &dev); if (r) return r;
    csdr_agc *h = new (std::nothrow) csdr_agc();
    if (!h) return CSDR_ERR_NOMEM;
    h->device = dev; h->C = nchan; h->max_n = max_samples ? max_samples : 4096; h->p = make_agc(threshold_db);
    if ((r = dev_alloc(&h->d_st, nchan)) || (r = dev_alloc(&h->d_z, (size_t)nchan * h->max_n))) { csdr_agc_destroy(h); return r; }
    if ((r = launch_agc_init(h->d_st, nchan, nullptr))) { csdr_agc_destroy(h); return r; }
    CSDR_HIP(hipDeviceSynchronize());
    *out = h;
    return CSDR_OK;
}
int csdr_agc_process(csdr_agc *h, const float *x, uint32_t n, float *y)
{
    if (!h || (n && (!x || !y))) { set_error("agc: null argument"); return CSDR_ERR_INVALID; }
    if (n > h->max_n) { set_error("agc: %u samples > max %u", n, h->max_n); return CSDR_ERR_SIZE; }
    if (!n) return CSDR_OK;
    DevGuard guard(h->device);
    if (!guard.ok) { set_error("agc: cannot select device %d", h->device); return CSDR_ERR_HIP; }
    const size_t bytes = sizeof(float2) * (size_t)h->C * n;
    CSDR_HIP(hipMemcpy(h->d_z, x, bytes, hipMemcpyHostToDevice));
    int r = launch_agc(h->d_z, h->C, n, h->d_st, h->p, nullptr);
    if (r) return r;
    CSDR_HIP(hipMemcpy(y, h->d_z, bytes, hipMemcpyDeviceToHost));
    return CSDR_OK;
}
int csdr_agc_destroy(csdr_agc *h)
{
    if (!h) return CSDR_OK;
    (void)hipFree(h->d_st); (void)hipFree(h->d_z);
    delete h;
    return CSDR_OK;
}

// ---------------------------------------------------------------------------
// fmDemodulator (nchan instances)
// ---------------------------------------------------------------------------
static float fm_ref_of(float kf) { return (float)(1.0 / (2.0 * 3.14159265358979323846 * (double)kf)); }

int csdr_freqdem_create(float kf, uint32_t nchan, uint32_t max_samples, csdr_freqdem **out)
{
    if (!out || !nchan || !(kf > 0.f)) { set_error("freqdem: bad arguments (kf must be > 0)"); return CSDR_ERR_INVALID; }
    int dev; int r = check_device(-1, &dev); if (r) return r;
    csdr_freqdem *h = new (std::nothrow) csdr_freqdem();
    if (!h) return CSDR_ERR_NOMEM;
    h->device = dev; h->C = nchan; h->max_n = max_samples ? max_samples : 4096; h->ref = fm_ref_of(kf);
    if ((r = dev_alloc(&h->d_rp[0], nchan)) || (r = dev_alloc(&h->d_rp[1], nchan)) ||
        (r = dev_alloc(&h->d_z, (size_t)nchan * h->max_n)) || (r = dev_alloc(&h->d_f, (size_t)nchan * h->max_n))) {
        csdr_freqdem_destroy(h); return r;
    }
    CSDR_HIP_CLEAN(hipMemset(h->d_rp[0], 0, sizeof(float2) * nchan), csdr_freqdem_destroy(h));
    CSDR_HIP_CLEAN(hipMemset(h->d_rp[1], 0, sizeof(float2) * nchan), csdr_freqdem_destroy(h));
    *out = h;
    return CSDR_OK;
}
int csdr_freqdem_process(csdr_freqdem *h, const float *x, uint32_t n, float *m)
{
    if (!h || (n && (!x || !m))) { set_error("freqdem: null argument"); return CSDR_ERR_INVALID; }
    if (n > h->max_n) { set_error("freqdem: %u samples > max %u", n, h->max_n); return CSDR_ERR_SIZE; }
    if (!n) return CSDR_OK;
    DevGuard guard(h->device);
    if (!guard.ok) { set_error("freqdem: cannot select device %d", h->device); return CSDR_ERR_HIP; }
    CSDR_HIP(hipMemcpy(h->d_z, x, sizeof(float2) * (size_t)h->C * n, hipMemcpyHostToDevice));
    int r = launch_fm(h->d_z, h->d_f, h->C, n, h->ref, h->d_rp[h->cur], h->d_rp[h->cur ^ 1], nullptr);
    if (r) return r;
    h->cur ^= 1;
    CSDR_HIP(hipMemcpy(m, h->d_f, sizeof(float) * (size_t)h->C * n, hipMemcpyDeviceToHost));
    return CSDR_OK;
}
int csdr_freqdem_destroy(csdr_freqdem *h)
{
    if (!h) return CSDR_OK;
    (void)hipFree(h->d_rp[0]); (void)hipFree(h->d_rp[1]); (void)hipFree(h->d_z); (void)hipFree(h->d_f);
    delete h;
    return CSDR_OK;
}

// ---------------------------------------------------------------------------
// iirFilter n fc f0 ap as (Liquid.chs:629-638), firDecimator m (Liquid.chs:485-501)
// ---------------------------------------------------------------------------
int csdr_iirfilt_create(uint32_t order, float fc, float f0, float ap, float as_db, uint32_t nchan, uint32_t max_samples, csdr_iirfilt **out)
{
    (void)f0; (void)ap; (void)as_db;
    if (!out || !nchan || !(fc > 0.f && fc < 0.5f)) { set_error("iirfilt: bad arguments (fc in (0, 0.5))"); return CSDR_ERR_INVALID; }
    if (order != 2) { set_error("iirfilt: only the reference's order-2 Butterworth low-pass is built (order %u)", order); return CSDR_ERR_INVALID; }
    int dev; int r = check_device(-1, &dev); if (r) return r;
    csdr_iirfilt *h = new (std::nothrow) csdr_iirfilt();
    if (!h) return CSDR_ERR_NOMEM;
    h->device = dev; h->C = nchan; h->max_n = max_samples ? max_samples : 4096; h->p = design_butter2_lowpass(fc);
    if (hipMalloc(&h->d_x, sizeof(float) * (size_t)nchan * h->max_n) != hipSuccess || hipMalloc(&h->d_st[0], sizeof(float2) * nchan) != hipSuccess ||
        hipMalloc(&h->d_st[1], sizeof(float2) * nchan) != hipSuccess) { set_error("iirfilt: device allocation failed"); csdr_iirfilt_destroy(h); return CSDR_ERR_HIP; }
    CSDR_HIP_CLEAN(hipMemset(h->d_st[0], 0, sizeof(float2) * nchan), csdr_iirfilt_destroy(h)); CSDR_HIP_CLEAN(hipMemset(h->d_st[1], 0, sizeof(float2) * nchan), csdr_iirfilt_destroy(h));
    *out = h;
    return CSDR_OK;
}
int csdr_iirfilt_process(csdr_iirfilt *h, const float *x, uint32_t n, float *y)
{
    if (!h || (n && (!x || !y))) { set_error("iirfilt: null argument"); return CSDR_ERR_INVALID; }
    if (n > h->max_n) { set_error("iirfilt: %u samples > max %u", n, h->max_n); return CSDR_ERR_SIZE; }
    if (!n) return CSDR_OK;
    DevGuard guard(h->device);
    CSDR_HIP(hipMemcpy(h->d_x, x, sizeof(float) * (size_t)h->C * n, hipMemcpyHostToDevice));
    int r = launch_biquad(h->d_x, h->d_x, h->C, n, h->p, h->d_st[h->cur], h->d_st[h->cur ^ 1], nullptr);
    if (r) return r;
    h->cur ^= 1;
    CSDR_HIP(hipMemcpy(y, h->d_x, sizeof(float) * (size_t)h->C * n, hipMemcpyDeviceToHost));
    return CSDR_OK;
}
int csdr_iirfilt_destroy(csdr_iirfilt *h)
{
    if (!h) return CSDR_OK;
    (void)hipFree(h->d_x); (void)hipFree(h->d_st[0]); (void)hipFree(h->d_st[1]);
    delete h;
    return CSDR_OK;
}
int csdr_firdecim_create(uint32_t decim, uint32_t nchan, uint32_t max_samples, csdr_firdecim **out)
{
    if (!out || !nchan || decim < 1 || decim > 4096) { set_error("firdecim: bad arguments"); return CSDR_ERR_INVALID; }
    int dev; int r = check_device(-1, &dev); if (r) return r;
    csdr_firdecim *h = new (std::nothrow) csdr_firdecim();
    if (!h) return CSDR_ERR_NOMEM;
    const std::vector<float> taps = design_firdecim_kaiser(decim, 10, 60.0f);          // firdecimCreate, Liquid.chs:485-490
    h->device = dev; h->C = nchan; h->max_n = max_samples ? max_samples : 4096; h->M = decim; h->h_len = (uint32_t)taps.size();
    const size_t hist = (size_t)nchan * (h->h_len - 1);
    if (hipMalloc(&h->d_x, sizeof(float) * (size_t)nchan * h->max_n) != hipSuccess || hipMalloc(&h->d_y, sizeof(float) * (size_t)nchan * (h->max_n / decim + 1)) != hipSuccess ||
        hipMalloc(&h->d_h, sizeof(float) * taps.size()) != hipSuccess || hipMalloc(&h->d_hist[0], sizeof(float) * hist) != hipSuccess ||
        hipMalloc(&h->d_hist[1], sizeof(float) * hist) != hipSuccess) { set_error("firdecim: device allocation failed"); csdr_firdecim_destroy(h); return CSDR_ERR_HIP; }
    CSDR_HIP_CLEAN(hipMemcpy(h->d_h, taps.data(), sizeof(float) * taps.size(), hipMemcpyHostToDevice), csdr_firdecim_destroy(h));
    CSDR_HIP_CLEAN(hipMemset(h->d_hist[0], 0, sizeof(float) * hist), csdr_firdecim_destroy(h)); CSDR_HIP_CLEAN(hipMemset(h->d_hist[1], 0, sizeof(float) * hist), csdr_firdecim_destroy(h));
    *out = h;
    return CSDR_OK;
}
int csdr_firdecim_process(csdr_firdecim *h, const float *x, uint32_t n, float *y)
{
    if (!h || (n && (!x || !y))) { set_error("firdecim: null argument"); return CSDR_ERR_INVALID; }
    if (n > h->max_n) { set_error("firdecim: %u samples > max %u", n, h->max_n); return CSDR_ERR_SIZE; }
    if (n % h->M) { set_error("firdecim: %u samples are not a multiple of the decimation %u (Liquid.chs:495-497)", n, h->M); return CSDR_ERR_SIZE; }
    if (!n) return CSDR_OK;
    DevGuard guard(h->device);
    CSDR_HIP(hipMemcpy(h->d_x, x, sizeof(float) * (size_t)h->C * n, hipMemcpyHostToDevice));
    int r = launch_firdecim(h->d_x, h->d_y, h->C, n, h->M, h->d_h, h->h_len, h->d_hist[h->cur], h->d_hist[h->cur ^ 1], nullptr);
    if (r) return r;
    h->cur ^= 1;
    CSDR_HIP(hipMemcpy(y, h->d_y, sizeof(float) * (size_t)h->C * (n / h->M), hipMemcpyDeviceToHost));
    return CSDR_OK;
}
int csdr_firdecim_destroy(csdr_firdecim *h)
{
    if (!h) return CSDR_OK;
    (void)hipFree(h->d_x); (void)hipFree(h->d_y); (void)hipFree(h->d_h); (void)hipFree(h->d_hist[0]); (void)hipFree(h->d_hist[1]);
    delete h;
    return CSDR_OK;
}

// ---------------------------------------------------------------------------
// resampler r as (Liquid.chs:56-117)
// ---------------------------------------------------------------------------
int csdr_resamp_create(float rate, float As, uint32_t max_in, csdr_resamp **out)
{
    if (!out || rate < 0.f || !(rate == rate)) { set_error("resamp: bad arguments"); return CSDR_ERR_INVALID; }
    if (rate > 2.0f) { set_error("resamp: rate %g > 2 (interpolating half-band stages are not built)", rate); return CSDR_ERR_INVALID; }
    int dev; int r = check_device(-1, &dev); if (r) return r;
    csdr_resamp *h = new (std::nothrow) csdr_resamp();
    if (!h) return CSDR_ERR_NOMEM;
    h->device = dev; h->max_in = max_in ? max_in : 4096;
    if (rate == 0.f) { h->passthrough = true; *out = h; return CSDR_OK; }
    h->d = design_msresamp(rate, As);
    const uint32_t K = h->d.K, P = 2 * h->d.m_arb;
    h->d_buf.assign(K + 1, nullptr); h->d_h.assign(K, nullptr); h->H.assign(K + 1, 0); h->n_seen.assign(K + 1, 0);
    auto fail = [&](int rc) { csdr_resamp_destroy(h); return rc; };
    uint32_t cap = h->max_in;
    for (uint32_t s = 0; s <= K; s++) {
        h->H[s] = s < K ? 4 * h->d.m_hb[s] + 1 : P + 1;
        if ((r = dev_alloc(&h->d_buf[s], (size_t)h->H[s] + cap + 2))) return fail(r);
        CSDR_HIP_CLEAN(hipMemset(h->d_buf[s], 0, sizeof(float2) * ((size_t)h->H[s] + cap + 2)), csdr_resamp_destroy(h));
        if (s < K) {
            if (hipMalloc(&h->d_h[s], sizeof(float) * h->d.h_hb[s].size()) != hipSuccess) { set_error("resamp: allocation failed"); return fail(CSDR_ERR_HIP); }
            CSDR_HIP_CLEAN(hipMemcpy(h->d_h[s], h->d.h_hb[s].data(), sizeof(float) * h->d.h_hb[s].size(), hipMemcpyHostToDevice), csdr_resamp_destroy(h));
            cap = cap / 2 + 1;
        }
    }
    if (hipMalloc(&h->d_pfb, sizeof(float) * h->d.pfb.size()) != hipSuccess) { set_error("resamp: allocation failed"); return fail(CSDR_ERR_HIP); }
    CSDR_HIP_CLEAN(hipMemcpy(h->d_pfb, h->d.pfb.data(), sizeof(float) * h->d.pfb.size(), hipMemcpyHostToDevice), csdr_resamp_destroy(h));
    if ((r = dev_alloc(&h->d_out, (size_t)csdr_resamp_max_out(h, h->max_in)))) return fail(r);
    if (!getenv("CSDR_QUIET")) {
        // what msresamp_crcf_print shows in the reference ("Using resampler:", Liquid.chs:105-106)
        printf("csdr resampler: rate=%g = 2^-%u x %g, half-band taps:", rate, K, h->d.rho);
        for (uint32_t s = 0; s < K; s++) printf(" %u", 4 * h->d.m_hb[s] + 1);
        printf(", arbitrary stage: npfb=%u m=%u fc=%g As=%g\n", h->d.npfb, h->d.m_arb, h->d.fc, As);
        fflush(stdout);
    }
    *out = h;
    return CSDR_OK;
}
float csdr_resamp_get_rate(const csdr_resamp *h) { return h ? (h->passthrough ? 1.0f : h->d.rate) : 0.f; }
uint32_t csdr_resamp_max_out(const csdr_resamp *h, uint32_t n_in)
{
    if (!h) return 0;
    if (h->passthrough) return n_in;
    return 2u * (uint32_t)std::ceil((double)h->d.rate * n_in) + 2u;      // the reference's 2*ceil(r*nx) (Liquid.chs:81)
}
int csdr_resamp_process_device(csdr_resamp *h, const void *d_x, uint32_t n_in, void *d_y, uint32_t *n_out, void *stream)
{
    if (!h || !n_out) { set_error("resamp: null argument"); return CSDR_ERR_INVALID; }
    *n_out = 0;
    if (!n_in) return CSDR_OK;
    if (!d_x || !d_y) { set_error("resamp: null buffer"); return CSDR_ERR_INVALID; }
    if (n_in > h->max_in) { set_error("resamp: %u samples > max %u", n_in, h->max_in); return CSDR_ERR_SIZE; }
    DevGuard guard(h->device);
    hipStream_t s = (hipStream_t)stream;
    if (h->passthrough) {
        CSDR_HIP(hipMemcpyAsync(d_y, d_x, sizeof(float2) * (size_t)n_in, hipMemcpyDeviceToDevice, s));
        *n_out = n_in;
        return CSDR_OK;
    }
    const uint32_t K = h->d.K, P = 2 * h->d.m_arb;
    int r;
    if ((const void *)(h->d_buf[0] + h->H[0]) != d_x)
        CSDR_HIP(hipMemcpyAsync(h->d_buf[0] + h->H[0], d_x, sizeof(float2) * (size_t)n_in, hipMemcpyDeviceToDevice, s));
    uint32_t n = n_in;
    for (uint32_t st = 0; st < K; st++) {
        // stage input: absolute samples [N0 - H, N0 + n) sit at buffer positions [0, H + n); output j = sum h[i] x[2j+1-i]
        const uint64_t N0 = h->n_seen[st], N1 = N0 + n;
        const uint32_t ny = (uint32_t)(N1 / 2 - N0 / 2);
        const uint32_t base0 = (uint32_t)((2 * (N0 / 2) + 1) - N0 + h->H[st]);
        float2 *dst = h->d_buf[st + 1] + h->H[st + 1];
        if ((r = launch_hb_decim(h->d_buf[st], h->d_h[st], dst, ny, base0, h->d.m_hb[st], s))) return r;
        if ((r = launch_keep_tail(h->d_buf[st], h->H[st], n, s))) return r;
        h->n_seen[st] = N1;
        n = ny;
    }
    {
        const uint64_t N0 = h->n_seen[K], N1 = N0 + n;
        uint32_t ny = 0;
        if (N1 >= 2) {
            // outputs while floor(t) + 1 <= N1 - 1, i.e. t < (N1 - 1) * 2^32
            const uint64_t lim = (N1 - 1) << 32;
            if (h->t_next < lim) ny = (uint32_t)((lim - h->t_next + h->d.delta - 1) / h->d.delta);
        }
        // buffer position of absolute sample a is a - (N0 - H)
        const uint64_t t_first = h->t_next - (N0 << 32) + ((uint64_t)h->H[K] << 32);
        if ((r = launch_resamp_arb(h->d_buf[K], h->d_pfb, (float2 *)d_y, ny, t_first, h->d.delta, h->d.npfb, P, s))) return r;
        if ((r = launch_keep_tail(h->d_buf[K], h->H[K], n, s))) return r;
        h->t_next += (uint64_t)ny * h->d.delta;
        h->n_seen[K] = N1;
        *n_out = ny;
    }
    return CSDR_OK;
}
int csdr_resamp_process(csdr_resamp *h, const float *x, uint32_t n_in, float *y, uint32_t *n_out)
{
    if (!h || !n_out) { set_error("resamp: null argument"); return CSDR_ERR_INVALID; }
    *n_out = 0;
    if (!n_in) return CSDR_OK;
    if (!x || !y) { set_error("resamp: null buffer"); return CSDR_ERR_INVALID; }
    if (n_in > h->max_in) { set_error("resamp: %u samples > max %u", n_in, h->max_in); return CSDR_ERR_SIZE; }
    if (h->passthrough) { memcpy(y, x, sizeof(float2) * (size_t)n_in); *n_out = n_in; return CSDR_OK; }
    DevGuard guard(h->device);
    // the first stage's buffer doubles as the H2D landing area
    float2 *stage0 = h->d_buf[0] + h->H[0];
    CSDR_HIP(hipMemcpy(stage0, x, sizeof(float2) * (size_t)n_in, hipMemcpyHostToDevice));
    // process_device copies d_x into the same place: skip that by handing it the landing area itself
    uint32_t no = 0;
    int r = csdr_resamp_process_device(h, stage0, n_in, h->d_out, &no, nullptr);
    if (r) return r;
    CSDR_HIP(hipMemcpy(y, h->d_out, sizeof(float2) * (size_t)no, hipMemcpyDeviceToHost));
    *n_out = no;
    return CSDR_OK;
}
int csdr_resamp_destroy(csdr_resamp *h)
{
    if (!h) return CSDR_OK;
    for (float2 *p : h->d_buf) if (p) (void)hipFree(p);
    for (float *p : h->d_h) if (p) (void)hipFree(p);
    if (h->d_pfb) (void)hipFree(h->d_pfb);
    if (h->d_out) (void)hipFree(h->d_out);
    delete h;
    return CSDR_OK;
}

// ---------------------------------------------------------------------------
// amDemodulator (Liquid.chs:439-469)
// ---------------------------------------------------------------------------
int csdr_ampdem_create(float mod_index, uint32_t nchan, uint32_t max_samples, csdr_ampdem **out)
{
    if (!out || !nchan || !(mod_index > 0.f)) { set_error("ampdem: bad arguments (mod_index must be > 0)"); return CSDR_ERR_INVALID; }
    int dev; int r = check_device(-1, &dev); if (r) return r;
    csdr_ampdem *h = new (std::nothrow) csdr_ampdem();
    if (!h) return CSDR_ERR_NOMEM;
    h->device = dev; h->C = nchan; h->max_n = max_samples ? max_samples : 4096;
    if ((r = dev_alloc(&h->d_z, (size_t)nchan * h->max_n)) || hipMalloc(&h->d_f, sizeof(float) * (size_t)nchan * h->max_n) != hipSuccess ||
        hipMalloc(&h->d_q[0], sizeof(float) * nchan) != hipSuccess || hipMalloc(&h->d_q[1], sizeof(float) * nchan) != hipSuccess) {
        set_error("ampdem: device allocation failed");
        csdr_ampdem_destroy(h); return r ? r : CSDR_ERR_HIP;
    }
    CSDR_HIP_CLEAN(hipMemset(h->d_q[0], 0, sizeof(float) * nchan), csdr_ampdem_destroy(h));
    CSDR_HIP_CLEAN(hipMemset(h->d_q[1], 0, sizeof(float) * nchan), csdr_ampdem_destroy(h));
    *out = h;
    return CSDR_OK;
}
int csdr_ampdem_process(csdr_ampdem *h, const float *x, uint32_t n, float *m)
{
    if (!h || (n && (!x || !m))) { set_error("ampdem: null argument"); return CSDR_ERR_INVALID; }
    if (n > h->max_n) { set_error("ampdem: %u samples > max %u", n, h->max_n); return CSDR_ERR_SIZE; }
    if (!n) return CSDR_OK;
    DevGuard guard(h->device);
    CSDR_HIP(hipMemcpy(h->d_z, x, sizeof(float2) * (size_t)h->C * n, hipMemcpyHostToDevice));
    int r = launch_am(h->d_z, h->d_f, h->C, n, h->d_q[h->cur], h->d_q[h->cur ^ 1], 0.01f, nullptr);
    if (r) return r;
    h->cur ^= 1;
    CSDR_HIP(hipMemcpy(m, h->d_f, sizeof(float) * (size_t)h->C * n, hipMemcpyDeviceToHost));
    return CSDR_OK;
}
int csdr_ampdem_destroy(csdr_ampdem *h)
{
    if (!h) return CSDR_OK;
    (void)hipFree(h->d_q[0]); (void)hipFree(h->d_q[1]); (void)hipFree(h->d_z); (void)hipFree(h->d_f);
    delete h;
    return CSDR_OK;
}

// ---------------------------------------------------------------------------
// fused chain
// ---------------------------------------------------------------------------
}  // extern "C"

// ---------------------------------------------------------------------------
// Route table: which plan and which kernels a chain configuration gets.  csdr_chain_create selects a row through
// route_select() and nothing else; csdr_route_table() prints the rows (README).  Per call the plan then picks
// between its run-sized and its chunk-sized kernel by the call's frame count (the thresholds are part of the row's text).
// ---------------------------------------------------------------------------
namespace {
enum RoutePlan { PLAN_FUSED256, PLAN_SMALL64, PLAN_BIG1024, PLAN_HUGE4096, PLAN_GENERIC };
constexpr uint32_t ST1 = 1u << 1, ST2 = 1u << 2, ST4 = 1u << 4, ST8 = 1u << 8, ST_ANY = ~0u;
struct RouteRow {
    uint32_t M;                 // channels the row is for; 0: any count
    uint32_t strides;           // bit g: chan_stride g accepted (bit 1: whole band or a contiguous shard)
    RoutePlan plan;
    const char *name, *run_sized, *other, *agc_tail;
};
const RouteRow ROUTES[] = {
    {256, ST1 | ST2 | ST4 | ST8, PLAN_FUSED256, "fused-256",
     "k_run256v2<FM | CF32[, G]> (whole band: calls of >= 1024 whole tiles of 16 frames; interleaved shards G = 2, 4, 8: every whole tile)",
     "k_tile256<FM | CF32> (look-back tile kernel: chunk-sized calls, ragged ends, and contiguous channel shards at every size) [+ k_shard_gather]",
     "channelizer -> CF32 plane (tile-major for run-sized calls of whole tiles) -> k_agc_spec_tm | k_agc_spec -> k_agc_fix [-> k_mix]"},
    {64, ST1, PLAN_SMALL64, "fused-k_run64",
     "k_run64v2 (CF32 output, whole band, nf % 64 == 0, >= 8 tiles of 64 frames per run)", "k_run64<FM | CF32> (FM output, shards, ragged calls)",
     "k_run64v2 -> CF32 plane (tile-major, run-sized calls) -> k_agc_spec_tm | k_run64<CF32> -> k_agc_spec; -> k_agc_fix [-> k_mix]"},
    {1024, ST1 | ST2 | ST4 | ST8, PLAN_BIG1024, "fused-k_run1024",
     "k_run1024v3<FM | CF32> (whole band, calls of whole 4-frame tiles; a call that ends inside a 128-byte line stores the front part of it); "
     "k_shard1024<FM | CF32, G> (interleaved shards G = 4, 8: fold of the aliasing branches behind the FIR + a (1024 / G)-point DFT across the lanes; run-sized calls of whole tiles); "
     "k_run1024v2<FM, 2> (interleaved shards G = 2, FM output)",
     "k_run1024<FM | CF32> (ragged calls, contiguous shards, the other calls of interleaved shards) [+ k_pfb1024_fixup, k_shard_gather1024]", "k_run1024v3<CF32> -> CF32 plane (tile-major) -> k_agc_spec_tm | k_run1024<CF32> -> k_agc_spec; -> k_agc_fix [-> k_mix]"},
    {4096, ST1, PLAN_HUGE4096, "fused-4096",
     "k_front4096 (DC blocker + pre-mix + FIR, branch-tiled: four sibling workgroups per frame, radix-4 split of the DFT on the registers) -> z (8 B / sample) "
     "-> k_back4096<CF32 | FM | FM,mix> (four 1024-point DFTs per frame, 16-frame blocks, whole 128-byte lines per row) [-> k_mix4096_finish]; whole band; "
     "DeNo --mix without the AGC keeps the any-M route's mix identity",
     "the same kernels (any call size; short calls use fewer runs)", "k_front4096 -> k_back4096<CF32> -> CF32 plane (tile-major: calls of whole 16-frame blocks, >= 4096 frames) -> k_agc_spec_tm | k_agc_spec -> k_agc_fix [-> k_mix]"},
    {0, ST_ANY, PLAN_GENERIC, "generic",
     "k_dc_tile -> k_pfb_fir (M = 1024, forced generic: k_pfb1024) -> k_fft_r16 | k_fft_pow2 | k_dft_direct [interleaved shard: k_fold + (M / G)-point DFT] "
     "-> k_transpose_fm | k_mix_frames | k_transpose;  DeNo --mix over all channels: k_dc_fold + k_mixid_finish (M % 4096 == 0) | k_dc_tile + k_branch0_fir "
     "(the sum of all bins of a frame is M x branch 0);  DeNo --mix of an interleaved shard (M % 4096 == 0, G = 2, 4, 8): k_dc_fold8 + k_mixid_shard_finish "
     "(the shard's channel sum is M / G x the phasor sum of the G surviving branches)",
     "the same kernels (any call size)", "... -> k_transpose -> k_agc_spec -> k_agc_fix [-> k_fm] [-> k_mix]"},
};
const RouteRow *route_select(uint32_t M, uint32_t p, uint32_t G, uint32_t flags)
{
    const RouteRow *generic = &ROUTES[sizeof(ROUTES) / sizeof(ROUTES[0]) - 1];
    if (M <= 1 || (flags & CSDR_FLAG_FORCE_GENERIC)) return generic;
    if (G > 1 && diag_env("CSDR_SHARD_GENERIC")) return generic;
    for (const RouteRow &r : ROUTES) {
        if (r.M != M || !(G < 32 && (r.strides >> G) & 1u)) continue;
        const bool ok = r.plan == PLAN_FUSED256 ? fused_supported(M, p) : r.plan == PLAN_SMALL64 ? small_supported(M, p)
                      : r.plan == PLAN_BIG1024 ? (big_supported(M, p) && !diag_env("CSDR_NO_RUN1024"))
                      : r.plan == PLAN_HUGE4096 ? (huge_supported(M, p) && !diag_env("CSDR_NO_RUN4096")) : true;
        if (ok) return &r;
    }
    return generic;
}
std::string route_table_text()
{
    std::string t;
    char buf[256];
    for (const RouteRow &r : ROUTES) {
        std::string st;
        if (r.strides == ST_ANY) st = "any";
        else for (uint32_t g : {1u, 2u, 4u, 8u}) if ((r.strides >> g) & 1u) st += (st.empty() ? "" : ", ") + std::to_string(g);
        snprintf(buf, sizeof buf, "[%s] channels = %s, chan_stride in {%s}\n", r.name, r.M ? std::to_string(r.M).c_str() : "any", st.c_str());
        t += buf;
        t += std::string("    run-sized calls : ") + r.run_sized + "\n    other calls     : " + r.other + "\n    AGC on          : " + r.agc_tail + "\n";
    }
    t += "[tail-only] CSDR_FLAG_TAIL_ONLY: channel-major CF32 plane -> k_agc_spec -> k_agc_fix [-> k_mix] (hybrid multi-GPU partition)\n";
    return t;
}
}  // namespace

extern "C" {

const char *csdr_route_table(void)
{
    static const std::string text = route_table_text();
    return text.c_str();
}

void csdr_chain_cfg_default(csdr_chain_cfg *cfg, uint32_t channels)
{
    if (!cfg) return;
    memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = sizeof(*cfg);
    cfg->channels = channels ? channels : 1;
    cfg->dc_block = 1;
    cfg->dc_alpha = 0.0005f;
    cfg->agc_threshold_db = 0.0f;
    cfg->demod = CSDR_DEMOD_NONE;
    cfg->kf = 0.3f;
    cfg->device = -1;
    cfg->max_frames = 4096;
    cfg->pfb_m = 7;
    cfg->pfb_as = 80.0f;
    cfg->wbfm_decim = 4;
    cfg->deemph_fc = 0.025f;
}

static int chain_init_state(csdr_chain *h, hipStream_t s)
{
    h->theta = 0; h->tab_pos = 0; h->rp_cur = 0;
    CSDR_HIP(hipMemsetAsync(h->d_dcstate, 0, sizeof(float2), s));
    if (h->d_u) CSDR_HIP(hipMemsetAsync(h->d_u, 0, sizeof(float2) * (size_t)(h->p - 1) * h->M, s));
    if (h->d_u0hist) { CSDR_HIP(hipMemsetAsync(h->d_u0hist, 0, sizeof(float2) * 2 * (h->p - 1) * (h->mix_identity_shard ? h->G : 1u), s)); h->u0_cur = 0; }
    if (h->d_agc) { int r = launch_agc_init(h->d_agc, h->C, s); if (r) return r; }
    if (h->agc_tail) agc_tail_reset(h->agc_tail);
    if (h->d_rp[0]) {
        CSDR_HIP(hipMemsetAsync(h->d_rp[0], 0, sizeof(float2) * h->C, s));
        CSDR_HIP(hipMemsetAsync(h->d_rp[1], 0, sizeof(float2) * h->C, s));
    }
    if (h->d_wbst[0]) {
        h->wb_cur = 0;
        for (int i = 0; i < 2; i++) {
            CSDR_HIP(hipMemsetAsync(h->d_wbst[i], 0, sizeof(float2) * h->C, s));
            CSDR_HIP(hipMemsetAsync(h->d_wbhist[i], 0, sizeof(float) * (size_t)h->C * (h->wb_hlen - 1), s));
        }
    }
    if (h->d_amq[0]) {
        h->amq_cur = 0;
        CSDR_HIP(hipMemsetAsync(h->d_amq[0], 0, sizeof(float) * h->C, s));     // ampmodem_reset: q_hat = 0
        CSDR_HIP(hipMemsetAsync(h->d_amq[1], 0, sizeof(float) * h->C, s));
    }
    if (h->fused) { int r = fused_reset(h->fused, s); if (r) return r; }
    if (h->small) { int r = small_reset(h->small, s); if (r) return r; }
    if (h->big) { int r = big_reset(h->big, s); if (r) return r; }
    if (h->huge) { int r = huge_reset(h->huge, s); if (r) return r; }
    if (h->dctile) { int r = dctile_reset(h->dctile, s); if (r) return r; }
    return 0;
}

int csdr_chain_create(const csdr_chain_cfg *cfg_in, csdr_chain **out)
{
    if (!cfg_in || !out) { set_error("chain: null argument"); return CSDR_ERR_INVALID; }
    if (cfg_in->struct_size != sizeof(csdr_chain_cfg)) { set_error("chain: cfg.struct_size %u != %zu", cfg_in->struct_size, sizeof(csdr_chain_cfg)); return CSDR_ERR_INVALID; }
    if (cfg_in->demod > CSDR_DEMOD_WBFM) { set_error("chain: unknown demod %u", cfg_in->demod); return CSDR_ERR_INVALID; }
    // DeAM = amDemodulator . agc (SoapySDR.hs:265-272): everything up to the per-channel CF32 samples is the DeNo
    // chain; the peak detector and the mix follow as a tail (see csdr_chain_process_device)
    csdr_chain_cfg eff = *cfg_in;
    const bool am = cfg_in->demod == CSDR_DEMOD_AM, am_mix = am && cfg_in->mix != 0 && cfg_in->channels > 1;
    if (am) { eff.demod = CSDR_DEMOD_NONE; eff.mix = 0; }
    // DeWBFM decim = firDecimator decim . iirDeemph . fmDemodulator 0.6 . agc (SoapySDR.hs:252-259, Liquid.chs:653-656)
    const bool wbfm = cfg_in->demod == CSDR_DEMOD_WBFM, wbfm_mix = wbfm && cfg_in->mix != 0 && cfg_in->channels > 1;
    if (wbfm) { eff.demod = CSDR_DEMOD_FM; eff.kf = 0.6f; eff.mix = 0; }
    const csdr_chain_cfg *cfg = &eff;
    if (cfg->channels < 1 || cfg->channels > (1u << 16)) { set_error("chain: channels %u out of range", cfg->channels); return CSDR_ERR_INVALID; }
    if (cfg_in->flags & CSDR_FLAG_TAIL_ONLY) {
        // the per-channel tail alone (hybrid multi-GPU partition): rows in, rows out; state = AGC {g, y2', mode, timer} + freqdem r' per row
        if (cfg_in->agc_threshold_db == 0.0f || am || wbfm) { set_error("chain: CSDR_FLAG_TAIL_ONLY needs the AGC on (-a != 0) and demod none / FM (without the AGC a time stripe needs no tail shard)"); return CSDR_ERR_INVALID; }
        if (cfg->demod == CSDR_DEMOD_FM && !(cfg->kf > 0.f)) { set_error("chain: FM needs kf > 0"); return CSDR_ERR_INVALID; }
        int dev; int r = check_device(cfg->device, &dev); if (r) return r;
        DevGuard guard(dev);
        if (!guard.ok) { set_error("chain: cannot select device %d", dev); return CSDR_ERR_HIP; }
        csdr_chain *h = new (std::nothrow) csdr_chain();
        if (!h) return CSDR_ERR_NOMEM;
        h->cfg = *cfg; h->cfg.chan_first = 0; h->cfg.chan_count = 0; h->cfg.chan_stride = 0; h->cfg.dc_block = 0;
        h->device = dev; h->tail_only = true;
        h->M = cfg->channels; h->C = cfg->channels; h->c0 = 0; h->G = 1; h->p = 0;
        h->max_nf = cfg->max_frames ? cfg->max_frames : 4096;
        h->max_nx = (uint64_t)h->max_nf * h->M;
        if (h->max_nx > 0xffffffffull) { set_error("chain: max_frames*channels exceeds 2^32-1 samples"); delete h; return CSDR_ERR_INVALID; }
        h->agc = make_agc(cfg->agc_threshold_db);
        if (cfg->demod == CSDR_DEMOD_FM) h->fm_ref = fm_ref_of(cfg->kf);
        auto failt = [&](int code) { csdr_chain_destroy(h); return code; };
        if ((r = dev_alloc(&h->d_dcstate, 1))) return failt(r);
        if ((r = dev_alloc(&h->d_agc, h->C))) return failt(r);
        if (cfg->demod == CSDR_DEMOD_FM && ((r = dev_alloc(&h->d_rp[0], h->C)) || (r = dev_alloc(&h->d_rp[1], h->C)))) return failt(r);
        if (cfg->mix && h->C > 1 && (r = dev_alloc(&h->d_B, (size_t)h->C * h->max_nf))) return failt(r);
        if (!(cfg->flags & CSDR_FLAG_AGC_SEQUENTIAL)) { if ((r = agc_tail_create(h->C, h->max_nf, &h->agc_tail))) return failt(r); }
        else if ((r = dev_alloc(&h->d_A, (size_t)h->C * h->max_nf))) return failt(r);        // the one-lane-per-channel kernel works in place
        h->path = std::string("tail-only+agc") + (h->agc_tail ? "-spec" : "");
        h->timed_kernel = "k_agc_spec";
        if ((r = chain_init_state(h, nullptr))) return failt(r);
        CSDR_HIP_CLEAN(hipDeviceSynchronize(), csdr_chain_destroy(h));
        if (!(cfg->flags & CSDR_FLAG_QUIET)) {
            printf("csdr chain [%s] on HIP device %d: %u channel rows, agc=%g dB demod=%s kf=%g mix=%u\n", h->path.c_str(), dev, h->C,
                   cfg->agc_threshold_db, cfg->demod == CSDR_DEMOD_FM ? "FM" : "none", cfg->kf, cfg->mix);
            fflush(stdout);
        }
        *out = h;
        return CSDR_OK;
    }
    if (cfg->demod == CSDR_DEMOD_FM && !(cfg->kf > 0.f)) { set_error("chain: FM needs kf > 0"); return CSDR_ERR_INVALID; }
    if (cfg->dc_block && !(cfg->dc_alpha > 0.f && cfg->dc_alpha < 1.f)) { set_error("chain: dc_alpha out of (0,1)"); return CSDR_ERR_INVALID; }
    const uint32_t M = cfg->channels;
    uint32_t c0 = cfg->chan_first, C = cfg->chan_count ? cfg->chan_count : M - c0;
    const uint32_t G = cfg->chan_stride > 1 ? cfg->chan_stride : 1;
    if (G > 1) {
        if (M % G || c0 >= G || (cfg->chan_count && cfg->chan_count != M / G)) {
            set_error("chain: interleaved shard %u of %u needs chan_stride | channels (%u), chan_first < chan_stride, chan_count 0 or channels/chan_stride", c0, G, M);
            return CSDR_ERR_INVALID;
        }
        C = M / G;
    } else if (c0 >= M || C == 0 || c0 + C > M) { set_error("chain: channel shard [%u,+%u) outside 0..%u", c0, C, M); return CSDR_ERR_INVALID; }
    int dev; int r = check_device(cfg->device, &dev); if (r) return r;
    DevGuard guard(dev);
    if (!guard.ok) { set_error("chain: cannot select device %d", dev); return CSDR_ERR_HIP; }

    csdr_chain *h = new (std::nothrow) csdr_chain();
    if (!h) return CSDR_ERR_NOMEM;
    h->cfg = *cfg; h->device = dev; h->M = M; h->C = C; h->c0 = c0; h->G = G;
    { int cus = 256; (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev); h->n_cus = (uint32_t)cus; }
    const uint32_t m = cfg->pfb_m ? cfg->pfb_m : 7;
    const float As = cfg->pfb_as > 0.f ? cfg->pfb_as : 80.0f;
    h->p = 2 * m;
    h->max_nf = cfg->max_frames ? cfg->max_frames : 4096;
    h->max_nx = (uint64_t)h->max_nf * M;
    if (h->max_nx > 0xffffffffull) { set_error("chain: max_frames*channels exceeds 2^32-1 samples"); delete h; return CSDR_ERR_INVALID; }
    h->timer.enabled = (cfg->flags & (CSDR_FLAG_TIME_KERNELS | CSDR_FLAG_TIME_REGION)) != 0;
    h->timer.region = (cfg->flags & CSDR_FLAG_TIME_REGION) != 0;
    if (cfg->dc_block) h->dc = make_dc(cfg->dc_alpha);
    if (cfg->agc_threshold_db != 0.0f) h->agc = make_agc(cfg->agc_threshold_db);
    if (cfg->demod == CSDR_DEMOD_FM) h->fm_ref = fm_ref_of(cfg->kf);

    auto fail = [&](int code) { csdr_chain_destroy(h); return code; };
    if ((r = dev_alloc(&h->d_dcstate, 1))) return fail(r);
    if ((r = dev_alloc(&h->d_scratch, 2 * (size_t)(h->max_nx / DC_BLOCK + 2)))) return fail(r);

    if (M > 1) {
        h->taps = design_pfb_taps(M, m, As);
        h->d_theta = nco_freq_word(pfb_premix_freq(M));
        h->tab_len = nco_period(h->d_theta, 1u << 17);
        if ((r = dev_alloc(&h->d_taps, h->taps.size()))) return fail(r);
        CSDR_HIP_CLEAN(hipMemcpy(h->d_taps, h->taps.data(), sizeof(float) * h->taps.size(), hipMemcpyHostToDevice), csdr_chain_destroy(h));
        if (h->tab_len) {
            std::vector<float2> tab(h->tab_len);
            for (uint32_t i = 0; i < h->tab_len; i++) { float c, s; nco_phasor(i * h->d_theta, &c, &s); tab[i] = make_float2(c, s); }
            if ((r = dev_alloc(&h->d_nco_tab, h->tab_len))) return fail(r);
            CSDR_HIP_CLEAN(hipMemcpy(h->d_nco_tab, tab.data(), sizeof(float2) * h->tab_len, hipMemcpyHostToDevice), csdr_chain_destroy(h));
        }
        std::vector<float2> tw(M);
        for (uint32_t i = 0; i < M; i++) {
            double a = -2.0 * 3.14159265358979323846 * (double)i / (double)M;
            tw[i] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
        if ((r = dev_alloc(&h->d_tw, M))) return fail(r);
        CSDR_HIP_CLEAN(hipMemcpy(h->d_tw, tw.data(), sizeof(float2) * M, hipMemcpyHostToDevice), csdr_chain_destroy(h));
    }
    if (cfg->agc_threshold_db != 0.0f && (r = dev_alloc(&h->d_agc, C))) return fail(r);
    if (cfg->demod == CSDR_DEMOD_FM && ((r = dev_alloc(&h->d_rp[0], C)) || (r = dev_alloc(&h->d_rp[1], C)))) return fail(r);

    // Path selection: the fused kernels cover M = 256 (DC blocker + pre-mix + PFB [+ freqdem]).
    // With the AGC on, the fused kernel stops at the channel-major CF32 samples and the
    // exactly-sequential per-channel AGC tail (one lane per channel) + freqdem + mix follow.
    // interleaved shards: the fused M = 256 chain takes strides 2, 4, 8 (k_run256v2<.., G>); every other shape the any-M route with a pruned DFT
    const RouteRow *route = route_select(M, h->p, G, cfg->flags);          // the one place a configuration is mapped to a plan (table above)
    if (route->plan == PLAN_HUGE4096 && (C != M || (cfg->mix && cfg->demod == CSDR_DEMOD_NONE && cfg->agc_threshold_db == 0.0f && !am)))
        route = route_select(M, h->p, G, cfg->flags | CSDR_FLAG_FORCE_GENERIC);   // contiguous shards; DeNo --mix over all channels (= M x branch 0: nothing beats not computing
                                                                                   // the bank; CSDR_FLAG_NO_MIX_IDENTITY's full bank + DFT + sum stays on the any-M kernels too)
    h->use_fused = route->plan != PLAN_GENERIC;
    if (h->use_fused) {
        const bool agc_on = cfg->agc_threshold_db != 0.0f;
        FusedConfig fc{};
        fc.M = M; fc.p = h->p; fc.C = C; fc.c0 = c0; fc.max_nf = h->max_nf; fc.G = G;
        fc.dc_block = cfg->dc_block != 0; fc.dc = h->dc;
        fc.fm = cfg->demod == CSDR_DEMOD_FM && !agc_on; fc.fm_ref = h->fm_ref;
        fc.mix = cfg->mix != 0 && !agc_on; fc.taps = h->taps.data(); fc.d_theta = h->d_theta;
        if (route->plan == PLAN_SMALL64) {
            if ((r = small_create(fc, &h->small))) return fail(r);
            h->path = std::string("fused-") + small_name(h->small) + (agc_on ? "+agc" : "");
            h->timed_kernel = small_name(h->small);
        } else if (route->plan == PLAN_HUGE4096) {
            if ((r = huge_create(fc, &h->huge))) return fail(r);
            h->path = std::string("fused-4096|") + huge_name(h->huge) + (agc_on ? "+agc" : "");
            h->timed_kernel = huge_name(h->huge);
        } else if (route->plan == PLAN_BIG1024) {
            if ((r = big_create(fc, &h->big))) return fail(r);
            h->path = std::string("fused-") + big_name(h->big) + (G > 1 ? "+interleaved-shard" : "") + (agc_on ? "+agc" : "");
            h->timed_kernel = big_name(h->big);
        } else {
            if ((r = fused_create(fc, &h->fused))) return fail(r);
            h->path = std::string("fused-256|") + fused_name(h->fused) + (G > 1 ? "+interleaved-shard" : "") + (agc_on ? "+agc" : "");
            h->timed_kernel = fused_name(h->fused);
        }
        if (agc_on) {
            // d_A: channel-major CF32 from the channelizer; d_B: per-channel tail output in front of --mix
            // (the fused M = 256 plans may write it tile-major for k_agc_spec_tm, which reads up to a segment in front of / behind the plane)
            // The guards (2 x 64 KiB per channel: 128 MiB at 1024 channels) exist only where a call can take that route: a time-parallel
            // tail (not CSDR_FLAG_AGC_SEQUENTIAL), a channel count k_agc_spec_tm takes, calls of >= 4 W = 4096 frames
            const bool tm_possible = (h->fused || h->big || h->small || h->huge) && !(cfg->flags & CSDR_FLAG_AGC_SEQUENTIAL) && h->max_nf >= 4096u &&
                                     (C % 64u == 0 || (C < 64u && 64u % C == 0));
            h->a_guard = tm_possible ? agc_tail_tm_guard(C) : 0;
            if ((r = dev_alloc(&h->d_A, (size_t)C * h->max_nf + 2 * h->a_guard))) return fail(r);
            if (h->a_guard) {
                // only the guards need defined contents (the warm-ups of the first and last segments read them, masked by position)
                CSDR_HIP_CLEAN(hipMemset(h->d_A, 0, sizeof(float2) * h->a_guard), csdr_chain_destroy(h));
                CSDR_HIP_CLEAN(hipMemset(h->d_A + h->a_guard + (size_t)C * h->max_nf, 0, sizeof(float2) * h->a_guard), csdr_chain_destroy(h));
            }
            if (cfg->mix && (r = dev_alloc(&h->d_B, (size_t)C * h->max_nf))) return fail(r);
        }
    } else {
        h->path = "generic";
        const bool use1024 = G == 1 && pfb1024_supported(M, h->p) && !(cfg->mix && cfg->agc_threshold_db == 0.0f) && !diag_env("CSDR_NO_PFB1024");
        h->timed_kernel = M > 1 ? (use1024 ? "k_pfb1024" : "k_pfb_fir") : "k_dc_apply";
        if (use1024) h->path = "generic+pfb1024";
        if (M > 1) {
            if ((r = dev_alloc(&h->d_u, (size_t)(h->p - 1) * M + h->max_nx)) || (r = dev_alloc(&h->d_hist_tmp, (size_t)(h->p - 1) * M))) return fail(r);
        }
        if ((r = dev_alloc(&h->d_A, h->max_nx)) || (r = dev_alloc(&h->d_B, use1024 && h->max_nx < 2048 ? 2048 : h->max_nx))) return fail(r);   // k_pfb1024 keeps yfirst|ylast (2 x nruns x 1024) in d_B
        if (M > 1 && cfg->dc_block && (r = dctile_create(h->dc, h->max_nx, &h->dctile))) return fail(r);
        h->mix_identity = M > 1 && G == 1 && C == M && cfg->mix && cfg->demod == CSDR_DEMOD_NONE && cfg->agc_threshold_db == 0.0f &&
                          h->dctile && !(cfg->flags & CSDR_FLAG_NO_MIX_IDENTITY) && !am;
        if (h->mix_identity) {
            if ((r = dev_alloc(&h->d_u0, (size_t)(h->p - 1) + h->max_nf)) || (r = dev_alloc(&h->d_u0hist, 2 * (h->p - 1)))) return fail(r);
            h->path = "generic+mix-identity"; h->timed_kernel = (M % 4096u == 0) ? "k_dc_fold" : "k_dc_tile";   // refined per call
        }
        // interleaved shard, DeNo --mix, no AGC: only the G branches (M / G) n2 survive the shard's channel sum (kernels_dc_tile.hip, k_dc_fold8)
        h->mix_identity_shard = M > 1 && (G == 2 || G == 4 || G == 8) && (uint64_t)C * G == M && cfg->mix && cfg->demod == CSDR_DEMOD_NONE && cfg->agc_threshold_db == 0.0f &&
                                h->dctile && !(cfg->flags & CSDR_FLAG_NO_MIX_IDENTITY) && !am && M % 4096u == 0 && (M / G) % 512u == 0;
        if (h->mix_identity_shard) {
            if ((r = dev_alloc(&h->d_u0hist, 2 * (size_t)(h->p - 1) * G))) return fail(r);
        }
        if (G > 1) {
            const uint32_t Mg = M / G;
            std::vector<float2> twg(Mg), ph(G + Mg);
            const double tp = -2.0 * 3.14159265358979323846;
            for (uint32_t i = 0; i < Mg; i++) twg[i] = make_float2((float)std::cos(tp * i / Mg), (float)std::sin(tp * i / Mg));
            for (uint32_t j2 = 0; j2 < G; j2++) ph[j2] = make_float2((float)std::cos(tp * ((uint64_t)j2 * c0 % G) / G), (float)std::sin(tp * ((uint64_t)j2 * c0 % G) / G));
            for (uint32_t j1 = 0; j1 < Mg; j1++) ph[G + j1] = make_float2((float)std::cos(tp * ((uint64_t)j1 * c0 % M) / M), (float)std::sin(tp * ((uint64_t)j1 * c0 % M) / M));
            if ((r = dev_alloc(&h->d_tw_g, Mg)) || (r = dev_alloc(&h->d_fold_ph, G + Mg)) || (r = dev_alloc(&h->d_fold, (size_t)Mg * h->max_nf))) return fail(r);
            if (hipMemcpy(h->d_tw_g, twg.data(), sizeof(float2) * Mg, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(h->d_fold_ph, ph.data(), sizeof(float2) * (G + Mg), hipMemcpyHostToDevice) != hipSuccess) { set_error("chain: fold table upload failed"); return fail(CSDR_ERR_HIP); }
            h->path = h->mix_identity_shard ? "generic+pruned-dft+shard-mix-identity" : "generic+pruned-dft";
            if (h->mix_identity_shard) h->timed_kernel = "k_dc_fold8";
        }
    }
    if ((cfg_in->flags & CSDR_FLAG_DFT_BACKWARD) && M > 1 && !(cfg_in->mix != 0)) {
        if (C != M || G > 1) { set_error("chain: CSDR_FLAG_DFT_BACKWARD is built for whole-band handles (no channel shard)"); return fail(CSDR_ERR_INVALID); }
        h->dft_backward = true;
        CSDR_HIP_CLEAN(hipMalloc(&h->d_perm, (size_t)C * h->max_nf * ((cfg_in->demod == CSDR_DEMOD_NONE) ? 8u : 4u)), csdr_chain_destroy(h));
        h->path += "+dft-backward";
    }
    if (am) {
        h->am = true; h->am_mix = am_mix;
        if ((r = dev_alloc(&h->d_amz, (size_t)C * h->max_nf))) return fail(r);
        if (am_mix) { CSDR_HIP_CLEAN(hipMalloc(&h->d_amf, sizeof(float) * (size_t)C * h->max_nf), csdr_chain_destroy(h)); }
        CSDR_HIP_CLEAN(hipMalloc(&h->d_amq[0], sizeof(float) * C), csdr_chain_destroy(h)); CSDR_HIP_CLEAN(hipMalloc(&h->d_amq[1], sizeof(float) * C), csdr_chain_destroy(h));
        h->path += "+am";
    }
    if (wbfm) {
        h->wbfm = true; h->wbfm_mix = wbfm_mix;
        h->wb_decim = cfg_in->wbfm_decim ? cfg_in->wbfm_decim : 4u;
        const float fc = cfg_in->deemph_fc > 0.f ? cfg_in->deemph_fc : 0.025f;
        if (!(fc < 0.5f)) { set_error("chain: deemph_fc %g out of (0, 0.5)", fc); return fail(CSDR_ERR_INVALID); }
        h->wb_bq = design_butter2_lowpass(fc);
        const std::vector<float> taps = design_firdecim_kaiser(h->wb_decim, 10, 60.0f);
        h->wb_hlen = (uint32_t)taps.size();
        const size_t rows = (size_t)C * h->max_nf, hist = (size_t)C * (h->wb_hlen - 1);
        if (hipMalloc(&h->d_wbf, sizeof(float) * rows) != hipSuccess || hipMalloc(&h->d_wbh, sizeof(float) * taps.size()) != hipSuccess ||
            hipMalloc(&h->d_wbhist[0], sizeof(float) * hist) != hipSuccess || hipMalloc(&h->d_wbhist[1], sizeof(float) * hist) != hipSuccess ||
            hipMalloc(&h->d_wbst[0], sizeof(float2) * C) != hipSuccess || hipMalloc(&h->d_wbst[1], sizeof(float2) * C) != hipSuccess ||
            (wbfm_mix && hipMalloc(&h->d_wbo, sizeof(float) * rows) != hipSuccess)) { set_error("chain: WBFM tail allocation failed"); return fail(CSDR_ERR_HIP); }
        CSDR_HIP_CLEAN(hipMemcpy(h->d_wbh, taps.data(), sizeof(float) * taps.size(), hipMemcpyHostToDevice), csdr_chain_destroy(h));
        h->path += "+wbfm";
    }
    if (h->d_agc && !(cfg->flags & CSDR_FLAG_AGC_SEQUENTIAL)) {
        if ((r = agc_tail_create(C, h->max_nf, &h->agc_tail))) return fail(r);
        h->path += h->use_fused ? "-spec" : "+agc-spec";
    }
    if ((r = chain_init_state(h, nullptr))) return fail(r);
    CSDR_HIP_CLEAN(hipDeviceSynchronize(), csdr_chain_destroy(h));

    if (!(cfg->flags & CSDR_FLAG_QUIET)) {
        // what the reference prints at create (Liquid.chs:577-579, 814-820, 714-716, 319-321)
        printf("csdr chain [%s] on HIP device %d: channels=%u (shard %u..%u) taps=%u (m=%u, As=%.1f) "
               "nco.d_theta=0x%08x dc_block=%u(alpha=%g) agc=%g dB demod=%s kf=%g mix=%u\n",
               h->path.c_str(), dev, M, c0, c0 + C - 1, M > 1 ? M * h->p : 0, m, As, h->d_theta, cfg->dc_block,
               cfg->dc_alpha, cfg->agc_threshold_db, am ? "AM" : (wbfm ? "WBFM" : (cfg->demod == CSDR_DEMOD_FM ? "FM" : "none")), cfg->kf, cfg_in->mix);
        // per-channel outputs / AGC / demod tails beyond the fused kernels' channel counts run stage by stage through HBM
        if (!h->use_fused && !h->mix_identity && M > 1024)
            printf("csdr chain: note: %u channels with per-channel output run on the any-M route (DC blocker, FIR, DFT and tail as separate kernels, "
                   "~0.09 of the HBM roofline on MI355X); fused kernels exist for 64, 256 and 1024 channels, and for DeNo --mix over all channels of any count\n", M);
        fflush(stdout);
    }
    *out = h;
    return CSDR_OK;
}

uint32_t csdr_chain_out_elem_size(const csdr_chain *h) { return h && (h->cfg.demod == CSDR_DEMOD_FM || h->am || h->wbfm) ? 4u : 8u; }

// AGC on: Z[C][nf] (channel-major CF32 in d_A) -> AGC + squelch [+ freqdem] [+ mix] -> d_out
static int chain_agc_tail(csdr_chain *h, const float2 *Z, uint32_t nf, void *d_out, hipStream_t s, bool tm = false)
{
    const bool fm = h->cfg.demod == CSDR_DEMOD_FM, mixo = h->cfg.mix && h->M > 1;
    void *T = mixo ? (void *)h->d_B : d_out;
    int r = agc_tail_process(h->agc_tail, Z, T, fm, nf, h->d_agc, h->agc, h->fm_ref,
                             fm ? h->d_rp[h->rp_cur] : nullptr, fm ? h->d_rp[h->rp_cur ^ 1] : nullptr, s, tm);
    if (r) return r;
    if (fm) h->rp_cur ^= 1;
    if (mixo) return launch_mix((const float *)T, (float *)d_out, h->C, fm ? nf : 2 * nf, s);
    return 0;
}

static int chain_generic(csdr_chain *h, const float2 *d_in, uint32_t nx, void *d_out, hipStream_t s)
{
    const uint32_t M = h->M, nf = nx / M, C = h->C;
    const bool fm = h->cfg.demod == CSDR_DEMOD_FM, mixo = h->cfg.mix && M > 1, agc = h->d_agc != nullptr;
    int r;
    // where the channel-major CF32 lands
    float2 *Z = (!fm && !mixo && !(agc && h->agc_tail)) ? (float2 *)d_out : h->d_A;
    NcoParams nco{};
    if (M > 1) {
        const size_t hist = (size_t)(h->p - 1) * M;
        nco.theta0 = h->theta; nco.d_theta = h->d_theta; nco.tab_len = h->tab_len; nco.tab_pos = h->tab_pos; nco.up = 0;
        if (h->mix_identity_shard && dctile_mix_identity_shard_supported(h->dctile, M, nx, h->p, h->G)) {
            // the shard's channel sum = (M / G) x sum of the G surviving branches' FIRs (whole frames of M % 4096 == 0 samples; other calls
            // fall through to the pruned-DFT route below, which shares no state with this one but the DC blocker's: see the create-time note)
            const size_t hs = (size_t)(h->p - 1) * h->G;
            float2 *hin = h->d_u0hist + (size_t)h->u0_cur * hs, *hout = h->d_u0hist + (size_t)(h->u0_cur ^ 1) * hs;
            h->timed_kernel = "k_dc_fold8";
            if ((r = h->timer.begin(s))) return r;
            if ((r = dctile_mix_identity_shard(h->dctile, d_in, nx, nco, h->d_nco_tab, h->d_taps, M, h->p, h->G, h->c0, hin, hout, (float2 *)d_out, s))) return r;
            if ((r = h->timer.end(s))) return r;
            h->u0_cur ^= 1;
            h->theta += nx * h->d_theta;
            if (h->tab_len) h->tab_pos = (uint32_t)(((uint64_t)h->tab_pos + nx) % h->tab_len);
            return 0;
        }
        if (h->mix_identity) {
            // sum over ALL channels of a frame = M * X_t[0]: DC blocker + pre-mix on the whole stream, every M-th sample kept,
            // then the 2m-tap FIR of polyphase branch 0 (no bank, no DFT, no channel sum; 8 B read per input sample)
            float2 *hin = h->d_u0hist + (size_t)h->u0_cur * (h->p - 1), *hout = h->d_u0hist + (size_t)(h->u0_cur ^ 1) * (h->p - 1);
            if (dctile_mix_identity_supported(h->dctile, M, nx, h->p)) {
                // k_dc_fold (one aggregate per tile: a plain streaming read) + k_mixid_finish (DC state, pick, pre-mix, FIR)
                h->timed_kernel = "k_dc_fold";
                if ((r = h->timer.begin(s))) return r;
                if ((r = dctile_mix_identity(h->dctile, d_in, nx, nco, h->d_nco_tab, h->d_taps, M, h->p, hin, hout, (float2 *)d_out, s))) return r;
                if ((r = h->timer.end(s))) return r;
            } else {
                h->timed_kernel = "k_dc_tile";
                CSDR_HIP(hipMemcpyAsync(h->d_u0, hin, sizeof(float2) * (h->p - 1), hipMemcpyDeviceToDevice, s));
                if ((r = h->timer.begin(s))) return r;
                if ((r = dctile_process(h->dctile, d_in, h->d_u0 + (h->p - 1), nx, true, nco, h->d_nco_tab, s, M))) return r;
                if ((r = h->timer.end(s))) return r;
                if ((r = launch_branch0_fir(h->d_u0, h->d_taps, (float2 *)d_out, hout, M, h->p, nf, s))) return r;
            }
            h->u0_cur ^= 1;
            h->theta += nx * h->d_theta;
            if (h->tab_len) h->tab_pos = (uint32_t)(((uint64_t)h->tab_pos + nx) % h->tab_len);
            return 0;
        }
        float2 *u_new = h->d_u + hist;
        if (h->dctile) r = dctile_process(h->dctile, d_in, u_new, nx, true, nco, h->d_nco_tab, s);
        else r = launch_dc_mix(d_in, u_new, nx, h->cfg.dc_block != 0, h->dc, h->d_dcstate, h->d_scratch, true, nco, h->d_nco_tab, s);
        if (r) return r;
        // M = 1024: FIR + DFT + transpose [+ freqdem] in one kernel (no X / Y round trips through HBM); the frame-major
        // mix tails without AGC still want Y in HBM and keep the three-kernel route
        const bool fused1024 = h->G == 1 && pfb1024_supported(M, h->p) && !(mixo && !agc) && !diag_env("CSDR_NO_PFB1024");
        if ((r = h->timer.begin(s))) return r;
        if (fused1024) {
            const bool fm_here = fm && !agc;                     // with the AGC on the tail does the freqdem
            void *o = fm_here ? d_out : (void *)Z;
            r = launch_pfb1024(u_new, h->d_taps, h->d_tw, o, fm_here, nf, h->c0, C, h->fm_ref, fm_here ? h->d_rp[h->rp_cur] : nullptr,
                               fm_here ? h->d_rp[h->rp_cur ^ 1] : nullptr, h->d_B, h->n_cus, s);
            if (!r && fm_here) h->rp_cur ^= 1;
        } else r = launch_pfb_fir(u_new, h->d_taps, h->d_A, M, h->p, nf, s);
        if (r) return r;
        if ((r = h->timer.end(s))) return r;
        // keep the last (p-1) frames of premixed input as the next call's history
        CSDR_HIP(hipMemcpyAsync(h->d_hist_tmp, h->d_u + nx, sizeof(float2) * hist, hipMemcpyDeviceToDevice, s));
        CSDR_HIP(hipMemcpyAsync(h->d_u, h->d_hist_tmp, sizeof(float2) * hist, hipMemcpyDeviceToDevice, s));
        if (fused1024) {
            h->theta += nx * h->d_theta;
            if (h->tab_len) h->tab_pos = (uint32_t)(((uint64_t)h->tab_pos + nx) % h->tab_len);
            if (fm && !agc) return 0;
            if (agc && h->agc_tail) return chain_agc_tail(h, Z, nf, d_out, s);
            if (agc) {
                if ((r = launch_agc(Z, C, nf, h->d_agc, h->agc, s))) return r;
                if (fm) {
                    float *F = mixo ? (float *)h->d_B : (float *)d_out;
                    if ((r = launch_fm(Z, F, C, nf, h->fm_ref, h->d_rp[h->rp_cur], h->d_rp[h->rp_cur ^ 1], s))) return r;
                    h->rp_cur ^= 1;
                    if (mixo && (r = launch_mix(F, (float *)d_out, C, nf, s))) return r;
                } else if (mixo && (r = launch_mix((const float *)Z, (float *)d_out, C, 2 * nf, s))) return r;
            }
            return 0;
        }
        // DeNo --mix over all channels: the frame sum happens inside the DFT kernel, Y never goes to HBM
        const bool fused_mix = !agc && !fm && mixo && C == M && dft_mix_supported(M);
        if (h->G > 1) {
            // interleaved shard: fold the G sub-blocks of every frame, then an (M/G)-point DFT: d_B = Y[t][c0 + G m]
            if ((r = launch_fold(h->d_A, h->d_fold, h->d_fold_ph, M, h->G, nf, s))) return r;
            r = launch_dft(h->d_fold, h->d_B, h->d_tw_g, M / h->G, nf, s);
        } else if (fused_mix) r = launch_dft_mix(h->d_A, (float2 *)d_out, h->d_tw, M, nf, s);
        else r = launch_dft(h->d_A, h->d_B, h->d_tw, M, nf, s);
        if (r) return r;
        h->theta += nx * h->d_theta;
        if (h->tab_len) h->tab_pos = (uint32_t)(((uint64_t)h->tab_pos + nx) % h->tab_len);
        if (fused_mix) return 0;
        const uint32_t Mw = M / h->G, cw = h->G > 1 ? 0u : h->c0;     // width of a DFT output frame, first owned bin in it
        if (!agc && (fm || mixo)) {
            // frame-major tails: no transpose in front of freqdem / mix
            if (mixo) r = launch_mix_frames(h->d_B, d_out, fm, Mw, nf, cw, C, h->fm_ref, h->d_rp[h->rp_cur], h->d_rp[h->rp_cur ^ 1], s);
            else r = launch_transpose_fm(h->d_B, (float *)d_out, Mw, nf, cw, C, h->fm_ref, h->d_rp[h->rp_cur], h->d_rp[h->rp_cur ^ 1], s);
            if (r) return r;
            if (fm) h->rp_cur ^= 1;
            return 0;
        }
        if ((r = launch_transpose(h->d_B, Z, Mw, nf, cw, C, s))) return r;
    } else {
        if ((r = h->timer.begin(s))) return r;
        if ((r = launch_dc_mix(d_in, Z, nx, h->cfg.dc_block != 0, h->dc, h->d_dcstate, h->d_scratch, false, nco, nullptr, s))) return r;
        if ((r = h->timer.end(s))) return r;
    }
    if (agc && h->agc_tail) return chain_agc_tail(h, Z, nf, d_out, s);
    if (agc && (r = launch_agc(Z, C, nf, h->d_agc, h->agc, s))) return r;
    if (fm) {
        float *F = mixo ? (float *)h->d_B : (float *)d_out;
        if ((r = launch_fm(Z, F, C, nf, h->fm_ref, h->d_rp[h->rp_cur], h->d_rp[h->rp_cur ^ 1], s))) return r;
        h->rp_cur ^= 1;
        if (mixo && (r = launch_mix(F, (float *)d_out, C, nf, s))) return r;
    } else if (mixo) {
        if ((r = launch_mix((const float *)Z, (float *)d_out, C, 2 * nf, s))) return r;
    }
    return 0;
}

static int chain_process_device_inner(csdr_chain *h, const void *d_in, uint32_t n_in, void *d_out, uint32_t *n_out, void *stream);

static int chain_process_device_any(csdr_chain *h, const void *d_in, uint32_t n_in, void *d_out, uint32_t *n_out, void *stream);

// A call on a caller's stream is remembered by an event behind its work, so that the host-buffer entry point (own stream s_k) can
// order itself behind it without keeping the caller's stream handle
static int chain_mark_user_stream(csdr_chain *h, hipStream_t stream)
{
    // only a handle that has used its host-buffer entry points has anything to order against (no event record per call on
    // the device-only hot path, and none inside a caller's stream capture); the first host-buffer call synchronises the device once
    if (!h->s_k) return CSDR_OK;
    if (!h->e_user) CSDR_HIP(hipEventCreateWithFlags(&h->e_user, hipEventDisableTiming));
    CSDR_HIP(hipEventRecord(h->e_user, stream));
    h->user_stream_dirty = true;
    return CSDR_OK;
}

int csdr_chain_process_device(csdr_chain *h, const void *d_in, uint32_t n_in, void *d_out, uint32_t *n_out, void *stream)
{
    if (!h || h->in_submit) return chain_process_device_any(h, d_in, n_in, d_out, n_out, stream);
    const bool own = h->s_k && (hipStream_t)stream == h->s_k;       // called by csdr_chain_submit / csdr_chain_process on the handle's stream
    DevGuard guard(h->device);
    if (!own && h->s_k && h->q_count) CSDR_HIP(hipStreamSynchronize(h->s_k));   // host-buffer chunks still in flight come first
    if (h->s_pd[0]) {
        // the handle has pipelined chunks (csdr_chain_submit_device): this call follows them, and a later submit follows this call
        for (int i = 0; i < 2; i++) if (h->pd_used[i]) CSDR_HIP(hipStreamWaitEvent((hipStream_t)stream, h->e_pd_done[i], 0));
    }
    int r = chain_process_device_any(h, d_in, n_in, d_out, n_out, stream);
    if (r) return r;
    if (h->s_pd[0]) {
        CSDR_HIP(hipEventRecord(h->e_serial, (hipStream_t)stream));
        h->serial_pending = true;
    }
    return own ? CSDR_OK : chain_mark_user_stream(h, (hipStream_t)stream);
}

static int chain_process_device_any0(csdr_chain *h, const void *d_in, uint32_t n_in, void *d_out, uint32_t *n_out, void *stream);
static int chain_process_device_any(csdr_chain *h, const void *d_in, uint32_t n_in, void *d_out, uint32_t *n_out, void *stream)
{
    if (!h || !h->dft_backward || n_in == 0 || !d_out) return chain_process_device_any0(h, d_in, n_in, d_out, n_out, stream);
    // CSDR_FLAG_DFT_BACKWARD: the chain as built (forward DFT) into d_perm, then row k of the output = its row (M - k) mod M
    uint32_t n = 0;
    int r = chain_process_device_any0(h, d_in, n_in, h->d_perm, &n, stream);
    if (r) return r;
    if (n_out) *n_out = n;
    DevGuard guard(h->device);
    return launch_rows_reversed(h->d_perm, d_out, h->C, (size_t)(n / h->C) * csdr_chain_out_elem_size(h), (hipStream_t)stream);
}

static int chain_process_device_any0(csdr_chain *h, const void *d_in, uint32_t n_in, void *d_out, uint32_t *n_out, void *stream)
{
    if (h && h->wbfm) {
        if (n_out) *n_out = 0;
        if (n_in == 0) return CSDR_OK;
        if (!d_out) { set_error("chain: null buffer"); return CSDR_ERR_INVALID; }
        if (n_in % h->M == 0 && (n_in / h->M) % h->wb_decim) {
            set_error("chain: %u frames per call are not a multiple of the WBFM decimation %u (firDecim's `div`, Liquid.chs:495-497)", n_in / h->M, h->wb_decim);
            return CSDR_ERR_SIZE;
        }
        int r = chain_process_device_inner(h, d_in, n_in, h->d_wbf, nullptr, stream);
        if (r) return r;
        DevGuard guard(h->device);
        hipStream_t s = (hipStream_t)stream;
        const uint32_t nf = n_in / h->M, no = nf / h->wb_decim;
        if ((r = launch_biquad(h->d_wbf, h->d_wbf, h->C, nf, h->wb_bq, h->d_wbst[h->wb_cur], h->d_wbst[h->wb_cur ^ 1], s))) return r;
        float *O = h->wbfm_mix ? h->d_wbo : (float *)d_out;
        if ((r = launch_firdecim(h->d_wbf, O, h->C, nf, h->wb_decim, h->d_wbh, h->wb_hlen, h->d_wbhist[h->wb_cur], h->d_wbhist[h->wb_cur ^ 1], s))) return r;
        h->wb_cur ^= 1;
        if (h->wbfm_mix && (r = launch_mix(O, (float *)d_out, h->C, no, s))) return r;
        if (n_out) *n_out = h->wbfm_mix ? no : h->C * no;
        return CSDR_OK;
    }
    if (!h || !h->am) return chain_process_device_inner(h, d_in, n_in, d_out, n_out, stream);
    if (n_out) *n_out = 0;
    if (n_in == 0) return CSDR_OK;
    if (!d_out) { set_error("chain: null buffer"); return CSDR_ERR_INVALID; }
    int r = chain_process_device_inner(h, d_in, n_in, h->d_amz, nullptr, stream);
    if (r) return r;
    DevGuard guard(h->device);
    hipStream_t s = (hipStream_t)stream;
    const uint32_t nf = n_in / h->M;
    float *F = h->am_mix ? h->d_amf : (float *)d_out;
    if ((r = launch_am(h->d_amz, F, h->C, nf, h->d_amq[h->amq_cur], h->d_amq[h->amq_cur ^ 1], 0.01f, s))) return r;
    h->amq_cur ^= 1;
    if (h->am_mix && (r = launch_mix(F, (float *)d_out, h->C, nf, s))) return r;
    if (n_out) *n_out = h->am_mix ? nf : h->C * nf;
    return CSDR_OK;
}

static int chain_process_device_inner(csdr_chain *h, const void *d_in, uint32_t n_in, void *d_out, uint32_t *n_out, void *stream)
{
    if (!h) { set_error("chain: null handle"); return CSDR_ERR_INVALID; }
    if (n_out) *n_out = 0;
    if (n_in == 0) return CSDR_OK;
    if (!d_in || !d_out) { set_error("chain: null buffer"); return CSDR_ERR_INVALID; }
    if (n_in % h->M) { set_error("chain: n_in=%u is not a multiple of channels=%u (the reference misbehaves here; Liquid.chs:832-862)", n_in, h->M); return CSDR_ERR_SIZE; }
    if (n_in > h->max_nx) { set_error("chain: n_in=%u exceeds max_frames*channels=%llu", n_in, (unsigned long long)h->max_nx); return CSDR_ERR_SIZE; }
    DevGuard guard(h->device);
    if (!guard.ok) { set_error("chain: cannot select device %d", h->device); return CSDR_ERR_HIP; }
    hipStream_t s = (hipStream_t)stream;
    const uint32_t nf = n_in / h->M;
    int r;
    if (h->tail_only) {
        const bool fm = h->cfg.demod == CSDR_DEMOD_FM, mixo = h->cfg.mix && h->M > 1;
        if (h->agc_tail) {
            if ((r = chain_agc_tail(h, (const float2 *)d_in, nf, d_out, s))) return r;
        } else {
            CSDR_HIP(hipMemcpyAsync(h->d_A, d_in, sizeof(float2) * (size_t)n_in, hipMemcpyDeviceToDevice, s));
            if ((r = launch_agc(h->d_A, h->C, nf, h->d_agc, h->agc, s))) return r;
            if (fm) {
                float *F = mixo ? (float *)h->d_B : (float *)d_out;
                if ((r = launch_fm(h->d_A, F, h->C, nf, h->fm_ref, h->d_rp[h->rp_cur], h->d_rp[h->rp_cur ^ 1], s))) return r;
                h->rp_cur ^= 1;
                if (mixo && (r = launch_mix(F, (float *)d_out, h->C, nf, s))) return r;
            } else if (mixo) { if ((r = launch_mix((const float *)h->d_A, (float *)d_out, h->C, 2 * nf, s))) return r; }
            else CSDR_HIP(hipMemcpyAsync(d_out, h->d_A, sizeof(float2) * (size_t)n_in, hipMemcpyDeviceToDevice, s));
        }
        if (n_out) *n_out = mixo ? nf : h->C * nf;
        return CSDR_OK;
    }
    if (h->use_fused) {
        const bool agc_on = h->d_agc != nullptr, fm = h->cfg.demod == CSDR_DEMOD_FM, mixo = h->cfg.mix != 0;
        float2 *Z = (agc_on && (fm || mixo || h->agc_tail)) ? h->d_A + h->a_guard : (float2 *)d_out;
        FusedCall fcall{};
        // AGC tail behind the fused M = 256 / M = 1024 chains, run-sized calls of whole tiles: the plane between the two kernels is tile-major
        const bool tm = agc_on && h->agc_tail && h->a_guard && Z != (float2 *)d_out && agc_tail_tm_supported(h->agc_tail, nf) &&
                        (h->fused ? fused_tile_major_ok(h->fused, nf) : h->big ? big_tile_major_ok(h->big, nf) : h->small ? small_tile_major_ok(h->small, nf) : (h->huge && huge_tile_major_ok(h->huge, nf)));
        fcall.tile_major = tm;
        fcall.d_in = (const float2 *)d_in; fcall.d_out = agc_on ? (void *)Z : d_out; fcall.nf = nf; fcall.theta0 = h->theta;
        fcall.indep = h->call_indep; fcall.ev_tail = h->call_ev_tail;
        if (h->small) { if ((r = small_process(h->small, fcall, s, &h->timer))) return r; h->timed_kernel = small_name(h->small); }   // k_run64v2 or k_run64, by call
        else if (h->big) { if ((r = big_process(h->big, fcall, s, &h->timer))) return r; h->timed_kernel = big_name(h->big); }   // k_run1024v2 or k_run1024, by call
        else if (h->huge) { if ((r = huge_process(h->huge, fcall, s, &h->timer))) return r; }
        else if ((r = fused_process(h->fused, fcall, s, &h->timer))) return r;
        h->theta += n_in * h->d_theta;
        if (agc_on && h->agc_tail) {
            if ((r = chain_agc_tail(h, Z, nf, d_out, s, tm))) return r;
        } else if (agc_on) {
            if ((r = launch_agc(Z, h->C, nf, h->d_agc, h->agc, s))) return r;
            if (fm) {
                float *F = mixo ? (float *)h->d_B : (float *)d_out;
                if ((r = launch_fm(Z, F, h->C, nf, h->fm_ref, h->d_rp[h->rp_cur], h->d_rp[h->rp_cur ^ 1], s))) return r;
                h->rp_cur ^= 1;
                if (mixo && (r = launch_mix(F, (float *)d_out, h->C, nf, s))) return r;
            } else if (mixo) {
                if ((r = launch_mix((const float *)Z, (float *)d_out, h->C, 2 * nf, s))) return r;
            }
        }
    } else {
        if ((r = chain_generic(h, (const float2 *)d_in, n_in, d_out, s))) return r;
    }
    if (n_out) *n_out = (h->cfg.mix && h->M > 1) ? nf : h->C * nf;
    return CSDR_OK;
}

// Device-side spin-wait time-outs (k_tile256 look-back / hand-off, k_dc_tile look-back) set a sticky status word and the
// kernel carries on with garbage carries: report it as an error once, then clear it.  Synchronises the device.
static int chain_device_status(csdr_chain *h)
{
    unsigned st = 0, st2 = 0;
    int r;
    if (h->fused && (r = fused_status(h->fused, &st))) return r;
    if (h->dctile && (r = dctile_status(h->dctile, &st2))) return r;
    if (st || st2) {
        set_error("chain: an inter-workgroup wait timed out on the device (fused status 0x%x, dc-tile status 0x%x): the output of the affected calls is invalid", st, st2);
        return CSDR_ERR_HIP;
    }
    return CSDR_OK;
}

int csdr_chain_status(csdr_chain *h)
{
    if (!h) { set_error("chain: null handle"); return CSDR_ERR_INVALID; }
    DevGuard guard(h->device);
    return chain_device_status(h);
}

int csdr_chain_submit_device(csdr_chain *h, const void *d_in, uint32_t n_in, void *d_out, uint32_t *n_out, void *ready_event)
{
    if (!h) { set_error("chain: null handle"); return CSDR_ERR_INVALID; }
    if (n_out) *n_out = 0;
    if (n_in == 0) return CSDR_OK;
    if (!d_in || !d_out) { set_error("chain: null buffer"); return CSDR_ERR_INVALID; }
    if (n_in % h->M) { set_error("chain: n_in=%u is not a multiple of channels=%u (the reference misbehaves here; Liquid.chs:832-862)", n_in, h->M); return CSDR_ERR_SIZE; }
    if (n_in > h->max_nx) { set_error("chain: n_in=%u exceeds max_frames*channels=%llu", n_in, (unsigned long long)h->max_nx); return CSDR_ERR_SIZE; }
    DevGuard guard(h->device);
    if (!guard.ok) { set_error("chain: cannot select device %d", h->device); return CSDR_ERR_HIP; }
    if (!h->s_pd[0]) {
        CSDR_HIP(hipDeviceSynchronize());               // whatever the handle has queued on caller streams so far
        // created into locals and published together: a partial failure leaves the handle as it was
        hipStream_t st[2] = {nullptr, nullptr}; hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        bool ok = true;
        for (int i = 0; i < 2 && ok; i++) ok = hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking) == hipSuccess;
        for (int i = 0; i < 6 && ok; i++) ok = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            for (hipStream_t q : st) if (q) (void)hipStreamDestroy(q);
            for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
            (void)hipGetLastError();
            set_error("chain: cannot create the streams / events of the pipelined entry point");
            return CSDR_ERR_HIP;
        }
        for (int i = 0; i < 2; i++) { h->s_pd[i] = st[i]; h->e_pd_done[i] = ev[i]; }
        for (int i = 0; i < 3; i++) h->e_pd_tail[i] = ev[2 + i];
        h->e_serial = ev[5];
        if (h->fused) fused_keep_tail(h->fused);
    }
    if (h->s_k && h->q_count) CSDR_HIP(hipStreamSynchronize(h->s_k));       // host-buffer chunks still in flight come first
    const uint32_t nf = n_in / h->M, k = h->pd_count;
    const int si = (int)(k & 1);
    hipStream_t s = h->s_pd[si];
    // An independent launch (fused M = 256 chain without AGC / AM / WBFM tails, run-kernel-sized whole-tile chunk, previous
    // chunk's tail on file) reads nothing an earlier launch writes: it only waits for the input and for the copy of the
    // previous chunk's tail.  Every other call is ordered behind everything the handle has in flight.
    const bool overlap = h->use_fused && h->fused && !h->d_agc && !h->am && !h->wbfm && fused_can_overlap(h->fused, nf);
    if (ready_event) CSDR_HIP(hipStreamWaitEvent(s, (hipEvent_t)ready_event, 0));
    if (h->serial_pending) { CSDR_HIP(hipStreamWaitEvent(s, h->e_serial, 0)); h->serial_pending = false; }
    if (overlap) {
        if (h->pd_tail_rec) CSDR_HIP(hipStreamWaitEvent(s, h->e_pd_tail[(k + 2) % 3], 0));       // recorded by call k - 1
        // a call k - 1 that was NOT independent reads the plan's ping-pong state (run 0) that this launch's last run overwrites:
        // then this launch follows all of it, not only its tail copy
        if (!h->pd_last_indep && h->pd_used[si ^ 1]) CSDR_HIP(hipStreamWaitEvent(s, h->e_pd_done[si ^ 1], 0));
    } else if (h->pd_used[si ^ 1]) CSDR_HIP(hipStreamWaitEvent(s, h->e_pd_done[si ^ 1], 0));
    h->call_indep = overlap; h->call_ev_tail = h->e_pd_tail[k % 3]; h->in_submit = true;
    const int r = csdr_chain_process_device(h, d_in, n_in, d_out, n_out, s);
    h->call_indep = false; h->call_ev_tail = nullptr; h->in_submit = false;
    if (r) return r;
    h->pd_tail_rec = h->fused && fused_tail_recorded(h->fused);
    if (overlap) h->pd_indep_calls++;
    h->pd_last_indep = overlap;
    CSDR_HIP(hipEventRecord(h->e_pd_done[si], s));
    h->pd_used[si] = true; h->pd_count = k + 1;
    return CSDR_OK;
}

uint32_t csdr_chain_debug_independent_launches(const csdr_chain *h) { return h ? h->pd_indep_calls : 0u; }

int csdr_chain_wait_device(csdr_chain *h, void *stream)
{
    if (!h) { set_error("chain: null handle"); return CSDR_ERR_INVALID; }
    DevGuard guard(h->device);
    for (int i = 0; i < 2; i++) {
        if (!h->pd_used[i]) continue;
        if (stream) CSDR_HIP(hipStreamWaitEvent((hipStream_t)stream, h->e_pd_done[i], 0));
        else CSDR_HIP(hipEventSynchronize(h->e_pd_done[i]));
    }
    return CSDR_OK;
}

void *csdr_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void csdr_host_free(void *p) { if (p) (void)hipHostFree(p); }

static bool is_pinned(const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

static int chain_host_init(csdr_chain *h)
{
    if (h->s_k) return CSDR_OK;
    CSDR_HIP(hipDeviceSynchronize());                   // whatever csdr_chain_process_device calls have queued on caller streams so far
    CSDR_HIP(hipStreamCreateWithFlags(&h->s_in, hipStreamNonBlocking));
    CSDR_HIP(hipStreamCreateWithFlags(&h->s_k, hipStreamNonBlocking));
    CSDR_HIP(hipStreamCreateWithFlags(&h->s_out, hipStreamNonBlocking));
    for (auto &sl : h->slot) {
        CSDR_HIP(hipEventCreateWithFlags(&sl.e_in, hipEventDisableTiming));
        CSDR_HIP(hipEventCreateWithFlags(&sl.e_k, hipEventDisableTiming));
        CSDR_HIP(hipEventCreateWithFlags(&sl.e_out, hipEventDisableTiming));
    }
    return CSDR_OK;
}

int csdr_chain_submit(csdr_chain *h, const float *in, uint32_t n_in, void *out)
{
    if (!h) { set_error("chain: null handle"); return CSDR_ERR_INVALID; }
    if (n_in && (!in || !out)) { set_error("chain: null buffer"); return CSDR_ERR_INVALID; }
    if (n_in % h->M) { set_error("chain: n_in=%u is not a multiple of channels=%u (the reference misbehaves here; Liquid.chs:832-862)", n_in, h->M); return CSDR_ERR_SIZE; }
    if (n_in > h->max_nx) { set_error("chain: n_in=%u exceeds max_frames*channels=%llu", n_in, (unsigned long long)h->max_nx); return CSDR_ERR_SIZE; }
    if (h->q_count == CSDR_CHAIN_INFLIGHT) { set_error("chain: %d chunks already in flight (collect first)", CSDR_CHAIN_INFLIGHT); return CSDR_ERR_BUSY; }
    DevGuard guard(h->device);
    if (!guard.ok) { set_error("chain: cannot select device %d", h->device); return CSDR_ERR_HIP; }
    int r;
    if ((r = chain_host_init(h))) return r;
    if (h->user_stream_dirty) {                         // csdr_chain_process_device calls on caller streams precede this chunk
        CSDR_HIP(hipEventSynchronize(h->e_user));
        h->user_stream_dirty = false;
    }
    csdr_chain::HostSlot &sl = h->slot[(h->q_head + h->q_count) % CSDR_CHAIN_INFLIGHT];
    sl.user_out = out; sl.n_out = 0; sl.out_bytes = 0; sl.staged_out = false;
    if (n_in) {
        const size_t in_bytes = sizeof(float2) * (size_t)n_in, out_max = (size_t)h->C * h->max_nf * 8;
        if (!sl.d_in && ((r = dev_alloc(&sl.d_in, h->max_nx)))) return r;
        if (!sl.d_out) { CSDR_HIP(hipMalloc(&sl.d_out, out_max)); }
        // every resource of the slot exists before the chain's state moves on: a failure below this point cannot skip a chunk's state
        const bool stage_out = !is_pinned(out);
        if (stage_out && !sl.h_out) { CSDR_HIP(hipHostMalloc(&sl.h_out, out_max, hipHostMallocDefault)); }
        const void *src = in;
        if (!is_pinned(in)) {                               // pageable caller memory: one host copy into the slot's page-locked buffer
            if (!sl.h_in) { CSDR_HIP(hipHostMalloc(&sl.h_in, sizeof(float2) * h->max_nx, hipHostMallocDefault)); }
            memcpy(sl.h_in, in, in_bytes);
            src = sl.h_in;
        }
        CSDR_HIP(hipMemcpyAsync(sl.d_in, src, in_bytes, hipMemcpyHostToDevice, h->s_in));
        CSDR_HIP(hipEventRecord(sl.e_in, h->s_in));
        CSDR_HIP(hipStreamWaitEvent(h->s_k, sl.e_in, 0));
        // the slot's device output was last read by its own previous D2H copy
        CSDR_HIP(hipStreamWaitEvent(h->s_k, sl.e_out, 0));
        uint32_t no = 0;
        if ((r = csdr_chain_process_device(h, sl.d_in, n_in, sl.d_out, &no, h->s_k))) return r;
        CSDR_HIP(hipEventRecord(sl.e_k, h->s_k));
        sl.n_out = no; sl.out_bytes = (size_t)no * csdr_chain_out_elem_size(h);
        void *dst = out;
        if (stage_out) { dst = sl.h_out; sl.staged_out = true; }
        CSDR_HIP(hipStreamWaitEvent(h->s_out, sl.e_k, 0));
        CSDR_HIP(hipMemcpyAsync(dst, sl.d_out, sl.out_bytes, hipMemcpyDeviceToHost, h->s_out));
        CSDR_HIP(hipEventRecord(sl.e_out, h->s_out));
    }
    h->q_count++;
    return CSDR_OK;
}

int csdr_chain_collect(csdr_chain *h, uint32_t *n_out)
{
    if (!h) { set_error("chain: null handle"); return CSDR_ERR_INVALID; }
    if (n_out) *n_out = 0;
    if (!h->q_count) { set_error("chain: nothing submitted"); return CSDR_ERR_INVALID; }
    DevGuard guard(h->device);
    csdr_chain::HostSlot &sl = h->slot[h->q_head];
    if (sl.out_bytes) {
        CSDR_HIP(hipEventSynchronize(sl.e_out));        // (a failed wait leaves the chunk queued)
        if (sl.staged_out) memcpy(sl.user_out, sl.h_out, sl.out_bytes);
    }
    h->q_head = (h->q_head + 1) % CSDR_CHAIN_INFLIGHT; h->q_count--;
    if (n_out) *n_out = sl.n_out;
    return CSDR_OK;
}

int csdr_chain_process(csdr_chain *h, const float *in, uint32_t n_in, void *out, uint32_t *n_out)
{
    if (!h) { set_error("chain: null handle"); return CSDR_ERR_INVALID; }
    if (n_out) *n_out = 0;
    if (n_in == 0) return CSDR_OK;
    if (h->q_count) { set_error("chain: csdr_chain_process with chunks still in flight (collect them first)"); return CSDR_ERR_BUSY; }
    int r;
    if (is_pinned(in) && is_pinned(out)) {
        if ((r = csdr_chain_submit(h, in, n_in, out))) return r;
        if ((r = csdr_chain_collect(h, n_out))) return r;
    } else {
        // pageable caller memory: the runtime's own staged hipMemcpy beats a hand-made copy into page-locked memory
        if (!in || !out) { set_error("chain: null buffer"); return CSDR_ERR_INVALID; }
        if (n_in % h->M) { set_error("chain: n_in=%u is not a multiple of channels=%u (the reference misbehaves here; Liquid.chs:832-862)", n_in, h->M); return CSDR_ERR_SIZE; }
        if (n_in > h->max_nx) { set_error("chain: n_in=%u exceeds max_frames*channels=%llu", n_in, (unsigned long long)h->max_nx); return CSDR_ERR_SIZE; }
        DevGuard guard(h->device);
        if (!guard.ok) { set_error("chain: cannot select device %d", h->device); return CSDR_ERR_HIP; }
        if ((r = chain_host_init(h))) return r;
        if (h->user_stream_dirty) { CSDR_HIP(hipEventSynchronize(h->e_user)); h->user_stream_dirty = false; }
        csdr_chain::HostSlot &sl = h->slot[0];
        if (!sl.d_in && ((r = dev_alloc(&sl.d_in, h->max_nx)))) return r;
        if (!sl.d_out) { CSDR_HIP(hipMalloc(&sl.d_out, (size_t)h->C * h->max_nf * 8)); }
        CSDR_HIP(hipMemcpy(sl.d_in, in, sizeof(float2) * (size_t)n_in, hipMemcpyHostToDevice));
        uint32_t no = 0;
        if ((r = csdr_chain_process_device(h, sl.d_in, n_in, sl.d_out, &no, h->s_k))) return r;
        CSDR_HIP(hipStreamSynchronize(h->s_k));
        CSDR_HIP(hipMemcpy(out, sl.d_out, (size_t)no * csdr_chain_out_elem_size(h), hipMemcpyDeviceToHost));
        if (n_out) *n_out = no;
    }
    if ((r = chain_device_status(h))) return r;
    return CSDR_OK;
}

int csdr_chain_reset(csdr_chain *h)
{
    if (!h) return CSDR_ERR_INVALID;
    DevGuard guard(h->device);
    CSDR_HIP(hipDeviceSynchronize());                  // chunks still in flight are abandoned
    h->q_head = 0; h->q_count = 0;
    h->pd_used[0] = h->pd_used[1] = false; h->pd_tail_rec = false; h->serial_pending = false; h->pd_count = 0;
    h->user_stream_dirty = false; h->pd_last_indep = false;
    int r = chain_init_state(h, nullptr);
    if (r) return r;
    CSDR_HIP(hipDeviceSynchronize());
    return chain_device_status(h);
}

int csdr_chain_seek_frames(csdr_chain *h, uint64_t frames)
{
    int r = csdr_chain_reset(h);
    if (r) return r;
    const uint64_t n = frames * (uint64_t)h->M;
    h->theta = (uint32_t)(n * (uint64_t)h->d_theta);
    if (h->tab_len) h->tab_pos = (uint32_t)(n % h->tab_len);
    if (h->fused) fused_seek(h->fused, frames);
    if (h->small) small_seek(h->small, frames);
    if (h->big) big_seek(h->big, frames);
    if (h->huge) huge_seek(h->huge, frames);
    return CSDR_OK;
}

int csdr_chain_get_taps(const csdr_chain *h, float *taps, uint32_t n)
{
    if (!h || !taps) return CSDR_ERR_INVALID;
    if (n > h->taps.size()) n = (uint32_t)h->taps.size();
    memcpy(taps, h->taps.data(), sizeof(float) * n);
    return (int)n;
}
int csdr_chain_get_nco(const csdr_chain *h, uint32_t *theta, uint32_t *d_theta)
{
    if (!h) return CSDR_ERR_INVALID;
    if (theta) *theta = h->theta;
    if (d_theta) *d_theta = h->d_theta;
    return CSDR_OK;
}
int csdr_chain_debug_trace(csdr_chain *h, unsigned long long *out, uint32_t ntiles)
{
    if (!h || !out) return CSDR_ERR_INVALID;
    if (!h->fused) return 0;
    (void)hipDeviceSynchronize();
    return fused_trace(h->fused, out, ntiles);
}
int csdr_chain_debug_agc(csdr_chain *h, uint32_t *checked, uint32_t *redone)
{
    if (!h) return CSDR_ERR_INVALID;
    if (checked) *checked = 0;
    if (redone) *redone = 0;
    if (!h->agc_tail) return 0;
    DevGuard guard(h->device);
    (void)hipDeviceSynchronize();
    return agc_tail_stats(h->agc_tail, checked, redone);
}
uint32_t csdr_chain_debug_agc_tile_major_calls(const csdr_chain *h) { return h && h->agc_tail ? agc_tail_tm_calls(h->agc_tail) : 0u; }
const char *csdr_chain_path(const csdr_chain *h) { return h ? h->path.c_str() : ""; }
int csdr_chain_get_cfg(const csdr_chain *h, csdr_chain_cfg *cfg_out)
{
    if (!h || !cfg_out) { set_error("csdr_chain_get_cfg: null argument"); return CSDR_ERR_INVALID; }
    *cfg_out = h->cfg;
    return CSDR_OK;
}

const char *csdr_chain_kernel_time(csdr_chain *h, double *total_ms, uint32_t *launches)
{
    if (!h) return "";
    if (h->s_pd[0] && h->timer.region && h->timer.open && h->timer.last) {
        DevGuard guard(h->device);
        for (int i = 0; i < 2; i++) if (h->pd_used[i]) (void)hipStreamWaitEvent(h->timer.last, h->e_pd_done[i], 0);
    }
    (void)h->timer.drain();
    if (total_ms) *total_ms = h->timer.acc_ms;
    if (launches) *launches = h->timer.launches;
    h->timer.acc_ms = 0.0; h->timer.launches = 0;
    if (h->fused) h->timed_kernel = fused_name(h->fused);      // the kernel the last call launched
    return h->timed_kernel.c_str();
}

int csdr_chain_destroy(csdr_chain *h)
{
    if (!h) return CSDR_OK;
    DevGuard guard(h->device);
    (void)hipDeviceSynchronize();
    if (h->fused) fused_destroy(h->fused);
    if (h->small) small_destroy(h->small);
    if (h->big) big_destroy(h->big);
    if (h->huge) huge_destroy(h->huge);
    if (h->dctile) dctile_destroy(h->dctile);
    if (h->agc_tail) agc_tail_destroy(h->agc_tail);
    h->timer.destroy();
    void *ptrs[] = {h->d_taps, h->d_tw, h->d_nco_tab, h->d_dcstate, h->d_scratch, h->d_u, h->d_hist_tmp, h->d_A, h->d_B,
                    h->d_agc, h->d_rp[0], h->d_rp[1], h->d_amz, h->d_amf, h->d_amq[0], h->d_amq[1], h->d_tw_g, h->d_fold_ph, h->d_fold, h->d_u0, h->d_u0hist,
                    h->d_wbf, h->d_wbo, h->d_wbh, h->d_wbhist[0], h->d_wbhist[1], h->d_wbst[0], h->d_wbst[1], h->d_perm};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (auto &sl : h->slot) {
        if (sl.d_in) (void)hipFree(sl.d_in);
        if (sl.d_out) (void)hipFree(sl.d_out);
        if (sl.h_in) (void)hipHostFree(sl.h_in);
        if (sl.h_out) (void)hipHostFree(sl.h_out);
        for (hipEvent_t e : {sl.e_in, sl.e_k, sl.e_out}) if (e) (void)hipEventDestroy(e);
    }
    for (hipStream_t st : {h->s_in, h->s_k, h->s_out, h->s_pd[0], h->s_pd[1]}) if (st) (void)hipStreamDestroy(st);
    for (hipEvent_t e : {h->e_pd_done[0], h->e_pd_done[1], h->e_pd_tail[0], h->e_pd_tail[1], h->e_pd_tail[2], h->e_serial, h->e_user}) if (e) (void)hipEventDestroy(e);
    delete h;
    return CSDR_OK;
}

}  // extern "C"
