// Host-side filter / oscillator design for libcsdr_hip.so (product code).
// Mirrors what the reference obtains from liquid-dsp at object creation:
//   firpfbch_crcf_create_kaiser(0, M, 7, 80.0)            Liquid.chs:813
//   nco_crcf_create(LIQUID_VCO) + set_frequency(offset)    Liquid.chs:816-818
//   agc_crcf_squelch_set_threshold                         Liquid.chs:713
#include "csdr_internal.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

namespace csdr {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char *last_error() { return g_err; }

int hip_fail(hipError_t e, const char *what, const char *file, int line)
{
    set_error("HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    return e == hipErrorOutOfMemory ? -5 : -2;
}

// Modified Bessel function I0 by its power series (converged to f64 precision).
static double bessel_i0(double z)
{
    double q = 0.25 * z * z, term = 1.0, sum = 1.0;
    for (int k = 1; k < 200; k++) {
        term *= q / ((double)k * (double)k);
        sum += term;
        if (term < 1e-18 * sum) break;
    }
    return sum;
}

static double kaiser_beta(double As)
{
    As = std::fabs(As);
    if (As > 50.0) return 0.1102 * (As - 8.7);
    if (As > 21.0) return 0.5842 * std::pow(As - 21.0, 0.4) + 0.07886 * (As - 21.0);
    return 0.0;
}

std::vector<float> design_pfb_taps(uint32_t M, uint32_t m, float As)
{
    const uint32_t N = 2 * M * m + 1;          // designed length; the bank uses N-1 taps
    const double fc = 0.5 / (double)M;
    const double beta = kaiser_beta(As), ib = bessel_i0(beta);
    const double pi = 3.14159265358979323846;
    std::vector<float> h((size_t)M * 2 * m);
    for (uint32_t i = 0; i < N - 1; i++) {
        double t = (double)i - 0.5 * (double)(N - 1);
        double x = 2.0 * fc * t;
        // liquid's sincf uses a cosine product below |x| < 0.01 (exact to ~1e-10 there)
        double sinc = std::fabs(x) < 0.01
                          ? std::cos(pi * x / 2) * std::cos(pi * x / 4) * std::cos(pi * x / 8)
                          : std::sin(pi * x) / (pi * x);
        double r = 2.0 * t / (double)(N - 1);
        double a = 1.0 - r * r;
        double w = bessel_i0(beta * std::sqrt(a > 0 ? a : 0)) / ib;
        h[i] = (float)(sinc * w);
    }
    return h;
}

// liquid_firdes_kaiser(N, fc, As, 0): h[i] = sinc(2 fc t) w_kaiser(i), t = i - (N-1)/2 (f64)
static std::vector<double> firdes_kaiser(uint32_t N, double fc, double As)
{
    const double beta = kaiser_beta(As), ib = bessel_i0(beta);
    const double pi = 3.14159265358979323846;
    std::vector<double> h(N);
    for (uint32_t i = 0; i < N; i++) {
        double t = (double)i - 0.5 * (double)(N - 1);
        double x = 2.0 * fc * t;
        double sinc = std::fabs(x) < 0.01
                          ? std::cos(pi * x / 2) * std::cos(pi * x / 4) * std::cos(pi * x / 8)
                          : std::sin(pi * x) / (pi * x);
        double r = 2.0 * t / (double)(N - 1);
        double a = 1.0 - r * r;
        h[i] = sinc * bessel_i0(beta * std::sqrt(a > 0 ? a : 0)) / ib;
    }
    return h;
}

// ---- multi-stage resampler (msresamp_crcf's structure; "csdr msresamp v1" parameters) ----
ResampDesign design_msresamp(float rate, float As)
{
    ResampDesign d;
    d.rate = rate; d.rho = (double)rate; d.K = 0;
    while (d.rho < 0.5 && d.K < 24) { d.K++; d.rho *= 2.0; }
    for (uint32_t s = 0; s < d.K; s++) {
        // half-band stage s sees the final band at fb = 0.45 r 2^s of its input rate
        double fb = 0.45 * (double)rate * (double)(1u << s), ft = 0.5 - 2.0 * fb;
        if (ft < 0.01) ft = 0.01;
        double N = (std::fabs((double)As) - 7.95) / (14.36 * ft);
        int m = (int)std::ceil((N - 1.0) / 4.0);
        if (m < 2) m = 2;
        std::vector<double> hd = firdes_kaiser(4 * m + 1, 0.25, As);
        std::vector<float> h(4 * m + 1);
        for (size_t i = 0; i < h.size(); i++) h[i] = (float)(0.5 * hd[i]);
        d.m_hb.push_back((uint32_t)m); d.h_hb.push_back(h);
    }
    d.npfb = 256; d.m_arb = 7;
    double fc = 0.515 * d.rho; if (fc > 0.49) fc = 0.49;
    d.fc = (float)fc;
    const uint32_t P = 2 * d.m_arb;
    std::vector<double> hd = firdes_kaiser(P * d.npfb + 1, fc / (double)d.npfb, As);
    d.pfb.resize((size_t)d.npfb * P);
    for (uint32_t b = 0; b < d.npfb; b++)
        for (uint32_t j = 0; j < P; j++) d.pfb[(size_t)b * P + j] = (float)(2.0 * fc * hd[b + (size_t)j * d.npfb]);
    d.delta = (uint64_t)std::llround(4294967296.0 / d.rho);
    return d;
}

// 2nd-order Butterworth low-pass by the bilinear transform with pre-warping (liquid iirdes BUTTER/LOWPASS/SOS, order 2)
BiquadParams design_butter2_lowpass(float fc)
{
    BiquadParams p{};
    const double K = std::tan(3.14159265358979323846 * (double)fc), n = 1.0 / (1.0 + std::sqrt(2.0) * K + K * K);
    p.b0 = (float)(K * K * n); p.b1 = (float)(2.0 * K * K * n); p.b2 = p.b0;
    p.a1 = (float)(2.0 * (K * K - 1.0) * n); p.a2 = (float)((1.0 - std::sqrt(2.0) * K + K * K) * n);
    // powers of the state matrix of the f32 coefficients: A^(16), A^(32), ... A^(2048)
    double A[4] = {-(double)p.a1, -(double)p.a2, 1.0, 0.0};
    auto mul = [](const double *x, const double *y, double *z) {
        double t[4] = {x[0] * y[0] + x[1] * y[2], x[0] * y[1] + x[1] * y[3], x[2] * y[0] + x[3] * y[2], x[2] * y[1] + x[3] * y[3]};
        for (int i = 0; i < 4; i++) z[i] = t[i];
    };
    double P[4] = {A[0], A[1], A[2], A[3]};
    for (int i = 0; i < 4; i++) mul(P, P, P);                    // A^16
    for (int k = 0; k < 8; k++) {
        for (int i = 0; i < 4; i++) p.pw[k][i] = P[i];
        mul(P, P, P);
    }
    return p;
}

std::vector<float> design_firdecim_kaiser(uint32_t M, uint32_t m, float As)
{
    const uint32_t N = 2 * M * m + 1;
    std::vector<double> hd = firdes_kaiser(N, 0.5 / (double)M, As);
    std::vector<float> h(N);
    for (uint32_t i = 0; i < N; i++) h[i] = (float)hd[i];
    return h;
}

uint32_t nco_freq_word(float freq)
{
    float p = (float)((double)freq * 0.159154943091895);   // freq / 2pi, rounded to f32
    float fpart = p - (float)((long)p);
    if (fpart < 0.0f) fpart += 1.0f;
    return (uint32_t)(fpart * (float)0xffffffffu);
}

float pfb_premix_freq(uint32_t M)
{
    // evaluated left to right in binary32 like the Haskell expression
    float n = (float)M;
    float a = -0.5f * (n - 1.0f);
    a = a / n;
    a = a * 2.0f;
    a = a * (float)3.14159265358979323846;
    return a;
}

void nco_phasor(uint32_t theta, float *c, float *s)
{
    // nco_crcf_get_phase(): 2*pi*theta/2^32 with theta converted to f32 first
    float ph = (float)(2.0 * 3.14159265358979323846 * (double)(float)theta / 4294967296.0);
    *c = cosf(ph);
    *s = sinf(ph);
}

uint32_t nco_period(uint32_t d_theta, uint32_t limit)
{
    if (d_theta == 0) return 1;
    // period = 2^32 / gcd(d_theta, 2^32) = 2^(32 - ctz(d_theta))
    int tz = __builtin_ctz(d_theta);
    uint64_t per = 1ull << (32 - tz);
    return per <= limit ? (uint32_t)per : 0;
}

float agc_gain_threshold(float thr_db)
{
    // rssi(g) = (float)(-20*log10((double)g)) is non-increasing in g.  Find the smallest
    // positive f32 g with rssi(g) <= thr by bisection on the f32 bit pattern.
    auto exceeded = [&](float g) { return (float)(-20.0 * std::log10((double)g)) > thr_db; };
    uint32_t lo = 0x00800000u;     // smallest normal: rssi ~ +758 dB -> exceeded
    uint32_t hi = 0x7f7fffffu;     // FLT_MAX: rssi ~ -770 dB
    float flo, fhi;
    memcpy(&flo, &lo, 4); memcpy(&fhi, &hi, 4);
    if (!exceeded(flo)) return 0.0f;                 // never exceeded
    if (exceeded(fhi)) return INFINITY;              // always exceeded
    while (hi - lo > 1) {
        uint32_t mid = lo + (hi - lo) / 2;
        float fm; memcpy(&fm, &mid, 4);
        if (exceeded(fm)) lo = mid; else hi = mid;
    }
    memcpy(&fhi, &hi, 4);
    return fhi;
}

}  // namespace csdr
