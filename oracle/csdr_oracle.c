/*
 * csdr_oracle.c -- CPU restatement of the reference's DSP hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under composable_sdr_amd/ may include,
 * link or call this file; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, as the checker / the CPU baseline.
 *
 * What it restates
 * ----------------
 * The reference (mryndzionek/composable-sdr) is Haskell glue around liquid-dsp
 * (pinned v1.3.2 by /root/reference/.github/workflows/build.yml:15).  liquid-dsp
 * is NOT vendored in the reference and is not installed in this image, so the
 * arithmetic below restates liquid-dsp 1.3.2's published algorithms
 * (src/filter/src/firdes.c, src/math/src/windows.c, src/multichannel/src/
 * firpfbch.c, src/nco/src/nco.c, src/filter/src/iirfilt.c, src/agc/src/agc.c,
 * src/modem/src/freqdem.c) and anchors them on the reference's own call
 * sites in src/ComposableSDR/Liquid.chs (cited per function).
 *
 * Pin status: PARITY UNPINNED by any reference test (the reference has none,
 * README.md:306).  Anchors recovered from the reference's own artifact
 * images/ex1_5.gif and checked in tests/test_oracle_kat.py:
 *   KAT1  firpfbch_crcf_print taps for M=20, m=7, As=80 (31 values)
 *   KAT2  NCO frequency word 0x86666600 for M=20
 *   KAT3  DC blocker coefficient form b=[1,-1], a=[1,-(1-alpha)]
 *   KAT4  README Example 3 sizes (n/M samples per channel, 8 B each)
 * Not anchored (recalled only): FFT direction / output ordering of
 * firpfbch analyzer, agc_crcf internals, freqdem scaling.
 *
 * Numerics: f32 state and f32 operation order as in liquid's portable C
 * paths; filter design and the DFT are evaluated in f64 and rounded once
 * (liquid's FFTW/SIMD summation orders are not reproducible anyway; see
 * SURVEY.md Appendix A.8).  Build with -ffp-contract=off.
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef struct { float re, im; } cf32;

/* ------------------------------------------------------------------ */
/* Kaiser prototype: liquid_firdes_kaiser(2*M*m+1, 0.5/M, As, 0)       */
/* called from firpfbch_crcf_create_kaiser (Liquid.chs:813).           */
/* ------------------------------------------------------------------ */

/* liquid_besseli0f: 32-term series sum_k ((z/2)^k / k!)^2 */
static double orc_besseli0(double z)
{
    if (z == 0.0) return 1.0;
    double y = 0.0;
    for (int k = 0; k < 32; k++) {
        double t = k * log(0.5 * z) - lgamma((double)k + 1.0);
        y += exp(2.0 * t);
    }
    return y;
}

/* sincf: sin(pi x)/(pi x); liquid switches to a cosine product near 0 */
static double orc_sinc(double x)
{
    if (fabs(x) < 0.01)
        return cos(M_PI * x / 2.0) * cos(M_PI * x / 4.0) * cos(M_PI * x / 8.0);
    return sin(M_PI * x) / (M_PI * x);
}

/* kaiser_beta_As */
static double orc_kaiser_beta(double As)
{
    As = fabs(As);
    if (As > 50.0) return 0.1102 * (As - 8.7);
    if (As > 21.0) return 0.5842 * pow(As - 21.0, 0.4) + 0.07886 * (As - 21.0);
    return 0.0;
}

/* h[0 .. 2*M*m] ; validated by KAT1 (window argument is 2t/(N-1)). */
void orc_kaiser_prototype(unsigned M, unsigned m, float As, float *h)
{
    unsigned N = 2 * M * m + 1;
    double fc = 0.5 / (double)M;
    double beta = orc_kaiser_beta(As);
    double ib = orc_besseli0(beta);
    for (unsigned i = 0; i < N; i++) {
        double t = (double)i - (double)(N - 1) / 2.0;
        double r = 2.0 * t / (double)(N - 1);
        double a = 1.0 - r * r;
        if (a < 0.0) a = 0.0;
        double w = orc_besseli0(beta * sqrt(a)) / ib;
        h[i] = (float)(orc_sinc(2.0 * fc * t) * w);
    }
}

/* ------------------------------------------------------------------ */
/* nco_crcf, VCO type, uint32 phase accumulator (nco.c).               */
/* Reference: ncoCreate Liquid.chs:782-788, firpfbchCreate :816-818.   */
/* ------------------------------------------------------------------ */

/* nco_crcf_constrain: validated by KAT2 */
uint32_t orc_nco_constrain(float theta)
{
    float p = theta * 0.159154943091895;      /* f32 <- f64 product */
    float fpart = p - ((long)p);
    if (fpart < 0.) fpart += 1.;
    return (uint32_t)(fpart * 0xffffffff);      /* f32 * (f32)0xffffffff */
}

/* Haskell: -0.5 * (fromIntegral n - 1) / fromIntegral n * 2 * pi  :: Float */
float orc_pfb_offset(unsigned M)
{
    float n = (float)M;
    float a = -0.5f * (n - 1.0f);
    a = a / n;
    a = a * 2.0f;
    a = a * (float)M_PI;
    return a;
}

typedef struct { uint32_t theta, d_theta; } orc_nco;

orc_nco *orc_nco_create(float freq)
{
    orc_nco *q = (orc_nco *)calloc(1, sizeof(*q));
    q->theta = 0;
    q->d_theta = orc_nco_constrain(freq);
    return q;
}
void orc_nco_destroy(orc_nco *q) { free(q); }
uint32_t orc_nco_get_theta(const orc_nco *q) { return q->theta; }
uint32_t orc_nco_get_dtheta(const orc_nco *q) { return q->d_theta; }
void orc_nco_set_theta(orc_nco *q, uint32_t t) { q->theta = t; }

/* phasor for an integer phase: nco_crcf_get_phase then sinf/cosf */
static inline void orc_nco_sincos(uint32_t theta, float *s, float *c)
{
    float ph = 2.0f * M_PI * (float)theta / (float)(1LLU << 32);
    *s = sinf(ph);
    *c = cosf(ph);
}
void orc_nco_phasor(uint32_t theta, float *sc) { orc_nco_sincos(theta, &sc[0], &sc[1]); }

/* y = x * conj(v) ; v = cos + j sin   (nco_crcf_mix_block_down) */
void orc_nco_mix_down(orc_nco *q, const cf32 *x, cf32 *y, unsigned n)
{
    for (unsigned i = 0; i < n; i++) {
        float s, c;
        orc_nco_sincos(q->theta, &s, &c);
        float ns = -s;                              /* conj(v) = c + j(-s) */
        float re = x[i].re * c - x[i].im * ns;
        float im = x[i].re * ns + x[i].im * c;
        y[i].re = re; y[i].im = im;
        q->theta += q->d_theta;
    }
}
/* y = x * v  (nco_crcf_mix_block_up) */
void orc_nco_mix_up(orc_nco *q, const cf32 *x, cf32 *y, unsigned n)
{
    for (unsigned i = 0; i < n; i++) {
        float s, c;
        orc_nco_sincos(q->theta, &s, &c);
        float re = x[i].re * c - x[i].im * s;
        float im = x[i].re * s + x[i].im * c;
        y[i].re = re; y[i].im = im;
        q->theta += q->d_theta;
    }
}

/* ------------------------------------------------------------------ */
/* iirfilt_crcf_create_dc_blocker(alpha) + execute_block (iirfilt.c).  */
/* Reference: dcBlocker Liquid.chs:575-589 (alpha = 0.0005).           */
/* b = [1,-1], a = [1, -(1-alpha)] (KAT3); direct form II:             */
/*   v0 = x - a1*v1 ; y = b0*v0 + b1*v1 ; v1 <- v0                      */
/* ------------------------------------------------------------------ */
typedef struct { float a1; cf32 v1; } orc_dcblock;

orc_dcblock *orc_dcblock_create(float alpha)
{
    orc_dcblock *q = (orc_dcblock *)calloc(1, sizeof(*q));
    q->a1 = -1.0f + alpha;
    return q;
}
void orc_dcblock_destroy(orc_dcblock *q) { free(q); }
float orc_dcblock_a1(const orc_dcblock *q) { return q->a1; }
void orc_dcblock_get_state(const orc_dcblock *q, float *v) { v[0] = q->v1.re; v[1] = q->v1.im; }

void orc_dcblock_execute(orc_dcblock *q, const cf32 *x, unsigned n, cf32 *y)
{
    float a1 = q->a1;
    cf32 v1 = q->v1;
    for (unsigned i = 0; i < n; i++) {
        cf32 v0;
        v0.re = x[i].re - a1 * v1.re;
        v0.im = x[i].im - a1 * v1.im;
        y[i].re = v0.re - v1.re;                    /* 1*v0 + (-1)*v1 */
        y[i].im = v0.im - v1.im;
        v1 = v0;
    }
    q->v1 = v1;
}

/* ------------------------------------------------------------------ */
/* firpfbch_crcf (LIQUID_ANALYZER) -- firpfbch.c.                       */
/* Reference: firpfbchCreate Liquid.chs:811-821, analyzer_execute :843. */
/* ------------------------------------------------------------------ */
typedef struct {
    unsigned M, p;
    float *h;          /* first M*p prototype taps                         */
    float *hsub;       /* [M][p] reversed sub-filters: hsub[i][p-1-n]=h[i+nM] */
    cf32 *w;           /* [M][p] windows, w[i][0] oldest .. w[i][p-1] newest  */
    unsigned filter_index;
    double *twr, *twi; /* e^{-j 2 pi k / M}                                 */
    double *fr, *fi;   /* FFT work                                          */
    int pow2;
} orc_pfb;

orc_pfb *orc_pfb_create(unsigned M, unsigned m, float As)
{
    orc_pfb *q = (orc_pfb *)calloc(1, sizeof(*q));
    q->M = M; q->p = 2 * m;
    unsigned N = 2 * M * m + 1;
    float *hfull = (float *)malloc(sizeof(float) * N);
    orc_kaiser_prototype(M, m, As, hfull);
    q->h = (float *)malloc(sizeof(float) * M * q->p);
    memcpy(q->h, hfull, sizeof(float) * M * q->p);   /* last designed tap dropped */
    free(hfull);
    q->hsub = (float *)malloc(sizeof(float) * M * q->p);
    for (unsigned i = 0; i < M; i++)
        for (unsigned n = 0; n < q->p; n++)
            q->hsub[i * q->p + (q->p - n - 1)] = q->h[i + n * M];
    q->w = (cf32 *)calloc((size_t)M * q->p, sizeof(cf32));
    q->filter_index = M - 1;
    q->twr = (double *)malloc(sizeof(double) * M);
    q->twi = (double *)malloc(sizeof(double) * M);
    for (unsigned k = 0; k < M; k++) {
        q->twr[k] = cos(-2.0 * M_PI * (double)k / (double)M);
        q->twi[k] = sin(-2.0 * M_PI * (double)k / (double)M);
    }
    q->fr = (double *)malloc(sizeof(double) * 2 * M);
    q->fi = (double *)malloc(sizeof(double) * 2 * M);
    q->pow2 = (M & (M - 1)) == 0;
    return q;
}
void orc_pfb_destroy(orc_pfb *q)
{
    if (!q) return;
    free(q->h); free(q->hsub); free(q->w); free(q->twr); free(q->twi); free(q->fr); free(q->fi); free(q);
}
const float *orc_pfb_taps(const orc_pfb *q) { return q->h; }
/* The direction of firpfbch_crcf_analyzer_execute's transform (Liquid.chs:843) is RECALLED (forward, e^{-j}): nothing in the
 * reference pins it (hard part 1).  backward != 0 makes every later analyzer_execute use e^{+j 2 pi jk/M}
 * instead -- the other possible convention -- by conjugating the twiddle table; the product's CSDR_FLAG_DFT_BACKWARD is tested
 * against this. */
void orc_pfb_set_dft_backward(orc_pfb *q, int backward)
{
    for (unsigned k = 0; k < q->M; k++)
        q->twi[k] = sin((backward ? 2.0 : -2.0) * M_PI * (double)k / (double)q->M);
}

/* forward, unnormalised DFT of X[0..M) (f64 inside) */
static void orc_dft_forward(orc_pfb *q, const cf32 *X, cf32 *y)
{
    unsigned M = q->M;
    double *ar = q->fr, *ai = q->fi;
    if (q->pow2 && M > 1) {
        /* iterative radix-2 DIT with bit reversal */
        unsigned lg = 0; while ((1u << lg) < M) lg++;
        for (unsigned i = 0; i < M; i++) {
            unsigned r = 0;
            for (unsigned b = 0; b < lg; b++) if (i & (1u << b)) r |= 1u << (lg - 1 - b);
            ar[r] = X[i].re; ai[r] = X[i].im;
        }
        for (unsigned len = 2; len <= M; len <<= 1) {
            unsigned half = len >> 1, step = M / len;
            for (unsigned s = 0; s < M; s += len)
                for (unsigned k = 0; k < half; k++) {
                    double wr = q->twr[k * step], wi = q->twi[k * step];
                    double xr = ar[s + k + half], xi = ai[s + k + half];
                    double tr = xr * wr - xi * wi, ti = xr * wi + xi * wr;
                    ar[s + k + half] = ar[s + k] - tr; ai[s + k + half] = ai[s + k] - ti;
                    ar[s + k] += tr; ai[s + k] += ti;
                }
        }
        for (unsigned k = 0; k < M; k++) { y[k].re = (float)ar[k]; y[k].im = (float)ai[k]; }
    } else {
        for (unsigned k = 0; k < M; k++) {
            double sr = 0.0, si = 0.0;
            for (unsigned j = 0; j < M; j++) {
                unsigned idx = (unsigned)(((unsigned long long)j * k) % M);
                double wr = q->twr[idx], wi = q->twi[idx];
                sr += X[j].re * wr - X[j].im * wi;
                si += X[j].re * wi + X[j].im * wr;
            }
            y[k].re = (float)sr; y[k].im = (float)si;
        }
    }
}

/* firpfbch_crcf_analyzer_execute: push M samples, run, DFT. */
void orc_pfb_analyzer_execute(orc_pfb *q, const cf32 *x, cf32 *y)
{
    unsigned M = q->M, p = q->p;
    for (unsigned i = 0; i < M; i++) {
        cf32 *w = q->w + (size_t)q->filter_index * p;
        memmove(w, w + 1, sizeof(cf32) * (p - 1));     /* window_push */
        w[p - 1] = x[i];
        q->filter_index = (q->filter_index + M - 1) % M;
    }
    cf32 *X = (cf32 *)alloca(sizeof(cf32) * M);
    for (unsigned i = 0; i < M; i++) {
        const cf32 *w = q->w + (size_t)i * p;           /* index = (i+0) % M */
        const float *h = q->hsub + (size_t)i * p;
        float sr = 0.0f, si = 0.0f;                     /* dotprod_crcf, in order */
        for (unsigned n = 0; n < p; n++) { sr += h[n] * w[n].re; si += h[n] * w[n].im; }
        X[M - i - 1].re = sr; X[M - i - 1].im = si;
    }
    orc_dft_forward(q, X, y);
}

/* ------------------------------------------------------------------ */
/* firpfbchChan (Liquid.chs:828-862): premix + per-frame analyzer +     */
/* transpose to channel-major [M][nf].                                  */
/* ------------------------------------------------------------------ */
typedef struct { orc_pfb *fb; orc_nco *nco; unsigned M; } orc_chan;

orc_chan *orc_chan_create(unsigned M)
{
    orc_chan *q = (orc_chan *)calloc(1, sizeof(*q));
    q->M = M;
    q->fb = orc_pfb_create(M, 7, 80.0f);                /* Liquid.chs:813 */
    q->nco = orc_nco_create(orc_pfb_offset(M));         /* Liquid.chs:816-818 */
    return q;
}
void orc_chan_destroy(orc_chan *q) { if (q) { orc_pfb_destroy(q->fb); orc_nco_destroy(q->nco); free(q); } }
void orc_chan_set_dft_backward(orc_chan *q, int backward) { orc_pfb_set_dft_backward(q->fb, backward); }
uint32_t orc_chan_dtheta(const orc_chan *q) { return q->nco->d_theta; }
uint32_t orc_chan_theta(const orc_chan *q) { return q->nco->theta; }

/* x: nx samples (nx multiple of M) ; y: [M][nf] */
void orc_chan_process(orc_chan *q, const cf32 *x, unsigned nx, cf32 *y)
{
    unsigned M = q->M, nf = nx / M;
    cf32 *dx = (cf32 *)malloc(sizeof(cf32) * (nx ? nx : 1));
    cf32 *tmp = (cf32 *)malloc(sizeof(cf32) * M);
    orc_nco_mix_down(q->nco, x, dx, nx);
    for (unsigned i = 0; i < nf; i++) {
        orc_pfb_analyzer_execute(q->fb, dx + (size_t)M * i, tmp);
        for (unsigned j = 0; j < M; j++) y[(size_t)nf * j + i] = tmp[j];
    }
    free(dx); free(tmp);
}

/* ------------------------------------------------------------------ */
/* agc_crcf as configured by agcCreate (Liquid.chs:707-717) and driven  */
/* by agcExecuteBlock (:695-705): n=1 execute, then mute unless the     */
/* squelch status is SIGNALHI (3).                                      */
/* ------------------------------------------------------------------ */
enum { SQ_UNKNOWN = 0, SQ_ENABLED, SQ_RISE, SQ_SIGNALHI, SQ_FALL, SQ_SIGNALLO, SQ_TIMEOUT, SQ_DISABLED };

typedef struct {
    float g, scale, bandwidth, alpha, y2_prime;
    int is_locked, squelch_mode;
    float squelch_threshold;
    unsigned squelch_timeout, squelch_timer;
} orc_agc;

orc_agc *orc_agc_create_ref(float threshold_db)
{
    orc_agc *q = (orc_agc *)calloc(1, sizeof(*q));
    /* agc_crcf_create defaults */
    q->g = 1.0f; q->y2_prime = 1.0f; q->is_locked = 0; q->scale = 1.0f;
    q->squelch_mode = SQ_DISABLED; q->squelch_threshold = 0.0f; q->squelch_timeout = 100;
    q->squelch_timer = q->squelch_timeout;
    /* agcCreate */
    q->bandwidth = 0.1f; q->alpha = q->bandwidth;        /* set_bandwidth 0.1   */
    q->g = 1.0f / 1e-3f; q->y2_prime = 1.0f;            /* set_signal_level    */
    q->squelch_mode = SQ_ENABLED;                        /* squelch_enable      */
    q->squelch_threshold = threshold_db;                 /* set_threshold       */
    q->squelch_timeout = 1000;                           /* set_timeout 1000    */
    return q;
}
void orc_agc_destroy(orc_agc *q) { free(q); }
void orc_agc_get_state(const orc_agc *q, float *g, float *y2, int *mode, unsigned *timer)
{ *g = q->g; *y2 = q->y2_prime; *mode = q->squelch_mode; *timer = q->squelch_timer; }

static inline float orc_agc_rssi(const orc_agc *q) { return -20 * log10(q->g); }

static void orc_agc_squelch_update(orc_agc *q)
{
    int exceeded = (orc_agc_rssi(q) > q->squelch_threshold);
    switch (q->squelch_mode) {
    case SQ_ENABLED:  q->squelch_mode = exceeded ? SQ_RISE : SQ_ENABLED; break;
    case SQ_RISE:     q->squelch_mode = exceeded ? SQ_SIGNALHI : SQ_FALL; break;
    case SQ_SIGNALHI: q->squelch_mode = exceeded ? SQ_SIGNALHI : SQ_FALL; break;
    case SQ_FALL:
        q->squelch_mode = exceeded ? SQ_SIGNALHI : SQ_SIGNALLO;
        q->squelch_timer = q->squelch_timeout;
        break;
    case SQ_SIGNALLO:
        q->squelch_timer--;
        if (q->squelch_timer == 0) q->squelch_mode = SQ_TIMEOUT;
        else if (exceeded) q->squelch_mode = SQ_SIGNALHI;
        break;
    case SQ_TIMEOUT:  q->squelch_mode = SQ_ENABLED; break;
    default: break;
    }
}

/* agc_crcf_execute for one sample */
static inline void orc_agc_execute(orc_agc *q, cf32 x, cf32 *y)
{
    y->re = x.re * q->g; y->im = x.im * q->g;
    float y2 = y->re * y->re + y->im * y->im;             /* crealf(y*conjf(y)) */
    q->y2_prime = (1.0 - q->alpha) * q->y2_prime + q->alpha * y2;
    if (q->is_locked) return;
    if (q->y2_prime > 1e-6f)
        q->g *= expf(-0.5f * q->alpha * logf(q->y2_prime));
    if (q->g > 1e6f) q->g = 1e6f;
    orc_agc_squelch_update(q);
    y->re *= q->scale; y->im *= q->scale;
}

/* agcExecuteBlock (Liquid.chs:695-705) */
void orc_agc_execute_block_ref(orc_agc *q, const cf32 *x, unsigned n, cf32 *y)
{
    for (unsigned i = 0; i < n; i++) {
        orc_agc_execute(q, x[i], &y[i]);
        if (q->squelch_mode != SQ_SIGNALHI) { y[i].re = 0.0f; y[i].im = 0.0f; }
    }
}

/* ------------------------------------------------------------------ */
/* freqdem (freqdem.c); reference fmDemodulator Liquid.chs:303-334.     */
/* ------------------------------------------------------------------ */
typedef struct { float kf, ref; cf32 r_prime; } orc_freqdem;

orc_freqdem *orc_freqdem_create(float kf)
{
    orc_freqdem *q = (orc_freqdem *)calloc(1, sizeof(*q));
    q->kf = kf;
    q->ref = 1.0f / (2 * M_PI * q->kf);
    return q;
}
void orc_freqdem_destroy(orc_freqdem *q) { free(q); }
float orc_freqdem_ref(const orc_freqdem *q) { return q->ref; }

void orc_freqdem_demodulate_block(orc_freqdem *q, const cf32 *r, unsigned n, float *m)
{
    for (unsigned i = 0; i < n; i++) {
        /* cargf(conjf(r_prime) * r) * ref */
        float a = q->r_prime.re, b = -q->r_prime.im, c = r[i].re, d = r[i].im;
        float re = a * c - b * d;
        float im = a * d + b * c;
        m[i] = atan2f(im, re) * q->ref;
        q->r_prime = r[i];
    }
}

/* ------------------------------------------------------------------ */
/* msresamp_crcf(r, As); reference resampler Liquid.chs:56-117           */
/* (resampler (bw/fs) 60, SoapySDR.hs:190-194).                          */
/* STRUCTURE as in liquid-dsp 1.3.2's msresamp.c for decimation: as many  */
/* half-band decimators as doublings bring the rate into [0.5, 1), then   */
/* one arbitrary-rate polyphase resampler (resamp_crcf: bank of npfb      */
/* Kaiser filters, linear interpolation between adjacent phases).        */
/* PARAMETERS: liquid's internal choices (per-stage half-band lengths,    */
/* resamp m / fc / npfb, float phase accumulator) are NOT reliably        */
/* recalled and nothing in the reference prints them -> UNPINNED.  This   */
/* file fixes them as follows ("csdr msresamp v1"):        */
/*   K      = number of doublings: while (rho < 0.5) { K++; rho *= 2 }    */
/*   stage s (input rate fs/2^s): half-band Kaiser h[i] = 0.5 sinc(t/2)   */
/*            w_kaiser(i), t = i - 2 m_s, 4 m_s + 1 taps, As;              */
/*            m_s from Kaiser's length estimate for the transition        */
/*            [fb, 0.5 - fb], fb = 0.45 r 2^s (the final band seen at     */
/*            this stage's rate): N = (As - 7.95) / (14.36 (0.5 - 2 fb)), */
/*            m_s = max(2, ceil((N - 1) / 4));  y[j] = sum h[i] x[2j+1-i] */
/*   arbitrary stage: npfb = 256, m = 7 (14 taps per phase), prototype    */
/*            liquid_firdes_kaiser(2 m npfb + 1, fc / npfb, As),           */
/*            fc = min(0.515 rho, 0.49), scaled by 2 fc (unity DC gain);   */
/*            output k sits at input time t_k = k / rho kept EXACTLY as    */
/*            Q32.32 (delta = round(2^32 / rho)): n = floor(t), b = top 8  */
/*            bits of the fraction, mu = next 24 bits / 2^24,             */
/*            y = (1-mu) F_b(n) + mu F_{b+1}(n), F_b(n) = sum_j            */
/*            pfb[b][j] z[n-j], F_npfb(n) = F_0(n+1); produced once        */
/*            z[n+1] exists.  Integer time makes the output chunk-        */
/*            invariant (liquid's float accumulator is not).              */
/* Interpolation (r >= 1 up to 2) uses the arbitrary stage alone.         */
/* ------------------------------------------------------------------ */
#define ORC_RS_MAXST 24
typedef struct {
    float rate, As; unsigned K; double rho;
    unsigned m_hb[ORC_RS_MAXST]; float *h_hb[ORC_RS_MAXST];
    cf32 *hist_hb[ORC_RS_MAXST];            /* last 4m+1 inputs of the stage */
    uint64_t n_hb[ORC_RS_MAXST];            /* inputs received so far        */
    unsigned npfb, m_arb; float fc; float *pfb;   /* [npfb][2m]               */
    cf32 *hist_a; uint64_t n_a;             /* last 2m+1 inputs, inputs so far */
    uint64_t delta, t_next;                 /* Q32.32; t_next absolute time of the next output */
} orc_msresamp;

static void orc_firdes_kaiser(unsigned N, double fc, double As, double *h)
{
    double beta = orc_kaiser_beta(As), ib = orc_besseli0(beta);
    for (unsigned i = 0; i < N; i++) {
        double t = (double)i - (double)(N - 1) / 2.0;
        double r = 2.0 * t / (double)(N - 1), a = 1.0 - r * r;
        if (a < 0.0) a = 0.0;
        h[i] = orc_sinc(2.0 * fc * t) * orc_besseli0(beta * sqrt(a)) / ib;
    }
}

unsigned orc_msresamp_halfband_m(float rate, float As, unsigned s)
{
    double fb = 0.45 * (double)rate * (double)(1u << s);
    double ft = 0.5 - 2.0 * fb;
    if (ft < 0.01) ft = 0.01;
    double N = (fabs((double)As) - 7.95) / (14.36 * ft);
    int m = (int)ceil((N - 1.0) / 4.0);
    return (unsigned)(m < 2 ? 2 : m);
}

orc_msresamp *orc_msresamp_create(float rate, float As)
{
    if (!(rate > 0.0f) || rate > 2.0f) return NULL;
    orc_msresamp *q = (orc_msresamp *)calloc(1, sizeof(*q));
    q->rate = rate; q->As = As; q->rho = (double)rate;
    while (q->rho < 0.5 && q->K < ORC_RS_MAXST) { q->K++; q->rho *= 2.0; }
    for (unsigned s = 0; s < q->K; s++) {
        unsigned m = orc_msresamp_halfband_m(rate, As, s), N = 4 * m + 1;
        double *hd = (double *)malloc(sizeof(double) * N);
        orc_firdes_kaiser(N, 0.25, As, hd);
        q->m_hb[s] = m; q->h_hb[s] = (float *)malloc(sizeof(float) * N);
        for (unsigned i = 0; i < N; i++) q->h_hb[s][i] = (float)(0.5 * hd[i]);
        free(hd);
        q->hist_hb[s] = (cf32 *)calloc(N, sizeof(cf32));
    }
    q->npfb = 256; q->m_arb = 7;
    double fc = 0.515 * q->rho; if (fc > 0.49) fc = 0.49;
    q->fc = (float)fc;
    unsigned P = 2 * q->m_arb, N = P * q->npfb + 1;
    double *hd = (double *)malloc(sizeof(double) * N);
    orc_firdes_kaiser(N, fc / (double)q->npfb, As, hd);
    q->pfb = (float *)malloc(sizeof(float) * q->npfb * P);
    for (unsigned b = 0; b < q->npfb; b++)
        for (unsigned j = 0; j < P; j++) q->pfb[b * P + j] = (float)(2.0 * fc * hd[b + j * q->npfb]);
    free(hd);
    q->hist_a = (cf32 *)calloc(P + 1, sizeof(cf32));
    q->delta = (uint64_t)llround(4294967296.0 / q->rho);
    q->t_next = 0;
    return q;
}
void orc_msresamp_destroy(orc_msresamp *q)
{
    if (!q) return;
    for (unsigned s = 0; s < q->K; s++) { free(q->h_hb[s]); free(q->hist_hb[s]); }
    free(q->pfb); free(q->hist_a); free(q);
}
float orc_msresamp_get_rate(const orc_msresamp *q) { return q->rate; }
unsigned orc_msresamp_num_halfband(const orc_msresamp *q) { return q->K; }
unsigned orc_msresamp_halfband_len(const orc_msresamp *q, unsigned s) { return 4 * q->m_hb[s] + 1; }
void orc_msresamp_get_pfb(const orc_msresamp *q, float *out) { memcpy(out, q->pfb, sizeof(float) * q->npfb * 2 * q->m_arb); }

/* one stage over a block: hist = the H samples in front of x (oldest first), then shifted */
static unsigned orc_hb_block(orc_msresamp *q, unsigned s, const cf32 *x, unsigned n, cf32 *y)
{
    unsigned m = q->m_hb[s], H = 4 * m + 1;
    cf32 *w = (cf32 *)malloc(sizeof(cf32) * (H + n));
    memcpy(w, q->hist_hb[s], sizeof(cf32) * H);
    memcpy(w + H, x, sizeof(cf32) * n);
    uint64_t N0 = q->n_hb[s], N1 = N0 + n;
    unsigned ny = 0;
    for (uint64_t j = N0 / 2; j < N1 / 2; j++) {
        /* absolute input index a = 2j+1-i lives at w[a - (N0 - H)] */
        float re = 0.0f, im = 0.0f;
        int64_t base = (int64_t)(2 * j + 1) - ((int64_t)N0 - (int64_t)H);
        for (unsigned i = 0; i < H - 0 && i <= 4 * m; i++) {
            cf32 v = w[base - (int64_t)i];
            re += q->h_hb[s][i] * v.re; im += q->h_hb[s][i] * v.im;
        }
        y[ny].re = re; y[ny].im = im; ny++;
    }
    memcpy(q->hist_hb[s], w + n, sizeof(cf32) * H);
    q->n_hb[s] = N1;
    free(w);
    return ny;
}

/* x: n input samples; y must hold orc_msresamp_max_out(n) samples; returns the number written */
unsigned orc_msresamp_max_out(const orc_msresamp *q, unsigned n) { return (unsigned)ceil((double)q->rate * n) + 2; }
unsigned orc_msresamp_execute(orc_msresamp *q, const cf32 *x, unsigned n, cf32 *y)
{
    cf32 *a = (cf32 *)malloc(sizeof(cf32) * (n + 1)), *b = (cf32 *)malloc(sizeof(cf32) * (n + 1));
    const cf32 *cur = x; unsigned cn = n;
    for (unsigned s = 0; s < q->K; s++) {
        cf32 *dst = (cur == a) ? b : a;
        cn = orc_hb_block(q, s, cur, cn, dst);
        cur = dst;
    }
    unsigned P = 2 * q->m_arb, H = P + 1;
    cf32 *w = (cf32 *)malloc(sizeof(cf32) * (H + cn));
    memcpy(w, q->hist_a, sizeof(cf32) * H);
    memcpy(w + H, cur, sizeof(cf32) * cn);
    uint64_t N0 = q->n_a, N1 = N0 + cn;
    unsigned ny = 0;
    while (N1 >= 2 && (q->t_next >> 32) + 1 <= N1 - 1) {
        uint64_t nk = q->t_next >> 32; uint32_t frac = (uint32_t)q->t_next;
        unsigned bidx = frac >> 24; float mu = (float)(frac & 0xffffffu) * (1.0f / 16777216.0f);
        const float *f0 = q->pfb + bidx * P;
        const float *f1 = (bidx + 1 < q->npfb) ? q->pfb + (bidx + 1) * P : q->pfb;
        int64_t base0 = (int64_t)nk - ((int64_t)N0 - (int64_t)H);
        int64_t base1 = base0 + ((bidx + 1 < q->npfb) ? 0 : 1);
        float r0 = 0, i0 = 0, r1 = 0, i1 = 0;
        for (unsigned j = 0; j < P; j++) {
            cf32 v0 = w[base0 - (int64_t)j], v1 = w[base1 - (int64_t)j];
            r0 += f0[j] * v0.re; i0 += f0[j] * v0.im;
            r1 += f1[j] * v1.re; i1 += f1[j] * v1.im;
        }
        y[ny].re = (1.0f - mu) * r0 + mu * r1;
        y[ny].im = (1.0f - mu) * i0 + mu * i1;
        ny++;
        q->t_next += q->delta;
    }
    memcpy(q->hist_a, w + cn, sizeof(cf32) * H);
    q->n_a = N1;
    free(w); free(a); free(b);
    return ny;
}

/* ------------------------------------------------------------------ */
/* ampmodem (ampmodem.c); reference amDemodulator Liquid.chs:439-469:    */
/* ampmodem_create(0.8, 0 = LIQUID_AMPMODEM_DSB, 0 = carrier present).   */
/* RECALLED, UNPINNED: liquid-dsp 1.3.2's DSB / non-suppressed-carrier   */
/* branch is the non-coherent peak detector                             */
/*     t = cabsf(y);  q_hat = alpha*t + (1-alpha)*q_hat;  x = 2*(t-q_hat)*/
/* with ssb_alpha = 0.01 and q_hat = 0 after create/reset.  The          */
/* modulation index (0.8) is not used on this branch.  Nothing in the    */
/* reference (no test, no printed state) confirms the constants.         */
/* ------------------------------------------------------------------ */
typedef struct { float mod_index, alpha, q_hat; } orc_ampdem;

orc_ampdem *orc_ampdem_create(float mod_index)
{
    orc_ampdem *q = (orc_ampdem *)calloc(1, sizeof(*q));
    q->mod_index = mod_index; q->alpha = 0.01f; q->q_hat = 0.0f;
    return q;
}
void orc_ampdem_destroy(orc_ampdem *q) { free(q); }

void orc_ampdem_demodulate_block(orc_ampdem *q, const cf32 *y, unsigned n, float *x)
{
    for (unsigned i = 0; i < n; i++) {
        float t = hypotf(y[i].re, y[i].im);          /* cabsf */
        q->q_hat = q->alpha * t + (1.0f - q->alpha) * q->q_hat;
        x[i] = 2.0f * (t - q->q_hat);
    }
}

/* ------------------------------------------------------------------ */
/* WBFM tail: wbFMDemodulator quadRate decim =                           */
/*     firDecimator decim . iirDeemph . fmDemodulator 0.6                */
/* (Liquid.chs:653-656), iirDeemph = iirFilter 2 (5000/quadRate) 0 10 10 */
/*   = iirfilt_rrrf_create_prototype(BUTTER, LOWPASS, SOS, 2, fc, ...)   */
/*   (Liquid.chs:615-622), firDecimator m =                              */
/*   firdecim_rrrf_create_kaiser(m, 10, 60) (Liquid.chs:485-490).        */
/* RECALLED, UNPINNED (liquid-dsp 1.3.2 iirdes.c / iirfiltsos.c /        */
/* firdecim.c): 2nd-order Butterworth low-pass through the bilinear      */
/* transform with pre-warping (K = tan(pi fc)), unit DC gain, run as one  */
/* direct-form-II section; decimator prototype                           */
/* liquid_firdes_kaiser(2 M m + 1, 0.5/M, As), scale 1 (DC gain ~ M), one */
/* output when the FIRST sample of each block of M has been pushed:       */
/* y[j] = sum_i h[i] x[jM - i].                                           */
/* ------------------------------------------------------------------ */
typedef struct { float b[3], a[3], v1, v2; } orc_biquad;

orc_biquad *orc_butter2_lowpass_create(float fc)
{
    orc_biquad *q = (orc_biquad *)calloc(1, sizeof(*q));
    double K = tan(M_PI * (double)fc), n = 1.0 / (1.0 + M_SQRT2 * K + K * K);
    q->b[0] = (float)(K * K * n); q->b[1] = (float)(2.0 * K * K * n); q->b[2] = q->b[0];
    q->a[0] = 1.0f; q->a[1] = (float)(2.0 * (K * K - 1.0) * n); q->a[2] = (float)((1.0 - M_SQRT2 * K + K * K) * n);
    return q;
}
void orc_biquad_destroy(orc_biquad *q) { free(q); }
void orc_biquad_coeffs(const orc_biquad *q, float *ba) { memcpy(ba, q->b, 12); memcpy(ba + 3, q->a, 12); }
void orc_biquad_execute_block(orc_biquad *q, const float *x, unsigned n, float *y)
{
    for (unsigned i = 0; i < n; i++) {
        /* iirfiltsos execute_df2 */
        float v0 = x[i] - q->a[1] * q->v1 - q->a[2] * q->v2;
        y[i] = q->b[0] * v0 + q->b[1] * q->v1 + q->b[2] * q->v2;
        q->v2 = q->v1; q->v1 = v0;
    }
}

typedef struct { unsigned M, h_len; float *h, *hist; uint64_t n_seen; } orc_firdecim;

orc_firdecim *orc_firdecim_create_kaiser(unsigned M, unsigned m, float As)
{
    orc_firdecim *q = (orc_firdecim *)calloc(1, sizeof(*q));
    q->M = M; q->h_len = 2 * M * m + 1;
    double *hd = (double *)malloc(sizeof(double) * q->h_len);
    orc_firdes_kaiser(q->h_len, 0.5 / (double)M, As, hd);
    q->h = (float *)malloc(sizeof(float) * q->h_len);
    for (unsigned i = 0; i < q->h_len; i++) q->h[i] = (float)hd[i];
    free(hd);
    q->hist = (float *)calloc(q->h_len, sizeof(float));
    return q;
}
void orc_firdecim_destroy(orc_firdecim *q) { if (q) { free(q->h); free(q->hist); free(q); } }
unsigned orc_firdecim_len(const orc_firdecim *q) { return q->h_len; }
void orc_firdecim_taps(const orc_firdecim *q, float *h) { memcpy(h, q->h, sizeof(float) * q->h_len); }
/* x: n samples, n % M == 0 (the reference's `div`, Liquid.chs:495-497); y: n / M samples */
void orc_firdecim_execute_block(orc_firdecim *q, const float *x, unsigned n, float *y)
{
    unsigned H = q->h_len - 1;
    float *w = (float *)malloc(sizeof(float) * (H + n));
    memcpy(w, q->hist, sizeof(float) * H);
    memcpy(w + H, x, sizeof(float) * n);
    for (unsigned j = 0; j < n / q->M; j++) {
        float acc = 0.0f;
        const float *p = w + H + (size_t)j * q->M;
        for (unsigned i = 0; i < q->h_len; i++) acc += q->h[i] * *(p - i);
        y[j] = acc;
    }
    memcpy(q->hist, w + n, sizeof(float) * H);
    free(w);
}

/* ------------------------------------------------------------------ */
/* mix (Trans.hs:119-122): strict left fold of element-wise +.          */
/* ------------------------------------------------------------------ */
void orc_mix_f32(const float *chans, unsigned M, unsigned n, float *out)
{
    for (unsigned i = 0; i < n; i++) {
        float acc = chans[i];
        for (unsigned k = 1; k < M; k++) acc = acc + chans[(size_t)k * n + i];
        out[i] = acc;
    }
}

/* ------------------------------------------------------------------ */
/* The composition of assembleFold (SoapySDR.hs:208-226) for one        */
/* compacted chunk stream: dcBlocker -> [PFB] -> per-channel            */
/* (agc?) -> (fm?) -> (mix?).  Output channel-major [M][nf] (or [nf]    */
/* when mixed); element = cf32 (demod none) or f32 (FM).                */
/* ------------------------------------------------------------------ */
typedef struct {
    unsigned M;
    int dc_block, agc_enable, demod, mix;
    orc_dcblock *dc;
    orc_chan *chan;
    orc_agc **agc;
    orc_freqdem **fm;
    orc_ampdem **am;
    orc_biquad **de; orc_firdecim **dec; unsigned decim;
} orc_chain;

orc_chain *orc_chain_create_wbfm(unsigned M, int dc_block, int agc_enable, float agc_thr_db, float deemph_fc, unsigned decim, int mix);
orc_chain *orc_chain_create(unsigned M, int dc_block, int agc_enable, float agc_thr_db,
                            int demod, float kf, int mix)
{
    orc_chain *q = (orc_chain *)calloc(1, sizeof(*q));
    q->M = M; q->dc_block = dc_block; q->agc_enable = agc_enable; q->demod = demod; q->mix = mix;
    if (dc_block) q->dc = orc_dcblock_create(0.0005f);   /* Liquid.chs:577 */
    if (M > 1) q->chan = orc_chan_create(M);
    if (agc_enable) {
        q->agc = (orc_agc **)calloc(M, sizeof(orc_agc *));
        for (unsigned k = 0; k < M; k++) q->agc[k] = orc_agc_create_ref(agc_thr_db);
    }
    if (demod == 1) {
        q->fm = (orc_freqdem **)calloc(M, sizeof(orc_freqdem *));
        for (unsigned k = 0; k < M; k++) q->fm[k] = orc_freqdem_create(kf);
    }
    if (demod == 2) {                                    /* DeAM: amDemodulator . agc (SoapySDR.hs:265-272) */
        q->am = (orc_ampdem **)calloc(M, sizeof(orc_ampdem *));
        for (unsigned k = 0; k < M; k++) q->am[k] = orc_ampdem_create(0.8f);   /* Liquid.chs:455 */
    }
    return q;
}
/* DeWBFM decim: wbFMDemodulator outBW decim . agc (SoapySDR.hs:252-259); deemph_fc = 5000 / outBW */
orc_chain *orc_chain_create_wbfm(unsigned M, int dc_block, int agc_enable, float agc_thr_db, float deemph_fc, unsigned decim, int mix)
{
    orc_chain *q = orc_chain_create(M, dc_block, agc_enable, agc_thr_db, 1, 0.6f, mix);
    q->demod = 3; q->decim = decim;
    q->de = (orc_biquad **)calloc(M, sizeof(orc_biquad *));
    q->dec = (orc_firdecim **)calloc(M, sizeof(orc_firdecim *));
    for (unsigned k = 0; k < M; k++) { q->de[k] = orc_butter2_lowpass_create(deemph_fc); q->dec[k] = orc_firdecim_create_kaiser(decim, 10, 60.0f); }
    return q;
}
void orc_chain_set_dft_backward(orc_chain *q, int backward) { if (q->chan) orc_chan_set_dft_backward(q->chan, backward); }
void orc_chain_destroy(orc_chain *q)
{
    if (!q) return;
    if (q->de) { for (unsigned k = 0; k < q->M; k++) { orc_biquad_destroy(q->de[k]); orc_firdecim_destroy(q->dec[k]); } free(q->de); free(q->dec); }
    if (q->dc) orc_dcblock_destroy(q->dc);
    if (q->chan) orc_chan_destroy(q->chan);
    if (q->agc) { for (unsigned k = 0; k < q->M; k++) orc_agc_destroy(q->agc[k]); free(q->agc); }
    if (q->fm) { for (unsigned k = 0; k < q->M; k++) orc_freqdem_destroy(q->fm[k]); free(q->fm); }
    if (q->am) { for (unsigned k = 0; k < q->M; k++) orc_ampdem_destroy(q->am[k]); free(q->am); }
    free(q);
}

/* nx must be a multiple of M.  out sized M*nf elements (or nf if mix). */
void orc_chain_process(orc_chain *q, const cf32 *x, unsigned nx, void *out)
{
    unsigned M = q->M, nf = nx / M;
    size_t tot = (size_t)M * nf;
    cf32 *a = (cf32 *)malloc(sizeof(cf32) * (tot ? tot : 1));
    cf32 *b = (cf32 *)malloc(sizeof(cf32) * (tot ? tot : 1));
    const cf32 *cur = x;
    if (q->dc) { orc_dcblock_execute(q->dc, cur, nx, a); cur = a; }
    if (q->chan) { orc_chan_process(q->chan, cur, nx, b); cur = b; }
    /* cur is channel-major [M][nf] now (M == 1: the stream itself) */
    if (q->agc) {
        cf32 *t = (cur == a) ? b : a;
        for (unsigned k = 0; k < M; k++)
            orc_agc_execute_block_ref(q->agc[k], cur + (size_t)k * nf, nf, t + (size_t)k * nf);
        cur = t;
    }
    if (q->demod == 3) {
        /* out: [M][nf / decim] (or [nf / decim] mixed) */
        unsigned no = nf / q->decim;
        float *f = (float *)malloc(sizeof(float) * (tot ? tot : 1)), *g = (float *)malloc(sizeof(float) * (tot ? tot : 1));
        float *o = (float *)malloc(sizeof(float) * ((size_t)M * no + 1));
        for (unsigned k = 0; k < M; k++) {
            orc_freqdem_demodulate_block(q->fm[k], cur + (size_t)k * nf, nf, f + (size_t)k * nf);
            orc_biquad_execute_block(q->de[k], f + (size_t)k * nf, nf, g + (size_t)k * nf);
            orc_firdecim_execute_block(q->dec[k], g + (size_t)k * nf, nf, o + (size_t)k * no);
        }
        if (q->mix && M > 1) orc_mix_f32(o, M, no, (float *)out);
        else memcpy(out, o, sizeof(float) * (size_t)M * no);
        free(f); free(g); free(o);
    } else if (q->demod == 1 || q->demod == 2) {
        float *f = (float *)malloc(sizeof(float) * (tot ? tot : 1));
        for (unsigned k = 0; k < M; k++) {
            if (q->demod == 1) orc_freqdem_demodulate_block(q->fm[k], cur + (size_t)k * nf, nf, f + (size_t)k * nf);
            else orc_ampdem_demodulate_block(q->am[k], cur + (size_t)k * nf, nf, f + (size_t)k * nf);
        }
        if (q->mix && M > 1) orc_mix_f32(f, M, nf, (float *)out);
        else memcpy(out, f, sizeof(float) * tot);
        free(f);
    } else {
        if (q->mix && M > 1) {
            /* complex + is componentwise: fold re and im separately */
            cf32 *o = (cf32 *)out;
            for (unsigned i = 0; i < nf; i++) {
                cf32 acc = cur[i];
                for (unsigned k = 1; k < M; k++) {
                    acc.re = acc.re + cur[(size_t)k * nf + i].re;
                    acc.im = acc.im + cur[(size_t)k * nf + i].im;
                }
                o[i] = acc;
            }
        } else memcpy(out, cur, sizeof(cf32) * tot);
    }
    free(a); free(b);
}
