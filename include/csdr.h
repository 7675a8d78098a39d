/*
 * csdr.h -- C ABI of the MI355X-native DSP chain that replaces the liquid-dsp
 * FFI calls behind ComposableSDR's Pipe blocks.
 *
 * Every block of the reference is
 *     Pipe { _start :: IO r, _process :: r -> Array a -> IO (Array b), _done :: r -> IO () }
 * (/root/reference/src/ComposableSDR/Types.hs:51-55) whose three fields wrap a
 * liquid-dsp  *_create / *_execute_block / *_destroy  triple.  Each object
 * below exports the same triple at the same (chunk) granularity:
 *     int csdr_X_create (..., csdr_X **out);      <- _start
 *     int csdr_X_process(csdr_X *, in, n, out);   <- _process (one whole chunk)
 *     int csdr_X_destroy(csdr_X *);               <- _done
 * Conventions:
 *   - return 0 on success, <0 on error (so the Haskell side can reuse
 *     Common.hs:32-33 `try`); never exit(); csdr_last_error() gives the text.
 *     liquid-dsp 1.3.2 aborts the process on a bad configuration instead.
 *   - in/out buffers are caller-owned and only touched during the call; CF32 is
 *     interleaved little-endian float32 (re, im) = Types.hs:82-88.
 *   - handles are opaque, single-threaded, independent of each other; all
 *     stream state (DC-blocker v1, NCO phase, filterbank windows, per-channel
 *     AGC / freqdem state) lives in the handle, so results do not depend on how
 *     the stream is chunked.
 *   - *_process takes host pointers and blocks; *_process_device takes device
 *     pointers (HBM-resident data) and enqueues on a hipStream_t without
 *     synchronising.
 * All compute runs in hand-written HIP kernels for gfx950.  There is no CPU
 * fallback: with no usable GPU every create returns CSDR_ERR_NODEV.
 */
#ifndef CSDR_H
#define CSDR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* libcsdr_hip.so is built with -fvisibility=hidden: the declarations between this push and the
 * pop at the end of the file are the library's whole dynamic symbol table (what a Haskell
 * `foreign import ccall` can bind); the C++ internals behind them are not exported. */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define CSDR_OK            0
#define CSDR_ERR_INVALID  (-1)  /* bad argument / configuration                         */
#define CSDR_ERR_HIP      (-2)  /* HIP runtime error (text in csdr_last_error)          */
#define CSDR_ERR_NODEV    (-3)  /* no gfx950 device visible                              */
#define CSDR_ERR_SIZE     (-4)  /* chunk length not a multiple of the channel count, or
                                   larger than the handle was created for                */
#define CSDR_ERR_NOMEM    (-5)
#define CSDR_ERR_BUSY     (-6)  /* csdr_chain_submit: CSDR_CHAIN_INFLIGHT chunks already pending */

#define CSDR_DEMOD_NONE 0u      /* DeNo: per-channel CF32 out (SoapySDR.hs:236-243)      */
#define CSDR_DEMOD_FM   1u      /* DeNBFM kf: freqdem, F32 out (SoapySDR.hs:244-251)     */
#define CSDR_DEMOD_WBFM 3u      /* DeWBFM decim: firDecimator decim . iirDeemph . fmDemodulator 0.6,
                                   F32 out, n_in/channels/decim samples per channel
                                   (SoapySDR.hs:252-259, Liquid.chs:653-656)              */
#define CSDR_DEMOD_AM   2u      /* DeAM: ampmodem DSB peak detector, F32 out
                                   (SoapySDR.hs:265-272, Liquid.chs:439-469)              */

/* cfg.flags */
#define CSDR_FLAG_TIME_KERNELS 1u   /* bracket the dominant kernel with hipEvents         */
#define CSDR_FLAG_FORCE_GENERIC 2u  /* use the any-M multi-kernel path even where a fused
                                       kernel exists (for A/B tests)                      */
#define CSDR_FLAG_QUIET 4u          /* do not print the configuration at create           */
#define CSDR_FLAG_AGC_SEQUENTIAL 8u /* run the per-channel AGC as one lane per channel instead of
                                       the time-parallel verified tail; both give bit-identical
                                       output (for A/B tests)                                */
#define CSDR_FLAG_NO_MIX_IDENTITY 16u /* DeNo --mix over all channels of the any-M route: sum_k Y[k] = M * X[0] (the sum of all
                                        * DFT bins of a frame is M times its first input), so the product path computes the
                                        * DC blocker on the whole stream and the FIR of polyphase branch 0 only.  Set this flag
                                        * to run the full bank + DFT + channel sum instead (same result to f32 rounding).      */

#define CSDR_FLAG_TIME_REGION 32u   /* time the dominant kernel over a REGION instead of per launch: one hipEvent in front of the
                                     * first timed launch after the last csdr_chain_kernel_time() and one behind the last, recorded
                                     * when the total is read.  total / launches is then the launch cadence of back-to-back calls
                                     * (kernel + the gap to the next launch); it includes whatever else the call puts on the stream
                                     * between two timed launches, so it is a per-kernel figure only for one-kernel steps.  Per-launch
                                     * event pairs (CSDR_FLAG_TIME_KERNELS alone) cost the stream 10-20 us per launch.           */

#define CSDR_FLAG_TAIL_ONLY 64u     /* the handle is the per-channel TAIL of the chain alone: `automaticGainControl` (with the mute
                                     * rule) [-> `fmDemodulator kf`] [-> `mix`] on `channels` independent rows (Trans.hs:124-129 `mux`,
                                     * SoapySDR.hs:249).  process(): in = a channel-major CF32 plane [channels][nf] (n_in = channels *
                                     * nf samples: what a DeNo chain writes), out = [channels][nf] CF32 / F32 ([nf] with mix).  For the
                                     * hybrid multi-GPU partition: DC blocker, pre-mix, FIR and DFT run time-sharded, one
                                     * all-to-all turns time stripes into channel shards, every rank runs this tail on its channels for
                                     * the whole time span.  Needs agc_threshold_db != 0; dc_block, chan_*, pfb_* are ignored.         */

#define CSDR_FLAG_DFT_BACKWARD 128u /* The direction of firpfbch_crcf_analyzer_execute's transform (Liquid.chs:843) is recalled, not pinned by
                                     * anything in the reference (hard part 1): the library computes the FORWARD DFT
                                     * out[k] = sum_j X[j] e^{-j 2 pi jk/M}.  With this flag the handle delivers the other possible
                                     * convention, out[k] = sum_j X[j] e^{+j 2 pi jk/M} = forward bin (M - k) mod M: output row k (file
                                     * _ch<k+1>) is the forward chain's row (M - k) mod M, every per-channel tail (AGC, demod) unchanged
                                     * on its row.  One extra pass over the OUTPUT (a row permutation behind the fused kernels); whole-band
                                     * handles only (channel shards: CSDR_ERR_INVALID); with mix the sum over all channels is the same set
                                     * of terms, so only the fold order differs (no-op).  Should a diff against a real liquid-dsp 1.3.2
                                     * show the backward convention, this flag becomes the default and nothing else changes.        */

const char *csdr_last_error(void);
int  csdr_device_count(void);
/* library / build identification: "csdr-hip gfx950 <version>" */
const char *csdr_version(void);

/* ------------------------------------------------------------------------ *
 * dcBlocker  (Liquid.chs:575-589)
 *   replaces iirfilt_crcf_create_dc_blocker / _execute_block / _destroy
 *   (imports at Liquid.chs:550-567).  y = DC-blocked x, same length.
 * ------------------------------------------------------------------------ */
typedef struct csdr_dcblock csdr_dcblock;
int csdr_dcblock_create(float alpha, uint32_t max_samples, csdr_dcblock **out);
int csdr_dcblock_process(csdr_dcblock *h, const float *x_cf32, uint32_t n, float *y_cf32);
int csdr_dcblock_process_device(csdr_dcblock *h, const void *d_x, uint32_t n, void *d_y, void *stream);
int csdr_dcblock_destroy(csdr_dcblock *h);

/* ------------------------------------------------------------------------ *
 * mixDown / mixUp  (Liquid.chs:782-809)
 *   replaces nco_crcf_create(LIQUID_VCO) + set_frequency + mix_block_down/up
 *   + destroy (imports at Liquid.chs:746-780).  freq in rad/sample.
 * ------------------------------------------------------------------------ */
typedef struct csdr_nco csdr_nco;
int csdr_nco_create(float freq, uint32_t max_samples, csdr_nco **out);
int csdr_nco_mix_down(csdr_nco *h, const float *x_cf32, uint32_t n, float *y_cf32);
int csdr_nco_mix_up(csdr_nco *h, const float *x_cf32, uint32_t n, float *y_cf32);
int csdr_nco_get_words(const csdr_nco *h, uint32_t *theta, uint32_t *d_theta);
int csdr_nco_destroy(csdr_nco *h);

/* ------------------------------------------------------------------------ *
 * automaticGainControl tres  (Liquid.chs:693-728), `nchan` independent
 * instances as created by mux / distribute_ (Trans.hs:106-129).
 *   replaces agc_crcf_create + set_bandwidth 0.1 + set_signal_level 1e-3 +
 *   squelch_enable + squelch_set_threshold tres + squelch_set_timeout 1000,
 *   and per sample execute_block(n=1) + squelch_get_status + get_rssi with the
 *   reference's mute rule "status /= SIGNALHI => 0" (imports :660-691).
 *   x, y are channel-major [nchan][n] CF32.
 * ------------------------------------------------------------------------ */
typedef struct csdr_agc csdr_agc;
int csdr_agc_create(float threshold_db, uint32_t nchan, uint32_t max_samples, csdr_agc **out);
int csdr_agc_process(csdr_agc *h, const float *x_cf32, uint32_t n, float *y_cf32);
int csdr_agc_destroy(csdr_agc *h);

/* ------------------------------------------------------------------------ *
 * fmDemodulator kf  (Liquid.chs:303-334), `nchan` independent instances.
 *   replaces freqdem_create / freqdem_demodulate_block / freqdem_destroy
 *   (imports :305-315).  x is [nchan][n] CF32, m is [nchan][n] F32.
 * ------------------------------------------------------------------------ */
typedef struct csdr_freqdem csdr_freqdem;
int csdr_freqdem_create(float kf, uint32_t nchan, uint32_t max_samples, csdr_freqdem **out);
int csdr_freqdem_process(csdr_freqdem *h, const float *x_cf32, uint32_t n, float *m_f32);
int csdr_freqdem_destroy(csdr_freqdem *h);

/* ------------------------------------------------------------------------ *
 * iirFilter n fc f0 ap as  (Liquid.chs:629-638) = iirfilt_rrrf_create_prototype(BUTTER, LOWPASS,
 * SOS, n, fc, f0, ap, as) (imports :593-611), `nchan` independent real-valued instances.
 * Only what the reference instantiates is built: order 2 (the WBFM de-emphasis,
 * Liquid.chs:655); other orders return CSDR_ERR_INVALID.  f0 / ap / as do not enter a
 * Butterworth low-pass and are ignored, as in liquid.  x, y are [nchan][n] F32.
 * firDecimator m  (Liquid.chs:485-501) = firdecim_rrrf_create_kaiser(m, 10, 60) (imports
 * :473-483): x is [nchan][n] F32 with n % m == 0 (the reference's `div`), y is [nchan][n/m].
 * Arithmetic recalled from liquid-dsp 1.3.2 (unpinned).
 * ------------------------------------------------------------------------ */
typedef struct csdr_iirfilt csdr_iirfilt;
int csdr_iirfilt_create(uint32_t order, float fc, float f0, float ap, float as_db, uint32_t nchan, uint32_t max_samples,
                        csdr_iirfilt **out);
int csdr_iirfilt_process(csdr_iirfilt *h, const float *x_f32, uint32_t n, float *y_f32);
int csdr_iirfilt_destroy(csdr_iirfilt *h);
typedef struct csdr_firdecim csdr_firdecim;
int csdr_firdecim_create(uint32_t decim, uint32_t nchan, uint32_t max_samples, csdr_firdecim **out);
int csdr_firdecim_process(csdr_firdecim *h, const float *x_f32, uint32_t n, float *y_f32);
int csdr_firdecim_destroy(csdr_firdecim *h);

/* ------------------------------------------------------------------------ *
 * resampler r as  (Liquid.chs:56-117): rate r = bandwidth / samplerate, as = 60 dB
 * (SoapySDR.hs:190-194).
 *   replaces msresamp_crcf_create / _print / _get_rate / _execute / _destroy
 *   (imports :58-73).  Variable-length output like msresamp_crcf_execute's
 *   out-count pointer (:79-98): the caller provides csdr_resamp_max_out(h, n_in)
 *   samples of room (= the reference's 2*ceil(r*nx), :81).
 *   Structure = liquid-dsp's msresamp (half-band decimators + one arbitrary-rate
 *   polyphase stage); liquid's internal filter parameters are not recoverable from
 *   the reference, so they are fixed by this library: unpinned.
 *   rate == 0: pass-through (the reference's nullPtr resampler, :100-103);
 *   rate > 2: CSDR_ERR_INVALID.  CSDR_QUIET in the environment silences the print.
 * ------------------------------------------------------------------------ */
typedef struct csdr_resamp csdr_resamp;
int      csdr_resamp_create(float rate, float as_db, uint32_t max_in, csdr_resamp **out);
float    csdr_resamp_get_rate(const csdr_resamp *h);
uint32_t csdr_resamp_max_out(const csdr_resamp *h, uint32_t n_in);
int      csdr_resamp_process(csdr_resamp *h, const float *x_cf32, uint32_t n_in, float *y_cf32, uint32_t *n_out);
int      csdr_resamp_process_device(csdr_resamp *h, const void *d_x, uint32_t n_in, void *d_y, uint32_t *n_out, void *stream);
int      csdr_resamp_destroy(csdr_resamp *h);

/* ------------------------------------------------------------------------ *
 * amDemodulator  (Liquid.chs:439-469), `nchan` independent instances.
 *   replaces ampmodem_create(0.8, LIQUID_AMPMODEM_DSB, 0) / ampmodem_demodulate_block /
 *   ampmodem_destroy (imports :441-450).  x is [nchan][n] CF32, m is [nchan][n] F32.
 *   Arithmetic = liquid-dsp 1.3.2's non-coherent peak detector as recalled (unpinned):
 *   t = |x|, q <- 0.01 t + 0.99 q, m = 2 (t - q); mod_index is accepted and unused, as there.
 * ------------------------------------------------------------------------ */
typedef struct csdr_ampdem csdr_ampdem;
int csdr_ampdem_create(float mod_index, uint32_t nchan, uint32_t max_samples, csdr_ampdem **out);
int csdr_ampdem_process(csdr_ampdem *h, const float *x_cf32, uint32_t n, float *m_f32);
int csdr_ampdem_destroy(csdr_ampdem *h);

/* ------------------------------------------------------------------------ *
 * The fused chain: everything assembleFold (apps/SoapySDR.hs:208-226) puts
 * behind `compact`:
 *     dcBlocker                                   (SoapySDR.hs:213-214)
 *  -> firpfbchChannelizer M  = NCO pre-mix + firpfbch_crcf analyzer +
 *     transpose to channel-major                  (Liquid.chs:811-866)
 *  -> per channel: [automaticGainControl] -> [fmDemodulator kf]
 *                                                 (SoapySDR.hs:190-199, 249)
 *  -> [mix]                                       (Trans.hs:119-122)
 * replacing  mix . mux (replicate nch demod) . firpfbchChannelizer nc  and
 * firpfbchChannelizer nc + distribute_ (addPipe demod sink)  (SoapySDR.hs:218-225).
 * For channels == 1 it is  demod  alone behind the DC blocker (SoapySDR.hs:226).
 *
 * process(): in = n_in CF32 samples, n_in a multiple of `channels`
 *   (the reference's chunk is 4*channels*1024, SoapySDR.hs:215; the reference
 *   misbehaves for other remainders, SURVEY.md a6 -> CSDR_ERR_SIZE here).
 *   out = ONE contiguous channel-major buffer [chan_count][nf], nf = n_in/channels,
 *   element CF32 (demod none) or F32 (FM); with mix: [nf] only.  The caller
 *   slices it into per-channel arrays exactly like Liquid.chs:850-862.
 *   *n_out = number of output ELEMENTS written.  n_in == 0 is a no-op.
 * ------------------------------------------------------------------------ */
typedef struct csdr_chain csdr_chain;

typedef struct csdr_chain_cfg {
    uint32_t struct_size;       /* = sizeof(csdr_chain_cfg)                              */
    uint32_t channels;          /* -c M, >= 1                                            */
    uint32_t dc_block;          /* 1: include dcBlocker (assembleFold always does)       */
    float    dc_alpha;          /* 0.0005 (Liquid.chs:577)                               */
    float    agc_threshold_db;  /* -a tres; 0 = no AGC (SoapySDR.hs:195-198)             */
    uint32_t demod;             /* CSDR_DEMOD_*                                          */
    float    kf;                /* DeNBFM kf                                             */
    uint32_t mix;               /* --mix                                                 */
    uint32_t chan_first;        /* channel shard [chan_first, chan_first+chan_count)     */
    uint32_t chan_count;        /*   0 = all channels                                    */
    int32_t  device;            /* HIP device ordinal, -1 = current device               */
    uint32_t max_frames;        /* largest n_in/channels per call; 0 = 4096              */
    uint32_t flags;             /* CSDR_FLAG_*                                           */
    uint32_t pfb_m;             /* filter semi-length m, 0 = 7  (Liquid.chs:813)         */
    float    pfb_as;            /* stop-band attenuation, 0 = 80 dB (Liquid.chs:813)     */
    uint32_t wbfm_decim;        /* DeWBFM decim (SoapySDR.hs:252-259); 0 = 4             */
    float    deemph_fc;         /* DeWBFM de-emphasis corner 5000/quadRate (Liquid.chs:655); 0 = 0.025 */
    uint32_t chan_stride;       /* G > 1: interleaved channel ownership for channel-sharded multi-GPU runs
                                 * (SURVEY 8e(A)): this handle produces the channels chan_first, chan_first + G, ...
                                 * (chan_first < G, G divides channels, chan_count 0 or channels/G); output row m is
                                 * channel chan_first + G*m.  channels = 256 or 1024 with G = 2, 4, 8: the fused run kernels'
                                 * shard variants (the shift by chan_first rides on the pre-mix phasors, the DFT passes and
                                 * the freqdem / stores run for the owned channels only); every other shape: the M-point DFT
                                 * of a frame is pruned to one length-G fold plus one (channels/G)-point DFT on the any-M
                                 * route.  Either way a shard does 1/G of the DFT and tail work, while DC blocker, pre-mix
                                 * and FIR still see every branch.  0 / 1 = contiguous shard. */
} csdr_chain_cfg;

void csdr_chain_cfg_default(csdr_chain_cfg *cfg, uint32_t channels);
int  csdr_chain_create(const csdr_chain_cfg *cfg, csdr_chain **out);
int  csdr_chain_process(csdr_chain *h, const float *in_cf32, uint32_t n_in, void *out, uint32_t *n_out);
int  csdr_chain_process_device(csdr_chain *h, const void *d_in_cf32, uint32_t n_in,
                               void *d_out, uint32_t *n_out, void *stream);
/* Asynchronous host-buffer entry point (what a streaming caller such as the reference's fold uses; replaces the blocking
 * firpfbch/analyzer call at Liquid.chs:845 and the caller-owned buffers of Liquid.chs:82, :292): up to
 * CSDR_CHAIN_INFLIGHT chunks are in flight, H2D copy, kernels and D2H copy run on three streams so that the copies of
 * neighbouring chunks overlap the kernels.  `submit` returns as soon as the work is queued (CSDR_ERR_BUSY when
 * CSDR_CHAIN_INFLIGHT chunks are already pending); `collect` waits for the OLDEST submitted chunk, whose result is then
 * in the `out` given to its submit.  Buffers from csdr_host_alloc (page-locked) are copied from / to directly; any
 * other buffer is staged through page-locked memory owned by the handle (one extra host memcpy each way).  The
 * buffers of a chunk must stay valid until its collect.  csdr_chain_process = submit + collect. */
/* Pipelined DEVICE entry point, for callers that keep several chunks in flight in HBM (a capture ring, the bench): like
 * csdr_chain_process_device, but the work goes onto two handle-owned streams used alternately, and the call does not order
 * itself behind the previous chunk where the path allows it.  For the fused 256-channel chain without AGC / AM / WBFM
 * tails and chunks of whole 16-frame tiles at the run-kernel size, consecutive launches are INDEPENDENT: run 0 of a
 * chunk starts cold like every other run of the launch (DC state from a read-only warm-up over the previous chunk's last
 * tiles, which the handle keeps a copy of; FIR window and freqdem history from its last tile), so the next launch's
 * workgroups fill the compute units the previous launch has already left, and its cold start (tens of microseconds of
 * memory traffic with idle ALUs) runs under the previous launch's tile loops.  Same results as csdr_chain_process_device
 * to the run-start tolerance every run of a launch already has (DC state truncated at beta^24576).  Other configurations
 * are accepted and run serialized.
 *   ready_event: hipEvent_t after which d_in is complete, or NULL if it already is when the call is made.
 *   d_in must stay untouched and d_out unread until csdr_chain_wait_device (stream = NULL: the host waits for every
 *   submitted chunk; else `stream` is made to wait).  csdr_chain_process_device orders itself behind submitted chunks.
 * Stream lifetime: the handle never keeps a caller's `stream`; it records an event of its own behind the call's work, so a
 * caller stream may be destroyed once the caller itself has no further use for it (its pending work still completes). */
int  csdr_chain_submit_device(csdr_chain *h, const void *d_in_cf32, uint32_t n_in, void *d_out, uint32_t *n_out, void *ready_event);
int  csdr_chain_wait_device(csdr_chain *h, void *stream);
uint32_t csdr_chain_debug_independent_launches(const csdr_chain *h);   /* submit_device calls since create that ran as independent launches */
#define CSDR_CHAIN_INFLIGHT 3
void *csdr_host_alloc(size_t bytes);            /* page-locked host memory (hipHostMalloc); NULL on failure */
void  csdr_host_free(void *p);
int  csdr_chain_submit(csdr_chain *h, const float *in_cf32, uint32_t n_in, void *out);
int  csdr_chain_collect(csdr_chain *h, uint32_t *n_out);
/* Device-side health of the handle since the last check: CSDR_ERR_HIP (text in csdr_last_error) when an
 * inter-workgroup wait of the small-chunk kernels hit its spin limit (their output is then invalid), CSDR_OK
 * otherwise.  Synchronises the device; csdr_chain_process and csdr_chain_reset call it themselves, callers of the
 * device / async entry points call it at their own sync points. */
int  csdr_chain_status(csdr_chain *h);
int  csdr_chain_reset(csdr_chain *h);      /* back to the state right after create        */
/* Reset, then place the stream position at frame `frames` (n = frames*channels samples in):
 * the NCO pre-mix phase becomes what it would be there.  Used by time-striped multi-GPU runs,
 * where a rank starts in the middle of the stream behind a warm-up prefix. */
int  csdr_chain_seek_frames(csdr_chain *h, uint64_t frames);
int  csdr_chain_destroy(csdr_chain *h);

/* introspection used by the tests (mirrors what firpfbch_crcf_print / nco_crcf_print
 * show at Liquid.chs:814-820) */
uint32_t csdr_chain_out_elem_size(const csdr_chain *h);            /* 8 or 4 bytes        */
int  csdr_chain_get_taps(const csdr_chain *h, float *taps, uint32_t n); /* first M*2m taps */
int  csdr_chain_get_nco(const csdr_chain *h, uint32_t *theta, uint32_t *d_theta);
const char *csdr_chain_path(const csdr_chain *h);                  /* "fused-..." | "generic" */
/* The library's route table as text: which plan and which kernels a configuration (channels x chan_stride x output x AGC) gets,
 * per call shape.  csdr_chain_create selects from exactly this table. */
const char *csdr_route_table(void);
/* CSDR_FLAG_TIME_KERNELS: accumulated duration of the dominant kernel's launches
 * since the last call (synchronises the stream).  Returns the kernel's name. */
const char *csdr_chain_kernel_time(csdr_chain *h, double *total_ms, uint32_t *launches);

/* Diagnostics: with CSDR_TRACE=1 in the environment at create time the fused kernel records 16
 * s_memtime stamps per 16-frame tile; copies the stamps of the first `ntiles` tiles of the last
 * launch into out[ntiles][16] and returns the number of tiles copied (0 when tracing is off). */
int  csdr_chain_debug_trace(csdr_chain *h, unsigned long long *out, uint32_t ntiles);
/* Diagnostics of the time-parallel AGC tail since create: segments whose speculative start state was
 * checked against the true state, and how many of them had to be recomputed sequentially. */
int  csdr_chain_debug_agc(csdr_chain *h, uint32_t *checked, uint32_t *redone);
/* Calls since create whose AGC tail ran on a tile-major plane (k_agc_spec_tm: fused 256- and 1024-channel chains, run-sized calls of whole
 * 16-frame tiles; every other call takes the row-major k_agc_spec).  Both produce the sequential recurrence bit for bit. */
uint32_t csdr_chain_debug_agc_tile_major_calls(const csdr_chain *h);

/* the configuration a handle was created with (defaults filled in) */
int  csdr_chain_get_cfg(const csdr_chain *h, csdr_chain_cfg *cfg_out);

/* ------------------------------------------------------------------------ *
 * Collectives: one process per GPU, RCCL over xGMI (csrc/comm.cpp).
 *
 * The reference reduces `--mix` on one host thread: `mix` = foldl1 (+) over the channel
 * list (Trans.hs:119-122) behind `mux (replicate nch demod) . firpfbchChannelizer nc`
 * (SoapySDR.hs:217-222).  When the -c N channels are split over the GPUs of a node every
 * rank's chain folds its own channels and ONE all-reduce(SUM) of the nf output elements per
 * chunk adds the partial mixes (summation order differs from the strict left fold: tolerance).
 * Bootstrapping follows RCCL: rank 0 makes an id (csdr_comm_unique_id), the host hands the
 * CSDR_COMM_ID_BYTES to every rank by its own means (a file, an environment variable, the
 * Haskell program's command line), every rank calls csdr_comm_create with it (collective,
 * blocking).  librccl.so.1 is loaded on the first csdr_comm_* call, not with the library.
 * All calls enqueue on the caller's `stream` and do not synchronise.
 * ------------------------------------------------------------------------ */
#define CSDR_COMM_ID_BYTES 128
typedef struct csdr_comm csdr_comm;
int csdr_comm_unique_id(void *id_out);
int csdr_comm_create(int rank, int world, const void *id, int device /* -1 = current */, csdr_comm **out);
int csdr_comm_rank(const csdr_comm *c);
int csdr_comm_world(const csdr_comm *c);
int csdr_comm_destroy(csdr_comm *c);
/* d_buf (bytes) of rank `root` -> every rank: the chunk a channel-sharded node works on (8 B per input sample) */
int csdr_comm_broadcast(csdr_comm *c, void *d_buf, size_t bytes, int root, void *stream);
/* in-place sum over ranks of `count` floats */
int csdr_comm_allreduce_f32(csdr_comm *c, void *d_buf, size_t count, void *stream);
/* csdr_chain_process_device of a channel-shard handle created with mix = 1, then the all-reduce of its output: d_out holds
 * the mix over ALL channels on every rank (Trans.hs:119-122 across ranks).  *n_out = elements (nf). */
int csdr_chain_process_device_mix(csdr_chain *h, csdr_comm *c, const void *d_in_cf32, uint32_t n_in, void *d_out, uint32_t *n_out,
                                  void *stream);
/* The same for host buffers (blocking, like csdr_chain_process): what a host that keeps its chunks in its own memory calls --
 * the Haskell fold, host/soapy_sdr_file.cpp --world/--rank.  Every rank receives the full mix; the reference has ONE sink for it
 * (SoapySDR.hs:217-222), so rank 0 writes the file. */
int csdr_chain_process_mix(csdr_chain *h, csdr_comm *c, const float *in_cf32, uint32_t n_in, void *out, uint32_t *n_out);
/* The exchange of the hybrid partition (replaces the hand-over between firpfbchChannelizer and `mux`, Trans.hs:124-129,
 * when time stripes feed channel-block tails): d_plane = this rank's front-end output [world][chan_per_rank][stripe_frames[rank]]
 * (a DeNo chain's channel-major plane: destination p's rows are contiguous), elem_bytes 8 (CF32) or 4.  d_recv receives, for
 * p = 0 .. world-1 in time order, rank p's stripe of MY channel block: [chan_per_rank][stripe_frames[p]] at element offset
 * chan_per_rank * sum_{q<p} stripe_frames[q].  One grouped ncclSend / ncclRecv per peer (every xGMI link carries one block each way). */
int csdr_hybrid_exchange(csdr_comm *c, const void *d_plane, void *d_recv, uint32_t chan_per_rank, const uint32_t *stripe_frames,
                         uint32_t elem_bytes, void *stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif /* CSDR_H */
