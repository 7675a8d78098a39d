// Interface between the C ABI (capi.hip) and the fused channelizer kernels
// (kernels_fused.hip).  Product code.
#pragma once
#include "csdr_internal.h"

namespace csdr {

struct FusedConfig {
    uint32_t M, p, C, c0, max_nf;
    uint32_t G = 1;          // > 1: interleaved shard c0 of G (channels c0, c0 + G, ...; C = M / G rows out): k_run256v2<.., G>, G = 2, 4, 8
    bool dc_block; DcParams dc;
    bool fm; float fm_ref;
    bool mix;
    const float *taps;       // host, M*p
    uint32_t d_theta;
};

struct FusedCall {
    const float2 *d_in;      // nf*M new samples
    void *d_out;             // [C][nf] CF32 / F32, or [nf] when mixing
    uint32_t nf;
    uint32_t theta0;         // NCO phase of the first sample
    // pipelined entry point (csdr_chain_submit_device): run the launch without reading anything an earlier launch wrote, if the
    // plan can (fused_can_overlap); ev_tail is recorded on s once the chunk's last tiles are saved for the next call's run 0
    bool indep = false;
    hipEvent_t ev_tail = nullptr;
    // CF32 plans: write the output TILE-MAJOR -- block t / 16 holds the 128-byte lines of all C rows back to back (a tile's whole
    // output is one contiguous C x 128 bytes) -- instead of row-major [C][nf]: what the time-parallel AGC tail reads (k_agc_spec_tm).
    // Only for calls fused_tile_major_ok() accepts.
    bool tile_major = false;
};
bool fused_tile_major_ok(const FusedPlan *plan, uint32_t nf);
void fused_keep_tail(FusedPlan *plan);                  // from now on every run-kernel call saves its last WU + 1 raw tiles
bool fused_can_overlap(const FusedPlan *plan, uint32_t nf);
bool fused_tail_recorded(const FusedPlan *plan);       // the last call saved its tail (and recorded FusedCall::ev_tail)   // the next call of nf frames can run with FusedCall::indep

bool fused_supported(uint32_t M, uint32_t p);
int  fused_create(const FusedConfig &cfg, FusedPlan **out);
int  fused_reset(FusedPlan *plan, hipStream_t s);
int  fused_process(FusedPlan *plan, const FusedCall &call, hipStream_t s, KernelTimer *timer);
const char *fused_name(const FusedPlan *plan);
void fused_seek(FusedPlan *plan, uint64_t frames);   // after fused_reset: global frame index of the next frame
// sticky device-side error word (bit0/bit1: an inter-workgroup wait hit its spin limit); reads, then clears it; synchronises
int  fused_status(FusedPlan *plan, unsigned *status);
// CSDR_TRACE=1: per-tile s_memtime stamps (16 per tile) of the last launches; returns tiles copied
int  fused_trace(FusedPlan *plan, unsigned long long *out, uint32_t ntiles);
void fused_destroy(FusedPlan *plan);
// second-generation run kernel of the M = 256 chain (kernels_fused_v2.hip): whole-band calls of >= run_min_tiles tiles.
// run_args points at a RunArgs (fused_common.h) the plan fills.
int  run256_v2_launch(const void *run_args, bool fm, unsigned G, unsigned nruns, hipStream_t s);
int  run256_v2_blocks_per_cu(bool fm);
// RunArgs::nowu launches: the near-DC channels of every run's first 112 frames get what the true DC state at the run's start adds (Rt: host table
// [2 parities][DCFIX_F][4], fused_common.h)
int  run256_dcfix_launch(const void *run_args, bool fm, unsigned nruns, const float2 *Rt, hipStream_t s);
// round 4's experiment (tools/variants/kernels_run256_v3.hip, NOT part of the product library: measured not faster: DESIGN.md 8.1):
// one 512-thread workgroup per CU, front / back wave roles; linked only by tools/variants/build_run256_v3.sh (-DCSDR_WITH_RUN256_V3)
int  run256_v3_launch(const void *run_args, bool fm, unsigned nruns, hipStream_t s);


// M = 64 run kernel (kernels_fused_small.hip): same call interface
struct SmallPlan;
bool small_supported(uint32_t M, uint32_t p);
int  small_create(const FusedConfig &cfg, SmallPlan **out);
int  small_reset(SmallPlan *plan, hipStream_t s);
int  small_process(SmallPlan *plan, const FusedCall &call, hipStream_t s, KernelTimer *timer);
void small_seek(SmallPlan *plan, uint64_t frames);
const char *small_name(const SmallPlan *plan);
bool small_tile_major_ok(const SmallPlan *plan, uint32_t nf);     // the call goes to k_run64v2 as whole 64-frame tiles, no mix inside the plan
void small_destroy(SmallPlan *plan);


// M = 1024 run kernel (kernels_pfb1024.hip, DC = true): same call interface
struct BigPlan;
bool big_supported(uint32_t M, uint32_t p);
int  big_create(const FusedConfig &cfg, BigPlan **out);
int  big_reset(BigPlan *plan, hipStream_t s);
int  big_process(BigPlan *plan, const FusedCall &call, hipStream_t s, KernelTimer *timer);
void big_seek(BigPlan *plan, uint64_t frames);
const char *big_name(const BigPlan *plan);
bool big_tile_major_ok(const BigPlan *plan, uint32_t nf);     // FusedCall::tile_major for this call (CF32 output through k_run1024v3<CF32>)
void big_destroy(BigPlan *plan);

// M = 4096 (kernels_pfb4096.hip): branch-tiled front kernel (DC blocker + pre-mix + FIR + first radix-4 stage) -> z -> back kernel (four
// 1024-point DFTs per frame + tails); whole band; same call interface
struct HugePlan;
bool huge_supported(uint32_t M, uint32_t p);
int  huge_create(const FusedConfig &cfg, HugePlan **out);
int  huge_reset(HugePlan *plan, hipStream_t s);
int  huge_process(HugePlan *plan, const FusedCall &call, hipStream_t s, KernelTimer *timer);
void huge_seek(HugePlan *plan, uint64_t frames);
const char *huge_name(const HugePlan *plan);
bool huge_tile_major_ok(const HugePlan *plan, uint32_t nf);      // CF32 output, whole 16-frame blocks, no mix inside the plan
void huge_destroy(HugePlan *plan);

// k_run64v2 (kernels_run64_v2.hip): whole-band M = 64 calls with CF32 output and nf % 64 == 0; same state buffers as k_run64
struct Run64v2Host {
    const float2 *x; float2 *out;
    const float *taps; const float2 *tw, *wpre;
    const float2 *uhist_in; float2 *uhist_out; const float2 *vend_in; float2 *vend_out;
    // runs without warm-up windows (round 5, as k_run256v2: DESIGN 4): cpre [nruns + 1] DC state in front of every run's halo tile,
    // rt [2 parities][RUN64_DCFIX_F][4] the chain's response to a unit state at the channels 30..33; both null: warm-up windows
    float2 *cpre = nullptr; const float2 *rt = nullptr;
    uint32_t nf, nruns, parity0;
    uint32_t G = 1, g = 0;      // interleaved shard g of G (tables rotated by the plan)
    bool tile_major = false;    // CF32 plane of the AGC route: [block of 16 frames][64][128 B]
    bool dc_block;
    double beta;
};
constexpr int RUN64_DCFIX_F = 448;                      // output frames of a run the state correction covers (7 tiles: 32 768 samples behind the halo tile's start)
uint32_t run64_v2_runs(uint32_t nf, uint32_t cus);      // 0: not a call for the kernel
// Response of the chain (pre-mix table wpre [2][M], taps, forward DFT), at the channels k0 .. k0 + 3, to a DC-blocker state of 1 in
// front of a tile: frames f0 .. f0 + nfr - 1 behind the tile's start, both parities of the tile's first frame; rt [2][nfr][4], f64 inside
void dc_state_response(const FusedConfig &cfg, const float2 *wpre, uint32_t f0, uint32_t nfr, uint32_t k0, float2 *rt);
int run64_v2_launch(const Run64v2Host &h, hipStream_t s, KernelTimer *timer);

// k_run1024v2 (kernels_run1024_v2.hip): whole-band M = 1024 calls with nf % 4 == 0; same state buffers as k_run1024
struct Run1024v2Host {
    const float2 *x; void *out;
    const float4 *taps_q;       // [4][4][256] float4: taps of branch 256 q + j (14) + its even-frame pre-mix phasor
    const float2 *tw;
    const float2 *uhist_in; float2 *uhist_out; const float2 *vend_in; float2 *vend_out; const float2 *rp_in; float2 *rp_out;
    uint32_t nf, nruns, parity0;
    uint32_t G = 1, g = 0;      // interleaved shard g of G (tables rotated by the plan)
    bool tile_major = false;    // k_run1024v3<CF32>: the lines of a 16-frame block back to back ([block][1024][128 B]) instead of row-major [1024][nf]
    // k_run1024v3 without warm-up windows (round 5, as k_run256v2: DESIGN 4): cpre [nruns + 1] DC state four tiles in front of every run's
    // first tile, side [nruns][4][RUN1024_DCFIX_F] uncorrected Y of the channels 510..513 (FM), rt [2][RUN1024_DCFIX_F][4] the chain's
    // response to a unit state there; null: warm-up windows
    float2 *cpre = nullptr, *side = nullptr; const float2 *rt = nullptr;
    int mfix = -1;              // k_shard1024: the shard's row among the four channels around DC (side [nruns][RUN1024_DCFIX_F], rt channel 0), -1: none
    bool dc_block;
    double beta;
    float fm_ref;
};
uint32_t run1024_v2_runs(uint32_t nf, uint32_t cus);    // 0: call too short for the kernel
int run1024_v2_launch(const Run1024v2Host &h, bool fm, hipStream_t s, KernelTimer *timer);
// third-generation FM kernel (kernels_run1024_v3.hip): one 512-thread workgroup per CU, output lines staged in registers (no staging
// block); whole band, calls of whole output lines (nf = 0 mod 32 frames F32 / 16 frames CF32)
// frame -1 of a run (freqdem history) + its first eight tiles.  The first frame NOT corrected (32) has a window that reaches back to frame
// 35 behind the cold start: alpha |c| beta^35840 = 1.6e-8 alpha |c| of the missing state is left in it (with four tiles it was 5.9e-5:
// ~1e-5 of a unit signal at the strongest DC offset the tests use)
constexpr int RUN1024_DCFIX_F = 33;
// interleaved-shard run kernel (kernels_shard1024.hip, round 6): chan_stride G = 4, 8, F32 or CF32 rows of the owned channels; the fold of the
// aliasing branches behind the FIR + a (1024 / G)-point DFT across the lanes of a wave; same state arrays and tables as the plan's other kernels
uint32_t shard1024_runs(uint32_t nf, bool fm, uint32_t G, uint32_t cus);    // 0: the call is not for this kernel
int shard1024_launch(const Run1024v2Host &h, bool fm, uint32_t nruns, hipStream_t s, KernelTimer *timer);
uint32_t run1024_v3_runs(uint32_t nf, bool fm, uint32_t cus);    // 0: the call is not for this kernel
int run1024_v3_launch(const Run1024v2Host &h, bool fm, uint32_t nruns, hipStream_t s, KernelTimer *timer);


// single-pass DC blocker + NCO mix for whole chunks of the generic path (kernels_dc_tile.hip)
struct DcTilePlan;
int  dctile_create(const DcParams &dc, uint64_t max_samples, DcTilePlan **out);
int  dctile_reset(DcTilePlan *plan, hipStream_t s);
// pick > 0: y receives only the samples whose index in this call is a multiple of `pick` (n / pick of them)
int  dctile_process(DcTilePlan *plan, const float2 *x, float2 *y, uint32_t n, bool do_mix, const NcoParams &nco,
                    const float2 *nco_tab, hipStream_t s, uint32_t pick = 0);
// out[t] = M sum_n taps[(M-1) + n M] u0[(p-1) + t - n]; hist_out <- the last p - 1 samples of u0 (the next call's history)
bool dctile_mix_identity_supported(const DcTilePlan *plan, uint32_t M, uint32_t n, uint32_t taps_p);
int  dctile_mix_identity(DcTilePlan *plan, const float2 *x, uint32_t n, const NcoParams &nco, const float2 *nco_tab, const float *taps,
                         uint32_t M, uint32_t taps_p, const float2 *hist_in, float2 *hist_out, float2 *out, hipStream_t s);
// the mix identity of an interleaved channel shard g of G (G <= 8, (M / G) % 512 == 0): (M / G) sum_{n2 < G} W_G^(n2 g) x (FIR of branch
// (M / G) n2); hist_in / hist_out: [G][p - 1]
bool dctile_mix_identity_shard_supported(const DcTilePlan *plan, uint32_t M, uint32_t n, uint32_t taps_p, uint32_t G);
int  dctile_mix_identity_shard(DcTilePlan *plan, const float2 *x, uint32_t n, const NcoParams &nco, const float2 *nco_tab, const float *taps,
                               uint32_t M, uint32_t taps_p, uint32_t G, uint32_t g, const float2 *hist_in, float2 *hist_out, float2 *out, hipStream_t s);
int  launch_branch0_fir(const float2 *u0, const float *taps, float2 *out, float2 *hist_out, uint32_t M, uint32_t p, uint32_t nf, hipStream_t s);
// sticky device-side error word (a look-back wait hit its spin limit); reads, then clears it; synchronises
int  dctile_status(DcTilePlan *plan, unsigned *status);
void dctile_destroy(DcTilePlan *plan);

}  // namespace csdr
