// Device helpers shared by the fused kernels (kernels_fused.hip: M = 256; kernels_fused_small.hip:
// M = 64).  Everything here is M-agnostic: a TILE is 4096 consecutive input samples, a DC "row" is
// 256 samples (16 runs of 16), whatever the channel count.  Product code.
#pragma once
#include "fused.h"

#include <cmath>
#include <cstdlib>
#include <vector>

namespace csdr {
namespace {

constexpr int M256 = 256;
constexpr int P = 14;            // taps per branch (2m, m = 7)
constexpr int NB = 16;           // frames per tile
constexpr int FS_X = 272;        // float2 stride between frames, FIR -> pass-1 layout
constexpr int FS_Z = 289;        // float2 stride between frames, pass-1 -> pass-2 layout
constexpr int RS_Z = 18;         // float2 stride between k1 rows inside a frame (16 + 2 pad)
constexpr int RS_Y = 17;         // float2 stride between channel rows, pass-2 -> tail layout
constexpr int LDS_F2 = 16 * FS_Z;   // 4624 float2 = 36992 B (largest layout)
constexpr int E_OFF = 4096;      // float2 offset of the two 256-entry run-carry tables
constexpr int LOOKBACK = 10;     // tiles; beta^(4096*10) ~ 1.3e-9 for alpha = 0.0005
constexpr unsigned SPIN_LIMIT = 1u << 24;

typedef unsigned long long u64;
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 mulmj(float2 a) { return make_float2(a.y, -a.x); }   // * -j
__device__ __forceinline__ float2 cfma(float2 a, float s, float2 b)                    // a*s + b
{
    return make_float2(fmaf(a.x, s, b.x), fmaf(a.y, s, b.y));
}

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
#pragma unroll
    for (int a = 0; a < 4; a++) bfly4(v[a], v[a + 4], v[a + 8], v[a + 12]);
    v[1 + 4] = cmul(v[1 + 4], make_float2(C1, -S1));            // W16^1
    v[1 + 8] = cmul(v[1 + 8], make_float2(R2, -R2));            // W16^2
    v[1 + 12] = cmul(v[1 + 12], make_float2(S1, -C1));          // W16^3
    v[2 + 4] = cmul(v[2 + 4], make_float2(R2, -R2));            // W16^2
    v[2 + 8] = mulmj(v[2 + 8]);                                 // W16^4 = -j
    v[2 + 12] = cmul(v[2 + 12], make_float2(-R2, -R2));         // W16^6
    v[3 + 4] = cmul(v[3 + 4], make_float2(S1, -C1));            // W16^3
    v[3 + 8] = cmul(v[3 + 8], make_float2(-R2, -R2));           // W16^6
    v[3 + 12] = cmul(v[3 + 12], make_float2(-C1, S1));          // W16^9
#pragma unroll
    for (int q = 0; q < 4; q++) bfly4(v[4 * q + 0], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int r = q + 1; r < 4; r++) {
            const float2 t = v[4 * q + r];
            v[4 * q + r] = v[4 * r + q];
            v[4 * r + q] = t;
        }
}


// ---- packed-f32 complex helpers (v2f = {re, im}) ----
// hipcc folds a broadcast ({a.x, a.x}) or a whole-vector negation into VOP3P modifiers but not a swap of the
// two halves, so a complex multiply costs it 4 instructions (v_xor + v_mov + pk_mul + pk_fma) and a *(-j) two
// v_movs.  The helpers below spell the instruction with op_sel / neg_lo / neg_hi by hand: 2 per complex
// multiply, 0 extra per +-j rotation.  gfx950 interlocks dependent packed ops in hardware
// (tools/probes/pk_hazard_probe.hip), so no wait states are needed inside a block.
__device__ __forceinline__ v2f to_v(float2 a) { return (v2f){a.x, a.y}; }
__device__ __forceinline__ float2 to_f2(v2f a) { return make_float2(a.x, a.y); }
__device__ __forceinline__ v2f cmul_v(v2f a, v2f w)                    // a * w
{
    v2f t, r;
    asm("v_pk_mul_f32 %0, %2, %3 op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %1, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=&v"(t), "=&v"(r) : "v"(a), "v"(w));
    return r;
}
__device__ __forceinline__ void cmul2_v(v2f &a0, v2f w0, v2f &a1, v2f w1)   // two independent products, interleaved
{
    v2f t0, t1, r0, r1;
    asm("v_pk_mul_f32 %0, %4, %5 op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %1, %6, %7 op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %2, %4, %5, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
        "v_pk_fma_f32 %3, %6, %7, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=&v"(t0), "=&v"(t1), "=&v"(r0), "=&v"(r1) : "v"(a0), "v"(w0), "v"(a1), "v"(w1));
    a0 = r0; a1 = r1;
}
// a * (SL*P[I0] + j*SH*P[I1]) for a uniform constant pair P held in SGPRs; NL1 = (SL<0), NH1 = (SH<0),
// NL2 = (SH>0), NH2 = (SL<0)
#define CSDR_CMULK(NAME, I0, NL1, I1, NH1, NL2, NH2)                                                              \
    __device__ __forceinline__ v2f NAME(v2f a, v2f P)                                                            \
    {                                                                                                            \
        v2f t, r;                                                                                                \
        asm("v_pk_mul_f32 %0, %2, %3 op_sel:[0," #I0 "] op_sel_hi:[0," #I1 "] neg_lo:[0," #NL1 "] neg_hi:[0," #NH1 "]\n\t" \
            "v_pk_fma_f32 %1, %2, %3, %0 op_sel:[1," #I1 ",0] op_sel_hi:[1," #I0 ",1] neg_lo:[0," #NL2 ",0] neg_hi:[0," #NH2 ",0]" \
            : "=&v"(t), "=&v"(r) : "v"(a), "s"(P));                                                              \
        return r;                                                                                                \
    }
CSDR_CMULK(cmul_w1, 0, 0, 1, 1, 0, 0)      // P = (C1, S1): * (C1 - j S1) = W16^1
CSDR_CMULK(cmul_w3, 1, 0, 0, 1, 0, 0)      //               * (S1 - j C1) = W16^3
CSDR_CMULK(cmul_w9, 0, 1, 1, 0, 1, 1)      //               * (-C1 + j S1) = W16^9
CSDR_CMULK(cmul_w2, 0, 0, 0, 1, 0, 0)      // P = (R2, *):  * (R2 - j R2) = W16^2
CSDR_CMULK(cmul_w6, 0, 1, 0, 1, 0, 1)      //               * (-R2 - j R2) = W16^6
#undef CSDR_CMULK
__device__ __forceinline__ v2f add_mj(v2f x, v2f d)                     // x + (-j) d = x + (d.y, -d.x)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(d));
    return r;
}
__device__ __forceinline__ v2f sub_mj(v2f x, v2f d)                     // x - (-j) d = x + (-d.y, d.x)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(x), "v"(d));
    return r;
}
__device__ __forceinline__ void bfly4_v(v2f &x0, v2f &x1, v2f &x2, v2f &x3)
{
    const v2f s02 = x0 + x2, d02 = x0 - x2, s13 = x1 + x3, d = x1 - x3;
    x0 = s02 + s13; x2 = s02 - s13; x1 = add_mj(d02, d); x3 = sub_mj(d02, d);
}
__device__ __forceinline__ void bfly4_v_mj2(v2f &x0, v2f &x1, v2f &x2, v2f &x3)      // same with x2 standing for (-j) x2
{
    const v2f s02 = add_mj(x0, x2), d02 = sub_mj(x0, x2), s13 = x1 + x3, d = x1 - x3;
    x0 = s02 + s13; x2 = s02 - s13; x1 = add_mj(d02, d); x3 = sub_mj(d02, d);
}
// forward 16-point DFT, natural-order input; OUTPUT INDEX PERMUTED: v[4q + r] = X[q + 4r]
__device__ __forceinline__ void fft16_v(v2f (&v)[16])
{
    const v2f P = {0.92387953251128674f, 0.38268343236508977f}, Q = {0.70710678118654752f, 0.70710678118654752f};
#pragma unroll
    for (int a = 0; a < 4; a++) bfly4_v(v[a], v[a + 4], v[a + 8], v[a + 12]);
    v[5] = cmul_w1(v[5], P);   v[9] = cmul_w2(v[9], Q);    v[13] = cmul_w3(v[13], P);
    v[6] = cmul_w2(v[6], Q);   /* v[10] *= -j: folded */   v[14] = cmul_w6(v[14], Q);
    v[7] = cmul_w3(v[7], P);   v[11] = cmul_w6(v[11], Q);  v[15] = cmul_w9(v[15], P);
    bfly4_v(v[0], v[1], v[2], v[3]);
    bfly4_v(v[4], v[5], v[6], v[7]);
    bfly4_v_mj2(v[8], v[9], v[10], v[11]);
    bfly4_v(v[12], v[13], v[14], v[15]);
}
#define XIDX(i) (((i) >> 2) + 4 * ((i) & 3))   /* register slot i of fft16_v holds X[XIDX(i)] */

// atan2f for the freqdem tail: odd minimax polynomial of degree 17 on [0,1] (fit error 6e-9,
// f32 evaluation error <= 1.2e-7 rad) + octant folding.  Signed zeros follow IEEE atan2
// (atan2(+-0, -0) = +-pi, atan2(+-0, +0) = +-0), which is what cargf(conjf(0)*r) relies on.
__device__ __forceinline__ float fast_atan2f(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    float a = mn * __builtin_amdgcn_rcpf(mx);
    a = (mx == 0.0f) ? 0.0f : a;
    a = (mx == INFINITY) ? ((mn == INFINITY) ? 1.0f : 0.0f) : a;
    const float z = a * a;
    float p = 2.456645248e-03f;
    p = fmaf(p, z, -1.440101303e-02f);
    p = fmaf(p, z, 3.978060186e-02f);
    p = fmaf(p, z, -7.234797627e-02f);
    p = fmaf(p, z, 1.049891263e-01f);
    p = fmaf(p, z, -1.416121870e-01f);
    p = fmaf(p, z, 1.998590529e-01f);
    p = fmaf(p, z, -3.333259821e-01f);
    p = fmaf(p, z, 9.999998808e-01f);
    float r = p * a;
    r = (ay > ax) ? 1.57079632679489662f - r : r;
    r = (__float_as_uint(x) >> 31) ? 3.14159265358979324f - r : r;
    return copysignf(r, y);
}

// ref * atan2f(y, x) for the run kernel's tail: degree-15 odd minimax polynomial (f32 evaluation error
// <= 1.2e-7 rad), coefficients pre-scaled by ref; hp = ref*pi/2, pi = ref*pi.  Same signed-zero
// behaviour as fast_atan2f; inputs are finite by construction (no Inf guard).
struct PhaseK { float c[8]; float hp, pi, ref; };
__host__ __device__ __forceinline__ PhaseK phase_consts(float ref)
{
    PhaseK k;
    const float c[8] = {9.999993443e-01f, -3.332985938e-01f, 1.994656026e-01f, -1.390860826e-01f,
                        9.642146528e-02f, -5.591168255e-02f, 2.186254039e-02f, -4.054457881e-03f};
#pragma unroll
    for (int i = 0; i < 8; i++) k.c[i] = c[i] * ref;
    k.hp = 1.57079632679489662f * ref; k.pi = 3.14159265358979324f * ref; k.ref = ref;
    return k;
}
// Selects with the condition in an SGPR pair: on gfx950 the VCC form of v_cndmask (what hipcc emits for `c ? a : b`) costs
// 16 cycles per wave, the SGPR-mask form 4.6 (tools/probes/issue_probe2.hip).
__device__ __forceinline__ float sel_abs_gt(float a, float b, float t, float f)        // |a| > |b| ? t : f
{
    unsigned long long m; float r;
    asm("v_cmp_gt_f32_e64 %0, |%1|, |%2|" : "=s"(m) : "v"(a), "v"(b));
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(t), "s"(m));
    return r;
}
__device__ __forceinline__ float sel_neg(float a, float t, float f)                    // sign bit of a (incl. -0) ? t : f
{
    unsigned long long m; float r;
    asm("v_cmp_gt_i32_e64 %0, 0, %1" : "=s"(m) : "v"(a));
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(t), "s"(m));
    return r;
}
// ref * atan2f(y, x): literal (unscaled) coefficients keep the polynomial on plain v_fmaak (3 cycles; an SGPR operand costs
// 4.7), the scale goes onto a = min / max before the last product.  Same signed-zero behaviour as fast_atan2f.
__device__ __forceinline__ float scaled_atan2f(float y, float x, const PhaseK &k)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(fmaxf(ax, ay), 1e-37f), mn = fminf(ax, ay);
    const float a = mn * __builtin_amdgcn_rcpf(mx);
    const float z = a * a;
    float p = -4.054457881e-03f;
    p = fmaf(p, z, 2.186254039e-02f);
    p = fmaf(p, z, -5.591168255e-02f);
    p = fmaf(p, z, 9.642146528e-02f);
    p = fmaf(p, z, -1.390860826e-01f);
    p = fmaf(p, z, 1.994656026e-01f);
    p = fmaf(p, z, -3.332985938e-01f);
    p = fmaf(p, z, 9.999993443e-01f);
    float t = p * (a * k.ref);
    t = sel_abs_gt(y, x, k.hp - t, t);
    t = sel_neg(x, k.pi - t, t);
    return copysignf(t, y);
}

// 16 freqdem samples of one channel (k_run256's tail): the compiler interleaves the independent samples
__device__ __forceinline__ void freqdem16(const float2 (&v)[16], float2 prev, const PhaseK &k, float (&m)[16])
{
#pragma unroll
    for (int f = 0; f < 16; f++) {
        const float2 rp = f ? v[f - 1] : prev, r = v[f];
        m[f] = scaled_atan2f(fmaf(rp.x, r.y, -(rp.y * r.x)), fmaf(rp.x, r.x, rp.y * r.y), k);
    }
}

template <int CTRL> __device__ __forceinline__ float dpp(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL> __device__ __forceinline__ float2 dpp2(float2 v) { return make_float2(dpp<CTRL>(v.x), dpp<CTRL>(v.y)); }

struct TileArgs {
    const float2 *x;            // new input, nf*256 samples
    void *out;                  // [C][nf] F32 (FM) or CF32
    const float *taps;          // [P][256] prototype taps h[i + n*256]
    const float2 *tw;           // pass-1 twiddles, tw[16*k1 + b1] = W256^(b1*k1)
    const float2 *wpre;         // [2][256]: conj(nco phasor) of column j for even / odd global frames
    const float2 *yhist_in; float2 *yhist_out;   // [13][256] DC-blocked samples before the call / after it
    const float2 *vend_in;  float2 *vend_out;    // DC blocker state v1
    const float2 *rp_in;    float2 *rp_out;      // [C] freqdem r'
    unsigned *ticket;           // zeroed before every launch
    u64 *agg;                   // [nb][2] {epoch, bits} granules: tile aggregates
    u64 *ylast;                 // [nb][256] last Y frame of each tile
    unsigned *yflag;            // [nb]
    unsigned *status;           // sticky error word (spin limit hit)
    u64 *trace;                 // optional [nb][16] s_memtime stamps of thread 0 (CSDR_TRACE=1)
    uint32_t epoch, nf, nb, c0, C, parity0;
    uint32_t out_stride, out_t0;   // output row length (frames of the whole call) and this launch's first frame in it
    float alpha, beta, fm_ref;
    float wtile[LOOKBACK + 2];  // beta^(4096 k)
    float b16[16];              // beta^(16 r)
    float b256[17];             // beta^(256 f)
    float bj[16];               // beta^i
};

// Raw tile -> registers: 8 x 16 bytes per thread, addressed so that the LDS image below is
// XOR-swizzled at 16-byte granularity (both the run-major b128 accesses and the column-major
// b64 accesses are then bank-conflict free).
__device__ __forceinline__ void tile_load(const float4 *__restrict__ src, int valid_runs, float4 (&raw)[8], int tid)
{
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const int slot = 64 * (it * 4 + wave) + lane;          // 16-byte slot in LDS
        const int q = slot >> 3;                               // run (16 samples)
        const int i = (slot & 7) ^ ((q >> 1) & 7);             // which 16-byte piece of the run
        raw[it] = (q < valid_runs) ? src[8 * q + i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

__device__ __forceinline__ float2 scan_staged(float2 *R, float2 *E, float2 *T, const TileArgs &A, int tid);
// Registers -> LDS, then the zero-state DC scan of the tile.  On return R holds
// z[n] = x[n] - alpha*s[n-1] (s = scan inside the 16-sample run), E[q] the carry into run q from
// earlier runs of its frame, T[f] the frame totals.  Ends with a barrier.
__device__ __forceinline__ float2 stage_and_scan(const float4 (&raw)[8], float2 *R, float2 *E, float2 *T,
                                               const TileArgs &A, int tid)
{
    float4 *R4 = reinterpret_cast<float4 *>(R);
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int it = 0; it < 8; it++) R4[64 * (it * 4 + wave) + lane] = raw[it];
    __syncthreads();
    return scan_staged(R, E, T, A, tid);
}

// The same scan on a raw tile image that is already in LDS (k_run256v2 DMA's the halo tile of a run straight into it; the
// caller has waited for the DMA and synchronised).  Ends with a barrier.
__device__ __forceinline__ float2 scan_staged(float2 *R, float2 *E, float2 *T, const TileArgs &A, int tid)
{
    float4 *R4 = reinterpret_cast<float4 *>(R);
    const int q = tid, sw = (q >> 1) & 7;
    float2 s = make_float2(0.f, 0.f);
    const float na = -A.alpha, be = A.beta;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        float4 v = R4[8 * q + (i ^ sw)];
        float2 x0 = make_float2(v.x, v.y), x1 = make_float2(v.z, v.w);
        const float2 z0 = cfma(s, na, x0);
        s = cfma(s, be, x0);
        const float2 z1 = cfma(s, na, x1);
        s = cfma(s, be, x1);
        R4[8 * q + (i ^ sw)] = make_float4(z0.x, z0.y, z1.x, z1.y);
    }
    // inclusive decayed scan of the run totals across the 16 runs of a frame (one DPP row)
    float2 t;
    t = dpp2<0x111>(s); s = cfma(t, A.b16[1], s);
    t = dpp2<0x112>(s); s = cfma(t, A.b16[2], s);
    t = dpp2<0x114>(s); s = cfma(t, A.b16[4], s);
    t = dpp2<0x118>(s); s = cfma(t, A.b16[8], s);
    const float2 e = dpp2<0x111>(s);                           // v at my run's start (zero row carry)
    if (E) E[q] = e;
    if ((q & 15) == 15) T[q >> 4] = s;                         // row total
    __syncthreads();
    return e;
}

// scan_staged on an image whose frames are FS float2 apart (256 = dense; k_run256v2 pads its frames, kernels_fused_v2.hip)
// RSW: the image's run swizzle is q & 7 instead of (q >> 1) & 7 (kernels_fused_v2.hip V2_RSW)
template <int FS, bool RSW = false>
__device__ __forceinline__ float2 scan_staged_fs(float2 *R, float2 *E, float2 *T, const TileArgs &A, int tid)
{
    const int q = tid, sw = RSW ? (q & 7) : ((q >> 1) & 7);
    float4 *R4 = reinterpret_cast<float4 *>(R + FS * (q >> 4)) + 8 * (q & 15);
    float2 s = make_float2(0.f, 0.f);
    const float na = -A.alpha, be = A.beta;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        float4 v = R4[i ^ sw];
        float2 x0 = make_float2(v.x, v.y), x1 = make_float2(v.z, v.w);
        const float2 z0 = cfma(s, na, x0);
        s = cfma(s, be, x0);
        const float2 z1 = cfma(s, na, x1);
        s = cfma(s, be, x1);
        R4[i ^ sw] = make_float4(z0.x, z0.y, z1.x, z1.y);
    }
    float2 t;
    t = dpp2<0x111>(s); s = cfma(t, A.b16[1], s);
    t = dpp2<0x112>(s); s = cfma(t, A.b16[2], s);
    t = dpp2<0x114>(s); s = cfma(t, A.b16[4], s);
    t = dpp2<0x118>(s); s = cfma(t, A.b16[8], s);
    const float2 e = dpp2<0x111>(s);
    if (E) E[q] = e;
    if ((q & 15) == 15) T[q >> 4] = s;
    __syncthreads();
    return e;
}

// Zero-state v before frame f of a tile, from its 16 frame totals: every 16-lane row runs the
// same decayed DPP scan over T[0..15] and picks the entry of the frame before its own.
__device__ __forceinline__ float2 frame_carry_zero_state(const float2 *T, const TileArgs &A, int tid)
{
    float2 s = T[tid & 15];
    float2 t;
    t = dpp2<0x111>(s); s = cfma(t, A.b256[1], s);
    t = dpp2<0x112>(s); s = cfma(t, A.b256[2], s);
    t = dpp2<0x114>(s); s = cfma(t, A.b256[4], s);
    t = dpp2<0x118>(s); s = cfma(t, A.b256[8], s);
    // lane r now holds v after frame r; I need v after frame (f-1), f = tid >> 4
    const int f = (tid >> 4) & 15;
    const int srcl = (tid & 48) | ((f - 1) & 15);
    float2 r = make_float2(__shfl(s.x, srcl), __shfl(s.y, srcl));
    return f ? r : make_float2(0.f, 0.f);
}


constexpr int WU = 6;

// How a launch's tiles are split into runs.  Runs are grouped in "slots": slot s holds rps consecutive runs that
// share tiles[s] tiles evenly, starting at tile base[s].  One slot with rps = nruns is the plain balanced
// split; the run kernels use one slot per co-resident workgroup of a CU (blockIdx / #CUs) so that the share of
// a run can follow the issue priority its workgroup gets (oldest first) and all runs end together.
struct RunSplit {
    uint32_t rps, nslots;
    uint32_t base[8], tiles[8];
};
__host__ __device__ __forceinline__ void run_range(const RunSplit &sp, uint32_t w, uint32_t &first, uint32_t &last)
{
    const uint32_t s = w / sp.rps, i = w - s * sp.rps;
    first = sp.base[s] + (uint32_t)((unsigned long long)i * sp.tiles[s] / sp.rps);
    last = sp.base[s] + (uint32_t)((unsigned long long)(i + 1) * sp.tiles[s] / sp.rps);
}

struct RunArgs {
    TileArgs t;
    RunSplit split;
    PhaseK pk;                  // ref-scaled atan polynomial: uniform, so it lives in SGPRs
    uint32_t nruns;             // runs are balanced: run w covers tiles [w*nb/nruns, (w+1)*nb/nruns)
    float l2beta;               // log2(beta)
    uint32_t prio_div;          // > 0: rotate the wave priority per tile; CU slot of a run = blockIdx / prio_div
    uint32_t trace_light;       // CSDR_TRACE=2: only per-run s_memrealtime stamps (entry / warm-up / halo / end)
    const float4 *prev_tail;    // k_run256v2, indep: the previous chunk's last WU + 1 raw tiles (run 0's warm-up window and halo)
    uint32_t indep;             // k_run256v2: run 0 starts cold from prev_tail instead of the carried state: the launch reads nothing an earlier launch wrote
    uint32_t pair_align;        // k_run256v2<FM>: run boundaries rounded down to even tiles (whole 128-byte output lines per tile pair)
    uint32_t wu_batch6;         // k_run256v2: the six warm-up tiles in one batch of loads (0: two batches of three)
    uint32_t wu, wu_rot;        // k_run256v2: read-only warm-up tiles in front of a run's halo tile (WU = 6); runs walk them in rotated order
    uint32_t tile_step;         // k_run256v2 / v3: bytes between consecutive tiles in the output (16 frames x element size for row-major rows; C x 128 tile-major)
    // round 5, whole-band k_run256v2: no read-only warm-up window at a run's start.  The run starts its halo tile from DC state 0; what
    // the true state would have added is, 16 frames later, confined to the four channels around DC (the step's edge excites every channel
    // for 13 frames -- the length of the FIR -- inside the halo tile, which has no output): k_run256_dcfix adds c_true x (the chain's
    // response to a unit state, a host table) to the channels 126..129 of the run's first 112 frames.  cpre[w + 1]: the state in front of
    // run w's last tile (= run w + 1's halo tile), left by run w; side: [run][4][DCFIX_F] the uncorrected Y of those channels.
    uint32_t nowu;
    float2 *cpre, *side;
};
constexpr int DCFIX_F = 113;    // frame -1 of a run (freqdem history) + its first seven tiles

__device__ __forceinline__ float2 wg_sum(float2 v, float2 *red, int tid)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { v.x += __shfl_xor(v.x, d); v.y += __shfl_xor(v.y, d); }
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    const float2 r = cadd(cadd(red[0], red[1]), cadd(red[2], red[3]));
    __syncthreads();
    return r;
}

// ---- column-layout DC stage of k_run256 (no LDS staging) ----
// Thread j holds x[256 f + j], f = 0..15.  A run of 16 consecutive samples is exactly one 16-lane DPP row of one frame, so
// the zero-state scan inside a run is a row scan: s_i = sum_{k<=i} beta^(i-k) x_k by four v_fmac_f32_dpp steps
// (row_shr 1, 2, 4, 8 with beta^1, ^2, ^4, ^8; a lane without a source keeps its value), and
// z_i = x_i - alpha s_{i-1} is one more.  Four independent values are interleaved per block: a VGPR written by a VALU
// instruction must not be read through DPP for 2 wait states, and the compiler cannot see into the asm.
__device__ __forceinline__ void row_scan4(float &x0, float &x1, float &x2, float &x3, float &s0, float &s1, float &s2, float &s3,
                                          float b1, float b2, float b4, float b8, float na)
{
    asm volatile(
        "v_mov_b32_e32 %4, %0\n\t"
        "v_mov_b32_e32 %5, %1\n\t"
        "v_mov_b32_e32 %6, %2\n\t"
        "v_mov_b32_e32 %7, %3\n\t"
        "v_fmac_f32_dpp %4, %4, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %5, %5, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %6, %6, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %7, %7, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %4, %4, %9 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %5, %5, %9 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %6, %6, %9 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %7, %7, %9 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %4, %4, %10 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %5, %5, %10 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %6, %6, %10 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %7, %7, %10 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %4, %4, %11 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %5, %5, %11 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %6, %6, %11 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %7, %7, %11 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %4, %12 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %5, %12 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %6, %12 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %3, %7, %12 row_shr:1 row_mask:0xf bank_mask:0xf"
        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3)
        : "v"(b1), "v"(b2), "v"(b4), "v"(b8), "v"(na));
}

// nw[f] = x[256 f + j]: 16 coalesced 8-byte loads per thread (a wave instruction covers 512 contiguous bytes)
__device__ __forceinline__ void col_load(const float2 *__restrict__ src, float2 (&nw)[16], int j)
{
#pragma unroll
    for (int f = 0; f < 16; f++) nw[f] = src[256 * f + j];
}

// In: nw = x (column layout).  Out: nw = z = x - alpha * (zero-state scan inside the sample's run), TR[q] = total of
// run q (q = 16 f + (j >> 4)).  TR: 256 float2 of LDS; the caller synchronises before reading it.
__device__ __forceinline__ void col_run_scan(float2 (&nw)[16], float2 *TR, const TileArgs &A, int tid)
{
    const float b1 = A.bj[1], b2 = A.bj[2], b4 = A.bj[4], b8 = A.bj[8], na = -A.alpha;
#pragma unroll
    for (int f = 0; f < 16; f += 2) {
        float s0, s1, s2, s3;
        row_scan4(nw[f].x, nw[f].y, nw[f + 1].x, nw[f + 1].y, s0, s1, s2, s3, b1, b2, b4, b8, na);
        if ((tid & 15) == 15) {
            TR[16 * f + (tid >> 4)] = make_float2(s0, s1);
            TR[16 * (f + 1) + (tid >> 4)] = make_float2(s2, s3);
        }
    }
}

// run totals -> carry into every run from the earlier runs of its frame (zero frame carry) in E, frame totals in T
__device__ __forceinline__ void col_run_carries(const float2 *TR, float2 *E, float2 *T, const TileArgs &A, int tid)
{
    float2 s = TR[tid], t;
    t = dpp2<0x111>(s); s = cfma(t, A.b16[1], s);
    t = dpp2<0x112>(s); s = cfma(t, A.b16[2], s);
    t = dpp2<0x114>(s); s = cfma(t, A.b16[4], s);
    t = dpp2<0x118>(s); s = cfma(t, A.b16[8], s);
    E[tid] = dpp2<0x111>(s);
    if ((tid & 15) == 15) T[tid >> 4] = s;
}

// v after frame 15 of a tile (zero state) from its frame totals; every lane gets the value
__device__ __forceinline__ void frame_carries(const float2 *T, const TileArgs &A, int tid, float2 &before_mine, float2 &after_tile)
{
    float2 s = T[tid & 15];
    float2 t;
    t = dpp2<0x111>(s); s = cfma(t, A.b256[1], s);
    t = dpp2<0x112>(s); s = cfma(t, A.b256[2], s);
    t = dpp2<0x114>(s); s = cfma(t, A.b256[4], s);
    t = dpp2<0x118>(s); s = cfma(t, A.b256[8], s);
    const int f = (tid >> 4) & 15;
    const int srcl = (tid & 48) | ((f - 1) & 15);
    const float2 r = make_float2(__shfl(s.x, srcl), __shfl(s.y, srcl));
    before_mine = f ? r : make_float2(0.f, 0.f);
    const int endl = (tid & 48) | 15;
    after_tile = make_float2(__shfl(s.x, endl), __shfl(s.y, endl));
}


}  // namespace
}  // namespace csdr
