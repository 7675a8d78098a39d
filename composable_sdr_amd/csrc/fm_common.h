// freqdem sample shared by every kernel that is compared bit-for-bit with another one
// (k_fm / k_transpose_fm / k_mix_frames in kernels_generic.hip, the AGC tail in kernels_agc_tail.hip).
// liquid's freqdem_demodulate (reference call: Liquid.chs:303-334) is m = arg(conj(r') r) / (2 pi kf).
// Every multiply that feeds an add is written as an explicit fmaf (HIP's __fmul_rn / __fadd_rn are plain
// operators that -ffp-contract=fast fuses or not depending on the surrounding code -- measured: 1e-6 relative
// differences between two kernels on cancelling products), so the result depends on the inputs only.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace csdr {

// atan2f: odd minimax polynomial of degree 17 on [0,1] (fit error 6e-9, evaluation error <= 1.2e-7 rad) +
// octant folding; signed zeros and infinities follow IEEE atan2 (arg(conj(0) r) = atan2(+-0, +-0) matters:
// the first freqdem output after a muted sample is +-0 or +-pi)
__device__ __forceinline__ float atan2f_rn(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    // mn / mx, with 0/0 -> 0 and inf/inf -> 1: those are the two cases in which mn * rcp(mx) is 0 * inf = NaN
    // (finite / inf is already 0).  Written as selects on the product: with the tests on mx in front the compiler
    // builds exec-masked branches around the rcp, two per sample, on the critical path of the AGC tail.
    // (round 5) v_rcp_f32 takes a denormal for zero: with the AGC's gain collapsed to ~1e-21 (the reference's g0 = 1000 start transient on
    // a strong channel) the products conj(r') r are ~1e-41 and mn * rcp(mx) came out inf -> NaN outputs where liquid's
    // cargf gives an angle.  Products below 1e-30 are scaled by 2^90 first (exact; nothing changes for any other input).
    const float sc = (mx < 1e-30f) ? 0x1p90f : 1.0f;
    float a = (mn * sc) * __builtin_amdgcn_rcpf(mx * sc);
    const float t = (mx == 0.0f) ? 0.0f : 1.0f;
    a = (a == a) ? a : t;
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
    // r = p*a, or pi/2 - p*a in the upper octant: both spelled out so that neither can be re-fused
    const float lo = p * a, hi = fmaf(-p, a, 1.57079632679489662f);
    float r = (ay > ax) ? hi : lo;
    const float rn = 3.14159265358979324f - r;                  // r comes out of a select: nothing to fuse with
    r = (__float_as_uint(x) >> 31) ? rn : r;
    return copysignf(r, y);
}

__device__ __forceinline__ float fm_sample_rn(float2 rp, float2 r, float ref)
{
    const float re = fmaf(rp.x, r.x, rp.y * r.y);
    const float im = fmaf(rp.x, r.y, -(rp.y * r.x));
    return atan2f_rn(im, re) * ref;
}

// ---- four samples at once, BIT-IDENTICAL to four fm_sample_rn calls ----
// The products, the polynomial and the scalings run as packed f32 (v_pk_mul / v_pk_fma: one IEEE operation per component, the
// same roundings as the scalar code above), the selects take their condition from an SGPR pair (the VCC form hipcc emits costs 16
// cycles per wave on gfx950, the SGPR form 4.6: tools/probes/issue_probe2.hip), max / min / rcp stay per sample.  ~21
// instructions per sample instead of ~33: what the worker wave of the time-parallel AGC tail (kernels_agc_tail.hip) is paced by.
typedef float fm_v2f __attribute__((ext_vector_type(2)));
struct FmRnK { float c[9]; float hp, pi, ref, tthr, tsc; };         // VOP3P takes no literals: the constants live in VGPRs
__device__ __forceinline__ float fm_opaque(float x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ FmRnK fm_rn_consts(float ref)
{
    FmRnK k;
    const float c[9] = {9.999998808e-01f, -3.333259821e-01f, 1.998590529e-01f, -1.416121870e-01f, 1.049891263e-01f,
                        -7.234797627e-02f, 3.978060186e-02f, -1.440101303e-02f, 2.456645248e-03f};
#pragma unroll
    for (int i = 0; i < 9; i++) k.c[i] = fm_opaque(c[i]);
    k.hp = fm_opaque(1.57079632679489662f); k.pi = fm_opaque(3.14159265358979324f); k.ref = fm_opaque(ref);
    k.tthr = fm_opaque(1e-30f); k.tsc = fm_opaque(0x1p90f);
    return k;
}
__device__ __forceinline__ float fm_sel(unsigned long long m, float t, float f)              // m ? t : f, condition in an SGPR pair
{
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(t), "s"(m));
    return r;
}
__device__ __forceinline__ void fm_quad_rn(const float2 (&rp)[4], const float2 (&r)[4], const FmRnK &k, float (&m)[4])
{
    fm_v2f q[4];                                                     // (re, im) = conj(rp) r
#pragma unroll
    for (int u = 0; u < 4; u++) {
        fm_v2f t, o;
        const fm_v2f a = {rp[u].x, rp[u].y}, b = {r[u].x, r[u].y};
        asm("v_pk_mul_f32 %0, %2, %3 op_sel:[1,1] op_sel_hi:[1,0] neg_hi:[1,0]\n\t"        // (rp.y r.y, -(rp.y r.x))
            "v_pk_fma_f32 %1, %2, %3, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]"                 // (fma(rp.x, r.x, .), fma(rp.x, r.y, .))
            : "=&v"(t), "=&v"(o) : "v"(a), "v"(b));
        q[u] = o;
    }
    float a_[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        float mx, mn;
        asm("v_max_f32_e64 %0, |%1|, |%2|" : "=v"(mx) : "v"(q[u].x), "v"(q[u].y));
        asm("v_min_f32_e64 %0, |%1|, |%2|" : "=v"(mn) : "v"(q[u].x), "v"(q[u].y));
        // products below 1e-30 (a collapsed AGC gain) are scaled by 2^90 in front of the rcp, exactly as atan2f_rn does
        unsigned long long mt;
        float sc;
        asm("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(mt) : "v"(mx), "v"(k.tthr));
        asm("v_cndmask_b32_e64 %0, 1.0, %1, %2" : "=v"(sc) : "v"(k.tsc), "s"(mt));
        float a = (mn * sc) * __builtin_amdgcn_rcpf(mx * sc);
        // a = (a == a) ? a : ((mx == 0) ? 0 : 1)
        unsigned long long mz, mo;
        float t1;
        asm("v_cmp_eq_f32_e64 %0, %1, 0" : "=s"(mz) : "v"(mx));
        asm("v_cndmask_b32_e64 %0, 1.0, 0, %1" : "=v"(t1) : "s"(mz));
        asm("v_cmp_o_f32_e64 %0, %1, %1" : "=s"(mo) : "v"(a));
        a_[u] = fm_sel(mo, a, t1);
    }
#pragma unroll
    for (int h = 0; h < 4; h += 2) {
        const fm_v2f a = {a_[h], a_[h + 1]};
        const fm_v2f z = a * a;
        fm_v2f p = {k.c[8], k.c[8]};
#pragma unroll
        for (int i = 7; i >= 0; i--) p = __builtin_elementwise_fma(p, z, (fm_v2f){k.c[i], k.c[i]});
        const fm_v2f lo = p * a;
        const fm_v2f hi = __builtin_elementwise_fma(-p, a, (fm_v2f){k.hp, k.hp});
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const float re = q[h + u].x, im = q[h + u].y;
            unsigned long long mg, ms;
            asm("v_cmp_gt_f32_e64 %0, |%1|, |%2|" : "=s"(mg) : "v"(im), "v"(re));            // ay > ax
            float rr = fm_sel(mg, u ? hi.y : hi.x, u ? lo.y : lo.x);
            const float rn = k.pi - rr;
            asm("v_cmp_gt_i32_e64 %0, 0, %1" : "=s"(ms) : "v"(re));                          // sign bit of x (incl. -0)
            rr = fm_sel(ms, rn, rr);
            m[h + u] = copysignf(rr, im) * k.ref;
        }
    }
}

}  // namespace csdr
