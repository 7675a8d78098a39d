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
    float a = mn * __builtin_amdgcn_rcpf(mx);
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

}  // namespace csdr
