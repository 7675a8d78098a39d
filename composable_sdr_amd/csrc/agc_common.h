// The float path of agc_crcf_execute (reference call: Liquid.chs:691-707; liquid-dsp agc_crcf), shared by every
// kernel that must agree bit for bit (k_agc in kernels_generic.hip, the time-parallel tail in kernels_agc_tail.hip).
//   y = x g ;  y2' <- (1 - alpha) y2' + alpha |y|^2 ;  g <- g exp(-alpha/2 ln y2') ;  g <- min(g, 1e6)
// The recurrence is what bounds the AGC kernels (a lone wave issues a dependent instruction every ~10 cycles,
// tools/probes/chain_probe.hip), so it is arranged to keep the dependent chain short:
//   * alpha |y|^2 = ((alpha |x|^2) g) g: the input energy e = alpha |x|^2 is computed off the chain, the chain sees
//     e*g and one fma instead of x*g, y*y, fma, alpha*y2, fma (same value, rounded differently by an ulp);
//   * exp(-alpha/2 ln y2') = 2^(-alpha/2 log2 y2') with the hardware log2/exp2 (1 ulp);
//   * min(g, 1e6) as v_med3_f32(g, 0, 1e6) (g > 0 always): one instruction, no NaN canonicalisation in front.
// 120 -> 60 cycles per sample for a lone wave.  Every multiply that feeds an add is an explicit fmaf, so the result
// depends on the inputs only (see fm_common.h).
#pragma once
#include <hip/hip_runtime.h>

namespace csdr {

__device__ __forceinline__ float agc_energy(float2 x, float alpha) { return alpha * fmaf(x.x, x.x, x.y * x.y); }

__device__ __forceinline__ void agc_gain_update(float e, float &g, float &y2h, float alpha)
{
    // round 4: the chain through g is  e g -> fma -> log2 -> mul -> exp2 -> g upd -> med3  (7 dependent instructions, was 10):
    //   * y2' = (1 - alpha) y2' + (e g) g as fmaf(e g, g, (1 - alpha) y2'): the product with y2' runs beside the chain;
    //   * "g <- min(g upd, 1e6) if y2' > 1e-6, else g" is ONE v_med3 whose bounds the condition picks -- (0, 1e6), or (g, g), which
    //     returns g whatever g upd is (inf / NaN for y2' = 0 included) -- and the bounds are ready long before exp2 is.
    const float t2 = (1.0f - alpha) * y2h;
    y2h = fmaf(e * g, g, t2);
    const float upd = __builtin_amdgcn_exp2f((-0.5f * alpha) * __builtin_amdgcn_logf(y2h));
    unsigned long long mk;
    float lo, hi;
    const float thr = 1e-6f, zero = 0.0f, top = 1e6f;
    asm("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(mk) : "v"(y2h), "v"(thr));
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(lo) : "v"(g), "v"(zero), "s"(mk));
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(hi) : "v"(g), "v"(top), "s"(mk));
    g = __builtin_amdgcn_fmed3f(g * upd, lo, hi);
}

}  // namespace csdr
