// The float path of agc_crcf_execute (reference call: Liquid.chs:691-707; liquid-dsp agc_crcf), shared by every
// kernel that must agree bit for bit (k_agc in kernels_generic.hip, the time-parallel tail in kernels_agc_tail.hip).
//   y = x g ;  y2' <- (1 - alpha) y2' + alpha |y|^2 ;  g <- g exp(-alpha/2 ln y2') ;  g <- min(g, 1e6)
// The recurrence is what bounds the AGC kernels (a lone wave issues a dependent instruction every ~10 cycles,
// tools/probes/chain_probe.hip), so it is arranged to keep the dependent chain short:
//   * alpha |y|^2 = (alpha |x|^2) g^2: the input energy e = alpha |x|^2 is computed off the chain, the chain sees
//     g*g, e*(g*g) and one fma instead of x*g, y*y, fma, alpha*y2, fma (same value, rounded differently by an ulp);
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
    y2h = fmaf(1.0f - alpha, y2h, e * (g * g));
    const float upd = __builtin_amdgcn_exp2f((-0.5f * alpha) * __builtin_amdgcn_logf(y2h));
    // g = (y2h > 1e-6f) ? g * upd : g, the condition in an SGPR pair: the VCC select hipcc emits costs 16 cycles on the chain
    {
        unsigned long long mk;
        const float gu = g * upd, g0 = g, thr = 1e-6f;
        asm("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(mk) : "v"(y2h), "v"(thr));
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(g) : "v"(g0), "v"(gu), "s"(mk));
    }
    g = __builtin_amdgcn_fmed3f(g, 0.0f, 1e6f);
}

}  // namespace csdr
