// Packed-f32 radix-4 / radix-16 butterflies shared by the any-M DFT kernels (kernels_generic.hip) and the fused
// FIR + DFT kernel for M = 1024 (kernels_pfb1024.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace csdr {

typedef float v2fg __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2fg g_mulmj(v2fg a) { return (v2fg){a.y, -a.x}; }
__device__ __forceinline__ v2fg g_cmul(v2fg a, v2fg b)
{
    const v2fg bx = {-b.y, b.x};
    return __builtin_elementwise_fma((v2fg){a.y, a.y}, bx, (v2fg){a.x, a.x} * b);
}
__device__ __forceinline__ void g_bfly4(v2fg &x0, v2fg &x1, v2fg &x2, v2fg &x3)
{
    const v2fg s02 = x0 + x2, d02 = x0 - x2, s13 = x1 + x3, d13 = g_mulmj(x1 - x3);
    x0 = s02 + s13; x1 = d02 + d13; x2 = s02 - s13; x3 = d02 - d13;
}
// natural-order input, output slot i holds X[(i >> 2) + 4 * (i & 3)]
__device__ __forceinline__ void g_fft16(v2fg (&v)[16])
{
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
#pragma unroll
    for (int a = 0; a < 4; a++) g_bfly4(v[a], v[a + 4], v[a + 8], v[a + 12]);
    v[5] = g_cmul(v[5], (v2fg){C1, -S1});   v[9] = g_cmul(v[9], (v2fg){R2, -R2});    v[13] = g_cmul(v[13], (v2fg){S1, -C1});
    v[6] = g_cmul(v[6], (v2fg){R2, -R2});   v[10] = g_mulmj(v[10]);                  v[14] = g_cmul(v[14], (v2fg){-R2, -R2});
    v[7] = g_cmul(v[7], (v2fg){S1, -C1});   v[11] = g_cmul(v[11], (v2fg){-R2, -R2}); v[15] = g_cmul(v[15], (v2fg){-C1, S1});
#pragma unroll
    for (int q = 0; q < 4; q++) g_bfly4(v[4 * q + 0], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}
#define GXIDX(i) (((i) >> 2) + 4 * ((i) & 3))

}  // namespace csdr
