// Helpers of the second-generation run kernels (kernels_fused_v2.hip: M = 256, kernels_run1024_v2.hip: M = 1024): LDS-only
// barrier, HBM -> LDS tile DMA (global_load_lds), packed freqdem (fm_quad).  What tools/probes/issue_probe*.hip measured on
// gfx950 is why they look the way they do; see the header of kernels_fused_v2.hip.  Product code.
#pragma once
#include "fused_common.h"

namespace csdr {
namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
#ifndef V2_DMA_AUX
#define V2_DMA_AUX ""       // cache-policy bits of the tile DMA (" nt", " sc1", ...) for A/B builds
#endif
#ifndef V2_DMA_AUX_STREAM
#define V2_DMA_AUX_STREAM V2_DMA_AUX    // the same for the tiles of a run that no other run's warm-up window covers (dma_tile<true>: selective policy, A/B builds)
#endif

__device__ __forceinline__ void bar()              // LDS-only barrier: outstanding global stores / LDS-DMA are not waited for
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ float opaque_v(float x) { asm volatile("" : "+v"(x)); return x; }

// HBM -> LDS without registers: lane l of wave instruction `it` fills 16-byte slot 64 (4 it + wave) + l of the RAW image.
// Spelled in asm so that hipcc does not count it: with the builtin it drains vmcnt(0) in front of the next LDS read.
// s_nop 4: the hazard recognizer does not look into inline asm, and a scalar operand that hipcc has just reloaded from a
// spill lane (v_readlane = a VALU write of an SGPR) must be five wait states old before a VMEM instruction reads it.
// The kernel waits for it by hand (s_waitcnt vmcnt(0) in front of the tile's output stores);
// lds_wave: LDS byte address of my wave's first slot.
// goff: byte offset of my first piece inside a tile; piece `it` lies 4096 bytes further (the swizzle term (q >> 1) & 7 of
// slot 64 (4 it + wave) + l does not depend on it), which goes onto the scalar base.
__device__ __forceinline__ unsigned dma_offset(int tid)
{
    const int wave = tid >> 6, lane = tid & 63;
    const int slot = 64 * wave + lane, q = slot >> 3, i = (slot & 7) ^ ((q >> 1) & 7);
    return (unsigned)(8 * q + i) * 16u;
}
// IT_STEP: LDS bytes between the destinations of consecutive instructions of a wave (4096 = the dense image; k_run256v2 pads
// its frames and passes 2 x its frame stride: instruction `it` of wave w lands in frame 2 it + (w >> 1))
// the same for an image whose run q keeps piece i in slot i ^ (q & 7) (k_run256v2's padded image: kernels_fused_v2.hip V2_RSW)
__device__ __forceinline__ unsigned dma_offset_rsw(int tid)
{
    const int wave = tid >> 6, lane = tid & 63;
    const int slot = 64 * wave + lane, q = slot >> 3, i = (slot & 7) ^ (q & 7);
    return (unsigned)(8 * q + i) * 16u;
}
template <bool STREAM = false, unsigned IT_STEP = 4096u>
__device__ __forceinline__ void dma_tile(const float4 *__restrict__ tile_base, unsigned goff, unsigned lds_wave)
{
    // the eight destination addresses are recomputed per call (one s_add each): as loop invariants they are sixteen SGPRs that
    // hipcc parks in spill lanes and reloads with v_readlane right here
    asm volatile("" : "+s"(lds_wave));
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const unsigned dst = lds_wave + IT_STEP * (unsigned)it;
        const float4 *src = tile_base + 256 * it;
        unsigned keep;
        if (STREAM)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %3" V2_DMA_AUX_STREAM "\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(goff), "s"(dst), "s"(src) : "memory");
        else
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %3" V2_DMA_AUX "\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(goff), "s"(dst), "s"(src) : "memory");
    }
}

// one of the eight instructions of dma_tile (piece `it`, a compile-time constant after unrolling): for kernels that spread a tile's DMA over
// their arithmetic instead of issuing it in one burst (the CU's address unit takes 16 cycles per 1 KiB wave instruction)
__device__ __forceinline__ void dma_piece(const float4 *__restrict__ tile_base, unsigned goff, unsigned lds_wave, const int it)
{
    asm volatile("" : "+s"(lds_wave));
    const unsigned dst = lds_wave + 4096u * (unsigned)it;
    const float4 *src = tile_base + 256 * it;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %3" V2_DMA_AUX "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(goff), "s"(dst), "s"(src) : "memory");
}

// ref * arg(conj(rp) r): the degree-15 minimax polynomial of scaled_atan2f with literal coefficients (v_fmaak), the
// scale applied to a = min/max before the last product; hp = ref pi/2, pi = ref pi, tiny = 1e-37 held in VGPRs
struct FmK { float tiny, ref, hp, pi; };
__device__ __forceinline__ float fm_sample(float2 rp, float2 r, const FmK &k)
{
    const float re = fmaf(rp.x, r.x, rp.y * r.y);
    const float im = fmaf(rp.x, r.y, -(rp.y * r.x));
    const float mx = fmaxf(fmaxf(fabsf(re), fabsf(im)), k.tiny);
    const float mn = fminf(fabsf(re), fabsf(im));
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
    t = sel_abs_gt(im, re, k.hp - t, t);
    t = sel_neg(re, k.pi - t, t);
    return copysignf(t, im);
}

// two samples at once: the complex products, the polynomial and the scalings as packed f32 (half the instructions: a lone
// wave issues one instruction per ~5.5 cycles whatever it is), selects and the transcendental per sample.  Coefficients
// are splat from VGPRs (VOP3P takes no literals).
struct FmK2 { float c[8]; float tiny, ref, hp, pi; };
__device__ __forceinline__ v2f conj_mul_v(v2f rp, v2f r)                 // conj(rp) * r = (rp.x r.x + rp.y r.y, rp.x r.y - rp.y r.x)
{
    v2f t, o;
    asm("v_pk_mul_f32 %0, %2, %3 op_sel:[1,1] op_sel_hi:[1,0] neg_hi:[1,0]\n\t"          // (rp.y r.y, -rp.y r.x)
        "v_pk_fma_f32 %1, %2, %3, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]"                   // (rp.x r.x, rp.x r.y) + t
        : "=&v"(t), "=&v"(o) : "v"(rp), "v"(r));
    return o;
}
__device__ __forceinline__ float max3_abs(float a, float b, float c)
{
    float r;
    asm("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float min_abs(float a, float b)
{
    float r;
    asm("v_min_f32_e64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// four samples (two packed pairs A, B) with the dependent chains of the two pairs interleaved step by step: a packed op
// that feeds the next packed op costs a wait state (s_nop) unless something independent sits in between
__device__ __forceinline__ void fm_quad(const float2 (&rp)[4], const float2 (&r)[4], const FmK2 &k, float (&m)[4])
{
    v2f q[4];
#pragma unroll
    for (int u = 0; u < 4; u++) q[u] = conj_mul_v(to_v(rp[u]), to_v(r[u]));                     // (re, im)
    const v2f mxA = {max3_abs(q[0].x, q[0].y, k.tiny), max3_abs(q[1].x, q[1].y, k.tiny)};
    const v2f mxB = {max3_abs(q[2].x, q[2].y, k.tiny), max3_abs(q[3].x, q[3].y, k.tiny)};
    const v2f mnA = {min_abs(q[0].x, q[0].y), min_abs(q[1].x, q[1].y)};
    const v2f mnB = {min_abs(q[2].x, q[2].y), min_abs(q[3].x, q[3].y)};
    const v2f rcA = {__builtin_amdgcn_rcpf(mxA.x), __builtin_amdgcn_rcpf(mxA.y)};
    const v2f rcB = {__builtin_amdgcn_rcpf(mxB.x), __builtin_amdgcn_rcpf(mxB.y)};
    const v2f aA = mnA * rcA, aB = mnB * rcB;
    const v2f zA = aA * aA, zB = aB * aB;
    const v2f refv = {k.ref, k.ref};
    const v2f sA = aA * refv, sB = aB * refv;
    v2f pA = __builtin_elementwise_fma((v2f){k.c[7], k.c[7]}, zA, (v2f){k.c[6], k.c[6]});
    v2f pB = __builtin_elementwise_fma((v2f){k.c[7], k.c[7]}, zB, (v2f){k.c[6], k.c[6]});
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        pA = __builtin_elementwise_fma(pA, zA, (v2f){k.c[i], k.c[i]});
        pB = __builtin_elementwise_fma(pB, zB, (v2f){k.c[i], k.c[i]});
    }
    const v2f tA = pA * sA, tB = pB * sB;
    const v2f hpv = {k.hp, k.hp}, piv = {k.pi, k.pi};
    const v2f thA = hpv - tA, thB = hpv - tB;
    const v2f uA = {sel_abs_gt(q[0].y, q[0].x, thA.x, tA.x), sel_abs_gt(q[1].y, q[1].x, thA.y, tA.y)};
    const v2f uB = {sel_abs_gt(q[2].y, q[2].x, thB.x, tB.x), sel_abs_gt(q[3].y, q[3].x, thB.y, tB.y)};
    const v2f wA = piv - uA, wB = piv - uB;
    m[0] = copysignf(sel_neg(q[0].x, wA.x, uA.x), q[0].y);
    m[1] = copysignf(sel_neg(q[1].x, wA.y, uA.y), q[1].y);
    m[2] = copysignf(sel_neg(q[2].x, wB.x, uB.x), q[2].y);
    m[3] = copysignf(sel_neg(q[3].x, wB.y, uB.y), q[3].y);
}

template <int CTRL> __device__ __forceinline__ float dpp_keep(float old, float v)      // lanes without a source keep `old`
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, 0xf, 0xf, false));
}

}  // namespace
// The run kernels re-read a few arguments from the kernarg segment right where they use them (k_run256v2, k_run1024v3: the state arrays of
// the launches without warm-up windows): as loop invariants they cost the tile loop
// SGPRs it does not have (the asm stores rely on a kernel without SGPR spills: tests/test_build_invariants.py).  A hand-written SCALAR
// load: a pointer fetched through the vector memory path would be waited for with an s_waitcnt vmcnt(N) that hipcc computes without
// knowing about the asm DMA / stores in flight (first version: the pointer was used before it had arrived).
template <size_t OFF> __device__ __forceinline__ unsigned kernarg_u32_s()
{
    unsigned v = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(4))) const char *kptr;
    kptr ka = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(ka), "n"(OFF) : "memory");
#endif
    return v;
}
template <size_t OFF> __device__ __forceinline__ float2 *kernarg_ptr_s()
{
    float2 *p = nullptr;
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(4))) const char *kptr;
    kptr ka = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    unsigned long long v;
    asm volatile("s_load_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(ka), "n"(OFF) : "memory");
    p = reinterpret_cast<float2 *>(v);
#endif
    return p;
}

}  // namespace csdr
