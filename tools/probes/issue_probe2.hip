// Issue-cost probe for gfx950: shader cycles per wave-instruction of the instruction kinds the fused channelizer
// kernels are made of, alone (1..4 waves per SIMD) and paired with another kind on the SAME SIMD (does an LDS or a
// packed stream hide under a VALU stream of the partner wave, or do the costs add?).
//   ./issue_probe
// Every wave runs REPS x 64 instructions of one kind on 8 independent register sets (no dependent chain) between two
// s_memtime reads; the table prints cycles per instruction per wave and per SIMD (cycles / waves on the SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>
#include <map>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP8(S) S S S S S S S S

enum Op { NONE = 0, FMA, PKFMA, PKFMA_BC, PKADD, PKMUL, PKADD_SEL, FMAC_DPP, MOV_DPP, RCP, CNDMASK, MAX3, PERM32, PERM16, BFI,
          DSW64, DSR64, DSR128, DSW128, DSW32, DSR32, MIX_PK_FMA, FMA_SGPR, ADD, NOPS,
          CND_E64, CND_VCC2, CMP_VCC, CMP_E64, MUL, FMAC, FMAAK, FMA_NEG, FMA_ABS, AND, LSHL, MIN, MED3, PKFMA_SGPR, PKFMA_NEG, MOV, MUL_DPP, ADD_NEG, SUB, XOR, FMA_2SRC, PKFMA_2SRC, FMAC_SGPR, MUL_SGPR };

template <int OP> __device__ __forceinline__ void body(float (&a)[8], v2f (&p)[8], float s, char *lds, int reps)
{
    const v2f w = {1.0000001f, 0.9999999f};
    const float c = 1e-9f;
    const unsigned la = (threadIdx.x & 63) * 8 + (threadIdx.x >> 6) * 4096;       // b64: conflict-free, per-wave region
    const unsigned la16 = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096;
    const unsigned la4 = (threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 4096;
    v4f q[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int r = 0; r < reps; r++) {
        if (OP == FMA) {
#define X(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n\t"
            asm volatile(REP8(R8(X)) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(w.x), "v"(c));
#undef X
        } else if (OP == FMA_SGPR) {
#define X(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n\t"
            asm volatile(REP8(R8(X)) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "s"(s), "v"(c));
#undef X
        } else if (OP == ADD) {
#define X(i) "v_add_f32_e32 %" #i ", %8, %" #i "\n\t"
            asm volatile(REP8(R8(X)) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));
#undef X
        } else if (OP == PKFMA) {
#define X(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n\t"
            asm volatile(REP8(R8(X)) : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(w), "v"(w));
#undef X
        } else if (OP == PKFMA_BC) {
#define X(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9 op_sel_hi:[1,0,1]\n\t"
            asm volatile(REP8(R8(X)) : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(w), "v"(w));
#undef X
        } else if (OP == PKADD) {
#define X(i) "v_pk_add_f32 %" #i ", %" #i ", %8\n\t"
            asm volatile(REP8(R8(X)) : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(w));
#undef X
        } else if (OP == PKADD_SEL) {
#define X(i) "v_pk_add_f32 %" #i ", %" #i ", %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
            asm volatile(REP8(R8(X)) : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(w));
#undef X
        } else if (OP == PKMUL) {
#define X(i) "v_pk_mul_f32 %" #i ", %" #i ", %8\n\t"
            asm volatile(REP8(R8(X)) : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(w));
#undef X
        } else if (OP == MIX_PK_FMA) {      // alternate packed and plain: does a plain op slip under a packed one?
#define X(i) "v_pk_fma_f32 %" #i ", %" #i ", %16, %16\n\tv_fma_f32 %1" #i ", %1" #i ", %17, %18\n\t"
            // (operand numbering below: p0..p7 = %0..%7, a0..a7 = %8..%15) -- spelled out instead of the macro
#undef X
            asm volatile(REP8(
                "v_pk_fma_f32 %0, %0, %16, %16\n\tv_fma_f32 %8, %8, %17, %18\n\t"
                "v_pk_fma_f32 %1, %1, %16, %16\n\tv_fma_f32 %9, %9, %17, %18\n\t"
                "v_pk_fma_f32 %2, %2, %16, %16\n\tv_fma_f32 %10, %10, %17, %18\n\t"
                "v_pk_fma_f32 %3, %3, %16, %16\n\tv_fma_f32 %11, %11, %17, %18\n\t")
                : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]),
                  "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                : "v"(w), "v"(w.x), "v"(c));
        } else if (OP == FMAC_DPP) {
#define X(i) "v_fmac_f32_dpp %" #i ", %" #i ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
            asm volatile(REP8(R8(X)) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));
#undef X
        } else if (OP == MOV_DPP) {
#define X(i) "v_mov_b32_dpp %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
            asm volatile(REP8(R8(X)) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
#undef X
        } else if (OP == RCP) {
#define X(i) "v_rcp_f32_e32 %" #i ", %" #i "\n\t"
            asm volatile(REP8(R8(X)) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
#undef X
        } else if (OP == CNDMASK) {
#define X(i) "v_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc\n\t"
            asm volatile("v_cmp_gt_f32_e32 vcc, %8, %0\n\t" REP8(R8(X)) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c) : "vcc");
#undef X
        } else if (OP == MAX3) {
#define X(i) "v_max3_f32 %" #i ", |%" #i "|, |%8|, %9\n\t"
            asm volatile(REP8(R8(X)) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(w.x), "v"(c));
#undef X
        } else if (OP == BFI) {
#define X(i) "v_bfi_b32 %" #i ", %8, %" #i ", %9\n\t"
            asm volatile(REP8(R8(X)) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(0x7fffffff), "v"(c));
#undef X
        } else if (OP == PERM32) {
            asm volatile(REP8("v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\t"
                              "v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\t")
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
        } else if (OP == PERM16) {
            asm volatile(REP8("v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\tv_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7\n\t"
                              "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\tv_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7\n\t")
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
        } else if (OP == DSW64) {
#define X(i) "ds_write_b64 %0, %1 offset:" #i "*512\n\t"
            asm volatile(REP8(R8(X)) "s_waitcnt lgkmcnt(0)\n\t" :: "v"(la), "v"(p[0]) : "memory");
#undef X
        } else if (OP == DSW32) {
#define X(i) "ds_write_b32 %0, %1 offset:" #i "*256\n\t"
            asm volatile(REP8(R8(X)) "s_waitcnt lgkmcnt(0)\n\t" :: "v"(la4), "v"(a[0]) : "memory");
#undef X
        } else if (OP == DSW128) {
            asm volatile(REP8("ds_write_b128 %0, %1\n\tds_write_b128 %0, %1 offset:1024\n\tds_write_b128 %0, %1 offset:2048\n\tds_write_b128 %0, %1 offset:3072\n\t"
                              "ds_write_b128 %0, %1\n\tds_write_b128 %0, %1 offset:1024\n\tds_write_b128 %0, %1 offset:2048\n\tds_write_b128 %0, %1 offset:3072\n\t")
                         "s_waitcnt lgkmcnt(0)\n\t" :: "v"(la16), "v"(q[0]) : "memory");
        } else if (OP == DSR64) {
#define X(i) "ds_read_b64 %" #i ", %8 offset:" #i "*512\n\t"
            asm volatile(REP8(R8(X) "s_waitcnt lgkmcnt(0)\n\t")
                         : "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]), "=&v"(p[4]), "=&v"(p[5]), "=&v"(p[6]), "=&v"(p[7]) : "v"(la) : "memory");
#undef X
        } else if (OP == DSR32) {
#define X(i) "ds_read_b32 %" #i ", %8 offset:" #i "*256\n\t"
            asm volatile(REP8(R8(X) "s_waitcnt lgkmcnt(0)\n\t")
                         : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(a[4]), "=&v"(a[5]), "=&v"(a[6]), "=&v"(a[7]) : "v"(la4) : "memory");
#undef X
        } else if (OP == DSR128) {
            asm volatile(REP8("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\t"
                              "s_waitcnt lgkmcnt(0)\n\t"
                              "ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\t"
                              "s_waitcnt lgkmcnt(0)\n\t")
                         : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]) : "v"(la16) : "memory");
        } else if (OP == CND_E64) {
#define X(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, %9\n\t"
            asm volatile(REP8(R8(X)) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c), "s"(0x5555aaaa5555aaaaull));
#undef X
        } else if (OP == CND_VCC2) {        // destination differs from the sources' registers of the neighbours
            asm volatile("v_cmp_gt_f32_e32 vcc, %8, %0\n\t"
                         REP8("v_cndmask_b32_e32 %0, %1, %8, vcc\n\tv_cndmask_b32_e32 %2, %3, %8, vcc\n\tv_cndmask_b32_e32 %4, %5, %8, vcc\n\tv_cndmask_b32_e32 %6, %7, %8, vcc\n\t"
                              "v_cndmask_b32_e32 %1, %0, %8, vcc\n\tv_cndmask_b32_e32 %3, %2, %8, vcc\n\tv_cndmask_b32_e32 %5, %4, %8, vcc\n\tv_cndmask_b32_e32 %7, %6, %8, vcc\n\t")
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c) : "vcc");
        } else if (OP == CMP_VCC) {
#define X(i) "v_cmp_gt_f32_e32 vcc, %8, %" #i "\n\t"
            asm volatile(REP8(R8(X)) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c) : "vcc");
#undef X
        } else if (OP == CMP_E64) {
            unsigned long long m0, m1;
            asm volatile(REP8("v_cmp_gt_f32_e64 %8, %10, %0\n\tv_cmp_gt_f32_e64 %9, %10, %1\n\tv_cmp_gt_f32_e64 %8, %10, %2\n\tv_cmp_gt_f32_e64 %9, %10, %3\n\t"
                              "v_cmp_gt_f32_e64 %8, %10, %4\n\tv_cmp_gt_f32_e64 %9, %10, %5\n\tv_cmp_gt_f32_e64 %8, %10, %6\n\tv_cmp_gt_f32_e64 %9, %10, %7\n\t")
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "=&s"(m0), "=&s"(m1) : "v"(c));
            a[0] += (float)(m0 & 1) + (float)(m1 & 1);
        } else if (OP == MUL) {
#define X(i) "v_mul_f32_e32 %" #i ", %8, %" #i "\n\t"
            asm volatile(REP8(R8(X)) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(w.x));
#undef X
        } else if (OP == MUL_SGPR) {
#define X(i) "v_mul_f32_e32 %" #i ", %8, %" #i "\n\t"
            asm volatile(REP8(R8(X)) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "s"(s));
#undef X
        } else if (OP == FMAC) {
#define X(i) "v_fmac_f32_e32 %" #i ", %8, %9\n\t"
            asm volatile(REP8(R8(X)) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(w.x), "v"(c));
#undef X
        } else if (OP == FMAC_SGPR) {
#define X(i) "v_fmac_f32_e32 %" #i ", %8, %9\n\t"
            asm volatile(REP8(R8(X)) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "s"(s), "v"(c));
#undef X
        } else if (OP == FMAAK) {
#define X(i) "v_fmaak_f32 %" #i ", %" #i ", %8, 0x3f800001\n\t"
            asm volatile(REP8(R8(X)) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));
#undef X
        } else if (OP == FMA_NEG) {
#define X(i) "v_fma_f32 %" #i ", -%" #i ", %8, %9\n\t"
            asm volatile(REP8(R8(X)) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(w.x), "v"(c));
#undef X
        } else if (OP == FMA_ABS) {
#define X(i) "v_fma_f32 %" #i ", |%" #i "|, %8, %9\n\t"
            asm volatile(REP8(R8(X)) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(w.x), "v"(c));
#undef X
        } else if (OP == FMA_2SRC) {        // two distinct VGPR sources only (a*a + c)
#define X(i) "v_fma_f32 %" #i ", %" #i ", %" #i ", %8\n\t"
            asm volatile(REP8(R8(X)) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));
#undef X
        } else if (OP == PKFMA_2SRC) {
#define X(i) "v_pk_fma_f32 %" #i ", %" #i ", %" #i ", %8\n\t"
            asm volatile(REP8(R8(X)) : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(w));
#undef X
        } else if (OP == AND) {
#define X(i) "v_and_b32_e32 %" #i ", %8, %" #i "\n\t"
            asm volatile(REP8(R8(X)) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(0x7fffffff));
#undef X
        } else if (OP == XOR) {
#define X(i) "v_xor_b32_e32 %" #i ", %8, %" #i "\n\t"
            asm volatile(REP8(R8(X)) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(0x80000000));
#undef X
        } else if (OP == LSHL) {
#define X(i) "v_lshlrev_b32_e32 %" #i ", 1, %" #i "\n\t"
            asm volatile(REP8(R8(X)) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
#undef X
        } else if (OP == MIN) {
#define X(i) "v_min_f32_e32 %" #i ", %8, %" #i "\n\t"
            asm volatile(REP8(R8(X)) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(w.x));
#undef X
        } else if (OP == MED3) {
#define X(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n\t"
            asm volatile(REP8(R8(X)) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(w.x), "v"(c));
#undef X
        } else if (OP == PKFMA_SGPR) {
            const unsigned long long sp = 0x3f8000013f800001ull;
#define X(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n\t"
            asm volatile(REP8(R8(X)) : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "s"(sp), "v"(w));
#undef X
        } else if (OP == PKFMA_NEG) {
#define X(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
            asm volatile(REP8(R8(X)) : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(w), "v"(w));
#undef X
        } else if (OP == MOV) {
            asm volatile(REP8("v_mov_b32_e32 %0, %1\n\tv_mov_b32_e32 %2, %3\n\tv_mov_b32_e32 %4, %5\n\tv_mov_b32_e32 %6, %7\n\t"
                              "v_mov_b32_e32 %1, %0\n\tv_mov_b32_e32 %3, %2\n\tv_mov_b32_e32 %5, %4\n\tv_mov_b32_e32 %7, %6\n\t")
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
        } else if (OP == MUL_DPP) {
#define X(i) "v_mul_f32_dpp %" #i ", %" #i ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
            asm volatile(REP8(R8(X)) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(w.x));
#undef X
        } else if (OP == ADD_NEG) {
#define X(i) "v_add_f32_e64 %" #i ", %8, -%" #i "\n\t"
            asm volatile(REP8(R8(X)) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));
#undef X
        } else if (OP == SUB) {
#define X(i) "v_sub_f32_e32 %" #i ", %8, %" #i "\n\t"
            asm volatile(REP8(R8(X)) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));
#undef X
        } else if (OP == NOPS) {
            asm volatile(REP8("s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\t"));
        }
    }
    a[0] += q[0].x + q[1].y + q[2].z + q[3].w;
    (void)lds;
}

// waves 0..3 of the block (one per SIMD) run OPA; waves 4..7 (second wave on each SIMD), if present, run OPB
template <int OPA, int OPB>
__global__ __launch_bounds__(512) void k_probe(float *out, uint64_t *cyc, unsigned *hwid, int reps, float s)
{
    extern __shared__ char lds[];
    float a[8];
    v2f p[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = 1.0f + 1e-3f * (threadIdx.x + i); p[i] = (v2f){a[i], 0.5f * a[i]}; }
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) body<OPA>(a, p, s, lds, reps);
    else body<OPB>(a, p, s, lds, reps);
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) r += a[i] + p[i].x + p[i].y;
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) {
        cyc[(size_t)blockIdx.x * 8 + wave] = t1 - t0;
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        hwid[(size_t)blockIdx.x * 8 + wave] = ((xcc & 0xf) << 16) | (hw & 0xffff);
    }
}

static float *d_out;
static uint64_t *d_cyc;
static unsigned *d_hw;
static int g_reps = 512;

template <int OPA, int OPB> static void run(const char *name, int threads, int blocks_per_cu)
{
    const int reps = g_reps, cus = 256, blocks = cus * blocks_per_cu;
    const size_t lds = 8 * 4096;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int it = 0; it < 2; it++) {
        (void)hipMemset(d_cyc, 0, sizeof(uint64_t) * 8 * blocks);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k_probe<OPA, OPB>), dim3(blocks), dim3(threads), lds, 0, d_out, d_cyc, d_hw, reps, 1.0000001f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<uint64_t> c(8 * (size_t)blocks);
    (void)hipMemcpy(c.data(), d_cyc, sizeof(uint64_t) * c.size(), hipMemcpyDeviceToHost);
    std::vector<unsigned> hw(8 * (size_t)blocks);
    (void)hipMemcpy(hw.data(), d_hw, sizeof(unsigned) * hw.size(), hipMemcpyDeviceToHost);
    // census: waves per SIMD actually co-scheduled = waves that reported the same (xcc, se, sh, cu, simd)
    std::map<unsigned, int> per_simd;
    double sa = 0, sb = 0, mina = 1e30, maxa = 0; size_t na = 0, nb = 0;
    for (int b = 0; b < blocks; b++)
        for (int w = 0; w < threads / 64; w++) {
            const double c1 = (double)c[8 * (size_t)b + w];
            per_simd[(hw[8 * (size_t)b + w] >> 4) & 0xffff3]++;     // drop wave id (bits 3:0) and pipe id (7:6)
            if (w < 4) { sa += c1; na++; if (c1 < mina) mina = c1; if (c1 > maxa) maxa = c1; } else { sb += c1; nb++; }
        }
    int wmin = 1 << 30, wmax = 0;
    for (auto &kv : per_simd) { if (kv.second < wmin) wmin = kv.second; if (kv.second > wmax) wmax = kv.second; }
    const double n_inst = 64.0 * reps;
    const int wps = (threads / 256) * blocks_per_cu;          // nominal waves per SIMD
    printf("%-22s w/SIMD %d (census: %zu SIMDs, %d..%d waves): A %6.2f [%6.2f..%6.2f] cyc/inst/wave", name, wps, per_simd.size(), wmin, wmax,
           sa / na / n_inst, mina / n_inst, maxa / n_inst);
    if (nb) printf("  B %6.2f", sb / nb / n_inst);
    printf("  wall %.1f us = %.2f cyc/inst/SIMD @2.4GHz\n", ms * 1e3, ms * 1e-3 * 2.4e9 / (n_inst * wps));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
}

#define SOLO(OP) do { run<OP, NONE>(#OP, 256, 1); run<OP, NONE>(#OP, 256, 2); run<OP, NONE>(#OP, 256, 3); run<OP, NONE>(#OP, 256, 4); } while (0)
#define PAIR(A, B) do { run<A, B>(#A " | " #B, 512, 1); run<A, B>(#A " | " #B, 512, 2); } while (0)

int main()
{
    (void)hipMalloc(&d_out, sizeof(float) * 512 * 1024);
    (void)hipMalloc(&d_cyc, sizeof(uint64_t) * 8 * 1024);
    (void)hipMalloc(&d_hw, sizeof(unsigned) * 8 * 1024);
#define S34(OP) do { run<OP, NONE>(#OP, 256, 3); run<OP, NONE>(#OP, 256, 4); } while (0)
    g_reps = 256;
    S34(FMA); S34(FMA_2SRC); S34(FMAC); S34(FMAC_SGPR); S34(FMAAK); S34(FMA_NEG); S34(FMA_ABS); S34(FMA_SGPR); S34(MUL); S34(MUL_SGPR); S34(ADD); S34(SUB); S34(ADD_NEG);
    S34(MIN); S34(MED3); S34(MAX3); S34(AND); S34(XOR); S34(LSHL); S34(BFI); S34(MOV); S34(MUL_DPP); S34(FMAC_DPP); S34(MOV_DPP);
    S34(PKFMA); S34(PKFMA_2SRC); S34(PKFMA_SGPR); S34(PKFMA_NEG); S34(PKADD); S34(PKMUL);
    S34(CNDMASK); S34(CND_VCC2); S34(CND_E64); S34(CMP_VCC); S34(CMP_E64); S34(RCP);
    return 0;
}
