// ubench.hip — developer tool: issue cost of the instruction kinds the lane-per-rollout pass is
// made of, on gfx950, at 1, 2 and 4 waves per SIMD (one workgroup per CU, every CU busy).
// Answers the question DESIGN.md 4.2 turns on: with two waves on a SIMD, is the pass bound by
// each wave's own serial issue (one instruction of any kind per ~4 cycles per wave) or by the
// SIMD's VALU (one wave64 VALU instruction per 2 or per 4 cycles)?
//   hipcc -O2 --offload-arch=gfx950 -o ubench ubench.hip && ./ubench      (needs a GPU)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e__ = (x);                                                          \
    if (e__ != hipSuccess) {                                                       \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__));                    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

#define X2(A) A A
#define X4(A) X2(A) X2(A)
#define X8(A) X4(A) X4(A)

enum Kind {
  K_FMA = 0, K_FMA_DEP, K_PK_FMA, K_PK_MUL, K_PK_ADD, K_ADD_F64, K_CVT_F64, K_FRACT_FLR, K_EXP, K_SQRT, K_MAX3,
  K_CMP_CND, K_VALU_SALU, K_VALU_SALU2, K_GPRIDX, K_LDS_U8, K_LDS_CHAIN, K_MAD_U24, K_RNDNE, K_MOV, K_FMA_SGPR,
  K_PK_FMA_SGPR, K_FMA_LIT, K_COUNT
};

struct KindInfo {
  const char* name;
  int instr;   // instructions per asm block
};
static const KindInfo kInfo[K_COUNT] = {
  {"v_fma_f32 x16, 8 independent chains", 16},
  {"v_fma_f32 x16, one dependent chain", 16},
  {"v_pk_fma_f32 x16, 8 chains", 16},
  {"v_pk_mul_f32 x16, 8 chains", 16},
  {"v_pk_add_f32 x16, 8 chains", 16},
  {"v_add_f64 x16, 8 chains", 16},
  {"v_cvt_f64_f32 + v_cvt_f32_f64, 8 pairs", 16},
  {"v_fract_f32 + v_cvt_flr_i32_f32, 8 pairs", 16},
  {"v_exp_f32 x16", 16},
  {"v_sqrt_f32 x16", 16},
  {"v_max3_f32 x16", 16},
  {"v_cmp_lt_f32 + v_cndmask_b32, 8 pairs", 16},
  {"v_fma_f32 + s_add_u32, 8 pairs", 16},
  {"v_fma_f32 + 2 s_add_u32, 8 triples", 24},
  {"s_set_gpr_idx_on + v_mov_b32 + off, 8 triples", 24},
  {"ds_read_u8 x16 independent, one wait", 16},
  {"ds_read_u8 -> ds_read_b64 dependent, 4 pairs, waited each", 8},
  {"v_mad_u32_u24 x16", 16},
  {"v_rndne_f32 x16", 16},
  {"v_mov_b32 x16", 16},
  {"v_fma_f32 x16 with an SGPR operand", 16},
  {"v_pk_fma_f32 x16 with an SGPR-pair operand", 16},
  {"v_fmaak_f32 (32-bit literal) x16", 16},
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(1024) bench(int iters, unsigned long long* out, float seed)
{
  __shared__ unsigned char lds[8192];
  const int lane = threadIdx.x & 63;
  float a0 = seed + lane, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f,
        a7 = a0 + 7.f;
  const float m = 0.999f, c = 1e-3f;
  f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4},
        p7 = {a7, a6};
  const f32x2 pm = {m, m}, pc = {c, c};
  double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
  const double dc = 1e-3;
  int i0 = lane, i1 = lane + 1, i2 = lane + 2, i3 = lane + 3, i4 = lane + 4, i5 = lane + 5, i6 = lane + 6, i7 = lane + 7;
  unsigned s0 = (unsigned)__builtin_amdgcn_readfirstlane((int)seed), s1 = s0 + 1;
  const float sm = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m)));
  const float sc = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, c)));
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (unsigned char)(i & 63);
  __syncthreads();
  unsigned l0 = (unsigned)(lane * 4), l1 = l0 + 256, l2 = l0 + 512, l3 = l0 + 768;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == K_FMA) {
      asm volatile(X2("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n"
                      "v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n"
                      "v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                   : "v"(m), "v"(c));
    } else if constexpr (KIND == K_FMA_DEP) {
      asm volatile(X8("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n") : "+v"(a0) : "v"(m), "v"(c));
    } else if constexpr (KIND == K_PK_FMA) {
      asm volatile(X2("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n"
                      "v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n"
                      "v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n")
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                   : "v"(pm), "v"(pc));
    } else if constexpr (KIND == K_PK_MUL) {
      asm volatile(X2("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n"
                      "v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n"
                      "v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n")
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                   : "v"(pm));
    } else if constexpr (KIND == K_PK_ADD) {
      asm volatile(X2("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n"
                      "v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n"
                      "v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n")
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                   : "v"(pc));
    } else if constexpr (KIND == K_ADD_F64) {
      asm volatile(X2("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                      "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n")
                   : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
                   : "v"(dc));
    } else if constexpr (KIND == K_CVT_F64) {
      asm volatile("v_cvt_f64_f32 %8, %0\n v_cvt_f64_f32 %9, %1\n v_cvt_f64_f32 %10, %2\n v_cvt_f64_f32 %11, %3\n"
                   "v_cvt_f64_f32 %12, %4\n v_cvt_f64_f32 %13, %5\n v_cvt_f64_f32 %14, %6\n v_cvt_f64_f32 %15, %7\n"
                   "v_cvt_f32_f64 %0, %8\n v_cvt_f32_f64 %1, %9\n v_cvt_f32_f64 %2, %10\n v_cvt_f32_f64 %3, %11\n"
                   "v_cvt_f32_f64 %4, %12\n v_cvt_f32_f64 %5, %13\n v_cvt_f32_f64 %6, %14\n v_cvt_f32_f64 %7, %15\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(d0), "+v"(d1),
                     "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));
    } else if constexpr (KIND == K_FRACT_FLR) {
      asm volatile("v_fract_f32 %8, %0\n v_cvt_flr_i32_f32 %12, %0\n v_fract_f32 %9, %1\n v_cvt_flr_i32_f32 %13, %1\n"
                   "v_fract_f32 %10, %2\n v_cvt_flr_i32_f32 %14, %2\n v_fract_f32 %11, %3\n v_cvt_flr_i32_f32 %15, %3\n"
                   "v_fract_f32 %0, %4\n v_cvt_flr_i32_f32 %12, %4\n v_fract_f32 %1, %5\n v_cvt_flr_i32_f32 %13, %5\n"
                   "v_fract_f32 %2, %6\n v_cvt_flr_i32_f32 %14, %6\n v_fract_f32 %3, %7\n v_cvt_flr_i32_f32 %15, %7\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(p0.x), "+v"(p1.x),
                     "+v"(p2.x), "+v"(p3.x), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));
    } else if constexpr (KIND == K_EXP) {
      asm volatile(X2("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                      "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if constexpr (KIND == K_SQRT) {
      asm volatile(X2("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                      "v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if constexpr (KIND == K_MAX3) {
      asm volatile(X2("v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n"
                      "v_max3_f32 %3, %3, %8, %9\n v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n"
                      "v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                   : "v"(m), "v"(c));
    } else if constexpr (KIND == K_CMP_CND) {
      asm volatile(X2("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %9, vcc\n v_cmp_lt_f32 vcc, %1, %8\n"
                      "v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %9, vcc\n"
                      "v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %9, vcc\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
                   : "v"(m), "v"(c), "v"(a4), "v"(a5), "v"(a6), "v"(a7)
                   : "vcc");
    } else if constexpr (KIND == K_VALU_SALU) {
      asm volatile(X2("v_fma_f32 %0, %0, %8, %9\n s_add_u32 s20, s20, 1\n v_fma_f32 %1, %1, %8, %9\n s_add_u32 s21, s21, 1\n"
                      "v_fma_f32 %2, %2, %8, %9\n s_add_u32 s20, s20, 1\n v_fma_f32 %3, %3, %8, %9\n s_add_u32 s21, s21, 1\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                   : "v"(m), "v"(c)
                   : "scc", "s20", "s21");
    } else if constexpr (KIND == K_VALU_SALU2) {
      asm volatile(X2("v_fma_f32 %0, %0, %8, %9\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n"
                      "v_fma_f32 %1, %1, %8, %9\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n"
                      "v_fma_f32 %2, %2, %8, %9\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n"
                      "v_fma_f32 %3, %3, %8, %9\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                   : "v"(m), "v"(c)
                   : "scc", "s20", "s21");
    } else if constexpr (KIND == K_GPRIDX) {
      asm volatile(X8("s_set_gpr_idx_on %2, gpr_idx(DST)\n v_mov_b32 %0, %1\n s_set_gpr_idx_off\n")
                   : "+v"(a0)
                   : "v"(a1), "s"(0)
                   : "m0");
    } else if constexpr (KIND == K_LDS_U8) {
      asm volatile(X4("ds_read_u8 %0, %4\n ds_read_u8 %1, %5\n ds_read_u8 %2, %6\n ds_read_u8 %3, %7\n")
                   "s_waitcnt lgkmcnt(0)\n"
                   : "=&v"(i0), "=&v"(i1), "=&v"(i2), "=&v"(i3)
                   : "v"(l0), "v"(l1), "v"(l2), "v"(l3));
    } else if constexpr (KIND == K_LDS_CHAIN) {
      asm volatile(X4("ds_read_u8 %0, %2\n s_waitcnt lgkmcnt(0)\n v_lshlrev_b32 %0, 3, %0\n ds_read_b64 %1, %0\n"
                      "s_waitcnt lgkmcnt(0)\n")
                   : "=&v"(i0), "=&v"(d0)
                   : "v"(l0));
    } else if constexpr (KIND == K_MAD_U24) {
      asm volatile(X2("v_mad_u32_u24 %0, %0, %8, %9\n v_mad_u32_u24 %1, %1, %8, %9\n v_mad_u32_u24 %2, %2, %8, %9\n"
                      "v_mad_u32_u24 %3, %3, %8, %9\n v_mad_u32_u24 %4, %4, %8, %9\n v_mad_u32_u24 %5, %5, %8, %9\n"
                      "v_mad_u32_u24 %6, %6, %8, %9\n v_mad_u32_u24 %7, %7, %8, %9\n")
                   : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7)
                   : "v"(l0), "v"(l1));
    } else if constexpr (KIND == K_RNDNE) {
      asm volatile(X2("v_rndne_f32 %0, %0\n v_rndne_f32 %1, %1\n v_rndne_f32 %2, %2\n v_rndne_f32 %3, %3\n"
                      "v_rndne_f32 %4, %4\n v_rndne_f32 %5, %5\n v_rndne_f32 %6, %6\n v_rndne_f32 %7, %7\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if constexpr (KIND == K_MOV) {
      asm volatile(X2("v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %6\n v_mov_b32 %3, %7\n"
                      "v_mov_b32 %4, %0\n v_mov_b32 %5, %1\n v_mov_b32 %6, %2\n v_mov_b32 %7, %3\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if constexpr (KIND == K_FMA_SGPR) {
      asm volatile(X2("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n"
                      "v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n"
                      "v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                   : "s"(sm), "v"(c));
    } else if constexpr (KIND == K_PK_FMA_SGPR) {
      const f32x2 spm = {sm, sc};
      asm volatile(X2("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n"
                      "v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n"
                      "v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n")
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                   : "s"(spm), "v"(pc));
    } else if constexpr (KIND == K_FMA_LIT) {
      asm volatile(X2("v_fmaak_f32 %0, %0, %8, 0x3c088735\n v_fmaak_f32 %1, %1, %8, 0x3c088735\n"
                      "v_fmaak_f32 %2, %2, %8, 0x3c088735\n v_fmaak_f32 %3, %3, %8, 0x3c088735\n"
                      "v_fmaak_f32 %4, %4, %8, 0x3c088735\n v_fmaak_f32 %5, %5, %8, 0x3c088735\n"
                      "v_fmaak_f32 %6, %6, %8, 0x3c088735\n v_fmaak_f32 %7, %7, %8, 0x3c088735\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                   : "v"(m));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
  // keep every chain alive
  const float sink = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.x + p4.x + p5.x + p6.x + p7.x +
                     p0.y + p1.y + p2.y + p3.y + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) +
                     (float)(i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7 + (int)s0 + (int)s1);
  if (sink == 123.456f) out[0] = 0;
}

template <int KIND>
static void run_kind(int ncu, unsigned long long* d_out, std::vector<unsigned long long>& h)
{
  const int iters = 2000;
  printf("%-58s", kInfo[KIND].name);
  for (int block : {64, 256, 512, 1024}) {   // 1 wave on the CU, then 1, 2, 4 waves per SIMD
    const int waves = ncu * (block / 64);
    hipLaunchKernelGGL(bench<KIND>, dim3(ncu), dim3(block), 0, 0, iters, d_out, 1.0f);   // warm-up
    hipLaunchKernelGGL(bench<KIND>, dim3(ncu), dim3(block), 0, 0, iters, d_out, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), d_out, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.begin() + waves);
    const double cyc = (double)h[waves / 2] / ((double)iters * kInfo[KIND].instr);
    const int per_simd = block >= 256 ? block / 256 : 1;
    printf("  %6.2f (%5.2f)", cyc, cyc / per_simd);
  }
  printf("\n");
}

template <int K>
static void run_all(int ncu, unsigned long long* d_out, std::vector<unsigned long long>& h)
{
  if constexpr (K < K_COUNT) {
    run_kind<K>(ncu, d_out, h);
    run_all<K + 1>(ncu, d_out, h);
  }
}

int main()
{
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("%s, %d CUs, clock %d kHz\n", prop.gcnArchName, ncu, prop.clockRate);
  printf("shader cycles (s_memtime) per instruction of ONE wave's stream; in brackets per instruction of the SIMD\n");
  printf("(= per-wave figure / waves per SIMD).  Columns: 1 wave per CU | 1 wave per SIMD | 2 per SIMD | 4 per SIMD\n");
  unsigned long long* d_out = nullptr;
  CK(hipMalloc(&d_out, ncu * 16 * sizeof(unsigned long long)));
  std::vector<unsigned long long> h(ncu * 16);
  run_all<0>(ncu, d_out, h);
  return 0;
}
