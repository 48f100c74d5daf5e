// smpc_device_math.h — device helpers shared by the two streaming passes
// (smpc_kernels.hip: wave per rollout, smpc_tpr.hip: lane per rollout).
#ifndef SMPC_DEVICE_MATH_H_
#define SMPC_DEVICE_MATH_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "smpc_dev.h"

// Keep a wave-uniform loop constant in a VGPR: the scoring loop has far more uniform
// values than the 102 SGPRs hold, and what the compiler spills from SGPRs comes back as one
// v_readlane per use.  VALU sources are free to be VGPRs.
template <typename V>
__device__ __forceinline__ V in_vgpr(V x)
{
  asm volatile("" : "+v"(x));
  return x;
}

// 1-ulp hardware sqrt / reciprocal (MODE 0: the terms they feed are continuous in
// their inputs, so a last-bit difference moves a cost by ~1e-7 relative)
__device__ __forceinline__ float fast_sqrt(float v) {return __builtin_amdgcn_sqrtf(v);}
__device__ __forceinline__ float fast_rcp(float v) {return __builtin_amdgcn_rcpf(v);}

// std::pow(double, unsigned) for the cost_power parameters (H5: pow promotes to double)
__device__ __forceinline__ double powu(double v, uint32_t p)
{
  if (p == 1u) return v;
  double r = 1.0;
  while (p) {
    if (p & 1u) r *= v;
    v *= v;
    p >>= 1;
  }
  return r;
}
// data.costs += xt::pow(v, power)
__device__ __forceinline__ float add_cost_pow(float c, double v, uint32_t power)
{
  return (float)((double)c + powu(v, power));
}

// utils::normalize_angles (tools/utils.hpp:258-263), double like the reference
__device__ __forceinline__ double normalize_angle(double a)
{
  const double theta = fmod(a + M_PI, 2.0 * M_PI);
  return theta <= 0.0 ? theta + M_PI : theta - M_PI;
}

// sin and cos of a rollout yaw.  Cody–Waite reduction by pi in two float pieces (each fma
// is exact or correctly rounded on the cancelled difference; what the two leave of pi is
// 3.4e-15, times |k| < 20 861 for |x| < 65536: 7e-11, three orders below the result's own
// rounding — the third piece rounds 1 and 2 carried cost an instruction per step and bought
// nothing) to r in [-pi/2, pi/2], near-
// minimax polynomials there (sin: r + r^3 S(r^2), 4.6e-9; cos: 1 - r^2/2 + r^4 C(r^2),
// 3.9e-10) and ONE sign, (-1)^k, for both — no quadrant swap, 7 instructions fewer per step
// than the pi/2 reduction.  Absolute error <= 1.3e-7 (libm's correctly rounded float: 3e-8)
// for |x| < 65536; relative accuracy near the zeros of cos is not kept (the rollout uses the
// products vx cos, vx sin, where only the absolute error counts).  Huge arguments take the
// library path.
__device__ __forceinline__ void smpc_sincos_fast(float x, float& sn, float& cs);
__device__ __forceinline__ void smpc_sincos(float x, float& sn, float& cs)
{
  if (__builtin_expect(!(fabsf(x) < 65536.0f), 0)) {
    sincosf(x, &sn, &cs);
    return;
  }
  smpc_sincos_fast(x, sn, cs);
}
// |x| < 65536 only (the caller checks)
__device__ __forceinline__ void smpc_sincos_fast(float x, float& sn, float& cs)
{
  // k = x / pi rounded to an integer, by adding 1.5 * 2^23 in the same fused multiply-add (the
  // sum's unit in the last place is 1: the rounding of the fma IS the rounding to an integer,
  // and the integer's parity is the sum's lowest mantissa bit) — one instruction instead of
  // multiply, v_rndne and v_cvt_i32, which issue at half the rate of an fma (tools/ubench)
#ifndef SMPC_X_MAGIC
#define SMPC_X_MAGIC 1
#endif
#ifndef SMPC_X_HWSIN
#define SMPC_X_HWSIN 0
#endif
  // SMPC_X_HWSIN: measured and rejected (tools/ubench/sincos_hw.hip, tools/variants.sh).  The
  // hardware's v_sin_f32 / v_cos_f32 on x / 2pi are 3 instructions instead of 20, but quarter
  // rate: the lane pass gains 4-6 % (60.5 -> 57.0 us, 394.6 -> 377.5 us) and the absolute error
  // grows from 1.3e-7 to 3.5e-7 for |yaw| <= pi, 7.3e-7 at 8 rad (the float product x / 2pi);
  // reducing by pi first (form 2) keeps 2e-7 and gains 2-4 %.  Not worth three times the
  // distance to the reference's libm.
#if SMPC_X_HWSIN == 1
  // experiment: the hardware's v_sin_f32 / v_cos_f32 (argument in revolutions)
  const float t = x * 0.15915494309189535f;
  sn = __builtin_amdgcn_sinf(t);
  cs = __builtin_amdgcn_cosf(t);
  return;
#elif SMPC_X_HWSIN == 2
  // experiment: reduce by pi first (exactly), then the hardware on |r| <= pi/2
  {
    const float kf = fmaf(x, 0.31830987334251404f, 12582912.0f);
    const float k = kf - 12582912.0f;
    float r = fmaf(-k, 3.1415927410125732f, x);
    r = fmaf(-k, -8.742277657347586e-08f, r);
    const float t = r * 0.15915494309189535f;
    const uint32_t sign = __float_as_uint(kf) << 31;
    sn = __uint_as_float(__float_as_uint(__builtin_amdgcn_sinf(t)) ^ sign);
    cs = __uint_as_float(__float_as_uint(__builtin_amdgcn_cosf(t)) ^ sign);
    return;
  }
#endif
#if SMPC_X_MAGIC
  const float kf = fmaf(x, 0.31830987334251404f, 12582912.0f);
  const float k = kf - 12582912.0f;
#else
  const float k = rintf(x * 0.31830987334251404f);
#endif
  float r = fmaf(-k, 3.1415927410125732f, x);
  r = fmaf(-k, -8.742277657347586e-08f, r);
  const float z = r * r;
  float ps = fmaf(z, 2.60005474e-06f, -1.98066152e-04f);
  ps = fmaf(ps, z, 8.33301729e-03f);
  ps = fmaf(ps, z, -1.66666571e-01f);
  const float s = fmaf(ps * z, r, r);
  float pc = fmaf(z, -2.61938150e-07f, 2.47693042e-05f);
  pc = fmaf(pc, z, -1.38885692e-03f);
  pc = fmaf(pc, z, 4.16666558e-02f);
  const float c = fmaf(pc * z, z, fmaf(z, -0.5f, 1.0f));
#if SMPC_X_MAGIC
  const uint32_t sign = __float_as_uint(kf) << 31;
#else
  const uint32_t sign = (uint32_t)(int)k << 31;
#endif
  sn = __uint_as_float(__float_as_uint(s) ^ sign);
  cs = __uint_as_float(__float_as_uint(c) ^ sign);
}

// Costmap2D::worldToMap along one axis, exactly as nav2_costmap_2d does it:
// reject w < origin, else (unsigned)((w - origin) / resolution) in double.
__device__ __forceinline__ bool cell_index_exact(double w, double o, double res, uint32_t n,
                                                 uint32_t& m)
{
  if (w < o) return false;
  const double q = (w - o) / res;
  if (!(q < 4294967296.0)) return false;  // the reference's cast would be UB: off-map
  m = (uint32_t)q;
  return m < n;
}

// Costmap2D::worldToMap + getCost through the LDS window (global fallback).
// Off-map -> NO_INFORMATION (obstacles_critic.cpp:209-212).
//
// The cell index is first formed in float, q = (x - origin_f) * (1/res)_f, whose
// distance to the double quotient the reference truncates is bounded by
// p.cell_eps (host: origin rounding + 3 float roundings).  Only a lane whose q
// lies within that bound of a cell edge can truncate differently; those lanes
// (a fraction ~4*cell_eps) redo both axes in double with the true division, so
// every lookup reads the cell the reference reads.
struct CellConsts {   // loop constants of cost_at, held in VGPRs
  float oxf, oyf, rinvf, lo, hi;
};
__device__ __forceinline__ uint32_t cost_at(const SmpcDev& p, const CellConsts& k,
                                            const uint8_t* s_map, float x, float y)
{
  const float qx = (x - k.oxf) * k.rinvf, qy = (y - k.oyf) * k.rinvf;
  const float fx = floorf(qx), fy = floorf(qy);
  const float rx = qx - fx, ry = qy - fy;
  const float lo = k.lo, hi = k.hi;
  uint32_t mx = (uint32_t)(int)fx, my = (uint32_t)(int)fy;   // negative / huge -> >= W
  bool on = mx < p.W && my < p.H;
  if (__builtin_expect(!(rx >= lo && rx <= hi && ry >= lo && ry <= hi), 0)) {
    on = cell_index_exact((double)x, p.ox, p.res, p.W, mx);
    on = cell_index_exact((double)y, p.oy, p.res, p.H, my) && on;
  }
  if (!on) return 255u;
  const uint32_t lx = mx - (uint32_t)p.win_x0, ly = my - (uint32_t)p.win_y0;
  // two separate loads (never a select of an LDS and a global pointer)
  const bool inw = lx < (uint32_t)p.win_w && ly < (uint32_t)p.win_h;
  uint32_t c = s_map[inw ? ly * p.win_w + lx : 0u];
  if (__builtin_expect(!inw, 0)) c = p.map[(size_t)my * p.W + mx];
  return c;
}


// ---- consider_footprint = true (general pass only): nav2_costmap_2d's
// FootprintCollisionChecker::footprintCostAtPose with nav2_util::LineIterator, restated
// (third party, ROS 2 Humble).  Costs come from the LDS window when the cell is inside it.
__device__ __noinline__ uint32_t cell_cost(const SmpcDev& p, const uint8_t* s_map, uint32_t mx,
                                           uint32_t my)
{
  const uint32_t lx = mx - (uint32_t)p.win_x0, ly = my - (uint32_t)p.win_y0;
  const bool inw = lx < (uint32_t)p.win_w && ly < (uint32_t)p.win_h;
  uint32_t c = s_map[inw ? ly * p.win_w + lx : 0u];
  if (!inw) c = p.map[(size_t)my * p.W + mx];
  return c;
}

__device__ __noinline__ float footprint_line_cost(const SmpcDev& p, const uint8_t* s_map, int x0,
                                                  int x1, int y0, int y1)
{
  float cost = 0.f;
  const int deltax = abs(x1 - x0), deltay = abs(y1 - y0);
  int x = x0, y = y0;
  int xinc1 = x1 >= x0 ? 1 : -1, xinc2 = xinc1;
  int yinc1 = y1 >= y0 ? 1 : -1, yinc2 = yinc1;
  int den, num, numadd, numpixels;
  if (deltax >= deltay) {
    xinc1 = 0; yinc2 = 0; den = deltax; num = deltax / 2; numadd = deltay; numpixels = deltax;
  } else {
    xinc2 = 0; yinc1 = 0; den = deltay; num = deltay / 2; numadd = deltax; numpixels = deltay;
  }
  for (int curpixel = 0; curpixel <= numpixels; ++curpixel) {
    const float pc = (float)cell_cost(p, s_map, (uint32_t)x, (uint32_t)y);
    if (pc == 254.0f) return pc;          // LETHAL_OBSTACLE
    cost = fmaxf(cost, pc);
    num += numadd;
    if (num >= den) {
      num -= den;
      x += xinc1;
      y += yinc1;
    }
    x += xinc2;
    y += yinc2;
  }
  return cost;
}

__device__ __noinline__ float footprint_cost_at_pose(const SmpcDev& p, const uint8_t* s_map, float xf,
                                                     float yf, float thetaf)
{
  const double x = (double)xf, y = (double)yf, theta = (double)thetaf;
  const double cos_th = cos(theta), sin_th = sin(theta);
  const uint32_t n = p.fp_n;
  uint32_t x0 = 0, y0 = 0, x1 = 0, y1 = 0, xs = 0, ys = 0;
  float fc = 0.f;
  for (uint32_t i = 0; i < n; ++i) {
    const double wx = x + (p.fp_x[i] * cos_th - p.fp_y[i] * sin_th);
    const double wy = y + (p.fp_x[i] * sin_th + p.fp_y[i] * cos_th);
    uint32_t mx = 0, my = 0;
    bool on = cell_index_exact(wx, p.ox, p.res, p.W, mx);
    on = cell_index_exact(wy, p.oy, p.res, p.H, my) && on;
    if (!on) return 254.0f;               // a vertex off the map
    if (i == 0) {
      xs = x0 = mx;
      ys = y0 = my;
      x1 = mx;
      y1 = my;
      continue;
    }
    x1 = mx;
    y1 = my;
    fc = fmaxf(footprint_line_cost(p, s_map, (int)x0, (int)x1, (int)y0, (int)y1), fc);
    x0 = x1;
    y0 = y1;
    if (fc == 254.0f) return fc;
  }
  if (n == 0) return 254.0f;
  return fmaxf(footprint_line_cost(p, s_map, (int)xs, (int)x1, (int)ys, (int)y1), fc);
}

#endif
